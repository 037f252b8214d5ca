// 'Same' k^3 convolution with a TINY reduction-channel count (CK <= 4) on the MFMA pipe: the data gradient
// of out_tr.conv1 (ncls -> 32 channels, vnet.py:165) and in_tr.conv1 forward-type problems.
//
// conv_halo_mfma_k walks K in 8-channel chunks, so CK = 3 would run at 3/8 MFMA efficiency; the VALU kernel
// (conv_halo_valu_k) needs CK*CN wave-uniform weights per tap and measured 1.93 ms for 3 -> 32 @ 2x128^3
// (0.64 ms of FMA time).  Here the GEMM K dimension enumerates (tap, channel) pairs TIGHTLY: one
// v_mfma_f32_32x32x2_f32 consumes the same channel of two taps (lane half 0: tap t, lane half 1: tap
// t + ceil(taps/2)), so K = 125*CK has no padding beyond one tap.  The halo tile is staged planar
// ([channel][voxel], odd pitch) so every A operand is a conflict-free ds_read_b32 at
// voxel + tap_offset[t], the offsets coming from a small LDS table.
#include "msk_conv.h"
#include "msk_wbf.h"

typedef _Float16 tk_f16x8 __attribute__((ext_vector_type(8)));
 __attribute__((ext_vector_type(16)));

namespace {

struct TKArgs {
  const float* src;
  int sld;
  float* dst;
  int dld;
  int N, D, H, W;
  int CN, npad;
  const float* wt;  // [HT][CK][2][npad]
  unsigned wt_bytes, src_bytes;
  const float* bias;
  int accumulate;
  int tiles_d, tiles_h, tiles_w, nblk;
};

__device__ __forceinline__ int xcd_remap_tk(int bid, int nb) {
  const int q = nb >> 3, r = nb & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__global__ void __launch_bounds__(256)
pack_tightk_weights_k(const float* __restrict__ w, int A, int B, int ks, int swap, int flip, int CK, int CN, int npad,
                      float* __restrict__ out) {
  const int taps = ks * ks * ks, HT = (taps + 1) / 2;
  const long total = (long)HT * CK * 2 * npad;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int n = (int)(idx % npad);
    long r = idx / npad;
    const int h = (int)(r & 1);
    r >>= 1;
    const int c = (int)(r % CK);
    const int ti = (int)(r / CK);
    const int tap = h * HT + ti;
    float v = 0.f;
    if (tap < taps && n < CN) {
      const int st = flip ? taps - 1 - tap : tap;  // flipping all three axes reverses the linear tap index
      const int ia = swap ? n : c, ib = swap ? c : n;
      v = w[((long)ia * B + ib) * taps + st];
    }
    out[idx] = v;
  }
}

template <int TD, int TH, int TW, int KS, int CK>
__global__ void __launch_bounds__(256)
conv_halo_tightk_k(TKArgs a) {
  constexpr int P = KS / 2;
  constexpr int HD = TD + 2 * P, HH = TH + 2 * P, HW = TW + 2 * P;
  constexpr int NV = HD * HH * HW, NVP = NV | 1;
  constexpr int TAPS = KS * KS * KS, HT = (TAPS + 1) / 2;
  constexpr int MR = TD * TH * TW / 128;
  __shared__ float lds[CK * NVP];
  __shared__ int toff[2 * HT];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  int tile = xcd_remap_tk(blockIdx.x, a.nblk);
  const int twi = tile % a.tiles_w;
  tile /= a.tiles_w;
  const int thi = tile % a.tiles_h;
  tile /= a.tiles_h;
  const int tdi = tile % a.tiles_d;
  const int n = tile / a.tiles_d;
  const int d0 = tdi * TD, h0 = thi * TH, w0 = twi * TW;
  const int nt = blockIdx.y;

  for (int i = tid; i < 2 * HT; i += 256) {
    const int tap = (i / HT) * HT + (i % HT);
    toff[i] = tap < TAPS ? ((tap / (KS * KS)) * HH + (tap / KS) % KS) * HW + tap % KS : 0;
  }
  const __amdgpu_buffer_rsrc_t sres = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
  for (int hv = tid; hv < NV; hv += 256) {
    const int hd = hv / (HH * HW), rem = hv % (HH * HW), hh = rem / HW, hw = rem % HW;
    const int gd = d0 - P + hd, gh = h0 - P + hh, gw = w0 - P + hw;
    const bool in = gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
    // 32-bit offsets into a buffer resource (tensors are chunked below 4 GiB); halo voxels outside the volume read
    // zeros through an out-of-range offset
    const unsigned voff = in ? (unsigned)((((n * a.D + gd) * a.H + gh) * a.W + gw) * a.sld) * 4u : 0xFFFFFFF0u;
#pragma unroll
    for (int c = 0; c < CK; ++c)
      lds[c * NVP + hv] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sres, (int)voff, 4 * c, 0));
  }
  __syncthreads();

  int abase[MR];
#pragma unroll
  for (int r = 0; r < MR; ++r) {
    const int l = (wave * MR + r) * 32 + li;
    abase[r] = ((l / (TH * TW)) * HH + (l / TW) % TH) * HW + l % TW;
  }
  f32x16 acc[MR];
#pragma unroll
  for (int r = 0; r < MR; ++r)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[r][j] = 0.f;

  // weights through a raw buffer resource: the (tap pair, channel) part of the address is wave-uniform and rides in
  // the scalar offset, the lane part is one constant (a flat pointer cost ~3 VALU instructions per load)
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);
  const unsigned wlane = (unsigned)(lh * a.npad + nt * 32 + li) * 4u;
  const int* tl = toff + lh * HT;
#pragma unroll 3
  for (int ti = 0; ti < HT; ++ti) {
    const int off = tl[ti];
#pragma unroll
    for (int c = 0; c < CK; ++c) {
      const float b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
          wres, (int)wlane, (int)((unsigned)((ti * CK + c) * 2 * a.npad) * 4u), 0));
#pragma unroll
      for (int r = 0; r < MR; ++r)
        acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[c * NVP + abase[r] + off], b, acc[r], 0, 0, 0);
    }
  }

  const int co = nt * 32 + li;
  if (co < a.CN) {
    const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
    for (int r = 0; r < MR; ++r) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = (j & 3) + 8 * (j >> 2) + 4 * lh;
        const int l = (wave * MR + r) * 32 + row;
        const int gd = d0 + l / (TH * TW), gh = h0 + (l / TW) % TH, gw = w0 + l % TW;
        if (gd < a.D && gh < a.H && gw < a.W) {
          float* o = a.dst + ((((long)n * a.D + gd) * a.H + gh) * a.W + gw) * a.dld + co;
          float v = acc[r][j] + bv;
          if (a.accumulate) v += *o;
          *o = v;
        }
      }
    }
  }
}

template <int TD, int TH, int TW, int KS>
int launch_tk(msk_ctx* ctx, TKArgs& a, int CK, int ntn, const char* tag) {
  a.tiles_d = msk_cdiv(a.D, TD);
  a.tiles_h = msk_cdiv(a.H, TH);
  a.tiles_w = msk_cdiv(a.W, TW);
  const long nblk = (long)a.N * a.tiles_d * a.tiles_h * a.tiles_w;
  if (nblk > 0x7fffffff) return msk_fail(ctx, __FILE__, __LINE__, "conv_halo_tightk", "grid too large");
  a.nblk = (int)nblk;
  msk_launch_scope ls(ctx, tag);
  dim3 grid((unsigned)nblk, ntn);
  switch (CK) {
    case 1: hipLaunchKernelGGL((conv_halo_tightk_k<TD, TH, TW, KS, 1>), grid, dim3(256), 0, ctx->stream, a); break;
    case 2: hipLaunchKernelGGL((conv_halo_tightk_k<TD, TH, TW, KS, 2>), grid, dim3(256), 0, ctx->stream, a); break;
    case 3: hipLaunchKernelGGL((conv_halo_tightk_k<TD, TH, TW, KS, 3>), grid, dim3(256), 0, ctx->stream, a); break;
    default: hipLaunchKernelGGL((conv_halo_tightk_k<TD, TH, TW, KS, 4>), grid, dim3(256), 0, ctx->stream, a); break;
  }
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// The same problem class with fp16 two-piece operands (msk_wbf.h, option "conv_split" 2) for CN = 32 (out_tr.conv1's data
// gradient): the kd taps are folded into the K dimension of v_mfma_f32_16x16x32_f16 -- K = (kd 0..4, channel 0..3) = 20 of
// 32 -- so one instruction per (kh, kw) tap, output-channel tile and piece product covers all five kd taps: 150 instead of
// 188 x 4 MFMA issue slots per 32 positions, on a pipe 16x faster.  A workgroup owns a 16 x 16 (h, w) column and marches
// along D; the last planes of the (tiny: <= 4 channels) source live in an LDS ring of 8 slots as [piece][voxel][4 x fp16],
// so a lane's A fragment is two 8-byte reads (planes kd = 2 kg, 2 kg + 1; the padding lanes read a zero slot).  D is
// produced transposed (weights as the A operand): a lane ends up with 4 consecutive output channels of one position.
struct TKH2Args {
  const float* src;
  int sld;
  float* dst;
  int dld;
  int N, D, H, W, CK;
  const uint4* wb;  // [step 16][n tile 2][piece 2][lane 64]: A fragment (8 fp16: K = 8*(lane/16) .. +7 = units 8 m + 2 (lane/16), + 1) of output channel 16*nt + lane%16
  int accumulate;
  int tiles_h, tiles_w, segs, seg_len, nblk;
  const float* x_amax;
  const float* w_amax;
};

// Round 4: DENSE K.  A "unit" is the (<= 4)-channel vector of one voxel of one plane = 4 of the 32 K slots of
// v_mfma_f32_16x16x32_f16; an output position needs the 125 units (kh, kw, kd) of its receptive field.  Round 3 gave every
// (kh, kw) tap its own instruction with the five kd units in 20 of the 32 slots (25 matrix steps per plane, 47 % of the K slots
// useful for three classes); the units are now numbered u = (kh*5 + kw)*5 + kd and packed EIGHT per instruction -- matrix step
// m holds units 8 m .. 8 m + 7, lane group lk the two units 8 m + 2 lk, + 1 -- so a plane takes 16 steps instead of 25
// (units 125..127 are zero).  A lane's fragment is still two 8-byte LDS reads per piece, only their addresses differ per lane
// group (computed once per kernel: kTKUnitsPerStep tables in registers).
constexpr int kTKSteps = 16;
__global__ void __launch_bounds__(256)
pack_tkh2_weights_k(const float* __restrict__ w, int A, int B, int swap, int flip, int CK, const float* __restrict__ w_amax,
                    unsigned short* __restrict__ out) {
  const float sw = wbf_scale_of(w_amax);
  const int total = kTKSteps * 2 * 64 * 8;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int e = idx & 7, lane = (idx >> 3) & 63, nt = (idx >> 9) & 1, m = idx >> 10;
    const int n = nt * 16 + (lane & 15);
    const int u = 8 * m + 2 * (lane >> 4) + (e >> 2), c = e & 3;   // K slot 8 (lane / 16) + e of step m
    float v = 0.f;
    if (u < 125 && c < CK) {
      const int kd = u % 5, tap2 = u / 5;
      int tap = kd * 25 + tap2;
      if (flip) tap = 124 - tap;
      const int ia = swap ? n : c, ib = swap ? c : n;
      v = w[((long)ia * B + ib) * 125 + tap] * sw;
    }
    const _Float16 h = (_Float16)v;
    out[(((m * 2 + nt) * 2 + 0) * 64 + lane) * 8 + e] = __builtin_bit_cast(unsigned short, h);
    out[(((m * 2 + nt) * 2 + 1) * 64 + lane) * 8 + e] = __builtin_bit_cast(unsigned short, (_Float16)(v - (float)h));
  }
}

__global__ void __launch_bounds__(256, 3)
conv_tk_h2_k(TKH2Args a) {
  constexpr int TH = 16, TW = 16, HH = TH + 4, HW = TW + 4, NV = HH * HW;  // 20 x 20 = 400 voxels per plane
  constexpr int RING = 8, SLOT = 2 * NV;                                    // 8-byte units per ring slot: [piece][voxel]
  __shared__ uint2 lds[RING * SLOT + 2];                                    // + one zero unit for the padding lanes
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 15, lk = lane >> 4;
  int t = xcd_remap_tk(blockIdx.x, a.nblk);
  const int twi = t % a.tiles_w;
  t /= a.tiles_w;
  const int thi = t % a.tiles_h;
  t /= a.tiles_h;
  const int seg = t % a.segs;
  const int n = t / a.segs;
  const int h0 = thi * TH, w0 = twi * TW;
  const int d_begin = seg * a.seg_len;
  const int d_end = min(a.D, d_begin + a.seg_len);
  const float sx = wbf_scale_of(a.x_amax);
  const float osc = 1.f / (sx * wbf_scale_of(a.w_amax));
  constexpr int ZERO = RING * SLOT;
  if (tid == 0) lds[ZERO] = make_uint2(0u, 0u);

  // staging: a thread owns up to two voxels of the plane tile
  long st_off[2];
  int st_v[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int v = tid + 256 * j;
    const int hh = v / HW, ww = v % HW;
    const int gh = h0 - 2 + hh, gw = w0 - 2 + ww;
    const bool inb = v < NV && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
    st_off[j] = inb ? ((((long)n * a.D) * a.H + gh) * a.W + gw) * a.sld : -1;
    st_v[j] = v < NV ? v : -1;
  }
  const long plane = (long)a.H * a.W * a.sld;
  // a plane is staged in two halves: its loads are issued before the MFMAs of the plane in front of it, the conversion and the LDS
  // stores follow them (loaded and stored in one piece the global round trip stood in front of every plane: tools/isa_scan.py)
  auto stage_load = [&](int p, float (&c)[2][4]) {  // plane p (may lie outside the volume: zeros)
    const bool live = p >= 0 && p < a.D;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) c[j][q] = 0.f;
      if (st_v[j] >= 0 && live && st_off[j] >= 0) {
        const float* sp = a.src + st_off[j] + (long)p * plane;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q < a.CK) c[j][q] = sp[q];
      }
    }
  };
  auto stage_store = [&](int p, const float (&c)[2][4]) {  // -> ring slot p & 7
    uint2* sl = lds + (p & (RING - 1)) * SLOT;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (st_v[j] < 0) continue;
      uint2 hv, lv;
      wbf_split2h_pair(c[j][0] * sx, c[j][1] * sx, hv.x, lv.x);
      wbf_split2h_pair(c[j][2] * sx, c[j][3] * sx, hv.y, lv.y);
      sl[st_v[j]] = hv;
      sl[NV + st_v[j]] = lv;
    }
  };

  f32x4 acc[4][2];  // [row of the wave][output-channel tile]
  const int steps = (d_end - d_begin) + 4;
  float pre[2][4];
  // planes d_begin-2 .. d_begin+2 first, then one new plane (d + 3: a sixth slot of the ring of eight) per output plane
#pragma unroll 1
  for (int s = 0; s < 5; ++s) {
    stage_load(d_begin - 2 + s, pre);
    stage_store(d_begin - 2 + s, pre);
  }
  __syncthreads();
#pragma unroll 1
  for (int s = 4; s < steps; ++s) {
    const int d = d_begin + s - 4;  // output plane: needs planes d-2 .. d+2
    stage_load(d + 3, pre);         // in flight during this plane's MFMAs
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int dring = (d - 2) & (RING - 1);
#pragma unroll 2
    for (int m = 0; m < kTKSteps; ++m) {
      uint4 wf[2][2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) wf[nt][pc] = a.wb[((m * 2 + nt) * 2 + pc) * 64 + lane];
      // LDS unit index of this lane's two units (u = 8 m + 2 lk, + 1) for row 0: ring slot of plane d - 2 + kd, voxel
      // (4 wave + kh, li + kw); units 125.. read the zero unit
      int b0, b1, st0, st1, lo0, lo1;
      {
        const int u0 = 8 * m + 2 * lk, u1 = u0 + 1;
        const int t0 = u0 / 5, t1 = u1 / 5;
        const int k0 = u0 - 5 * t0, k1 = u1 - 5 * t1;
        const int v0 = (4 * wave + t0 / 5) * HW + li + t0 % 5, v1 = (4 * wave + t1 / 5) * HW + li + t1 % 5;
        const bool z0 = u0 >= 125, z1 = u1 >= 125;
        b0 = z0 ? ZERO : ((dring + k0) & (RING - 1)) * SLOT + v0;
        b1 = z1 ? ZERO : ((dring + k1) & (RING - 1)) * SLOT + v1;
        st0 = z0 ? 0 : HW; st1 = z1 ? 0 : HW; lo0 = z0 ? 0 : NV; lo1 = z1 ? 0 : NV;
      }
      uint4 xh[4], xl[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint2 h0v = lds[b0 + r * st0], h1v = lds[b1 + r * st1];
        const uint2 l0v = lds[b0 + lo0 + r * st0], l1v = lds[b1 + lo1 + r * st1];
        xh[r] = make_uint4(h0v.x, h0v.y, h1v.x, h1v.y);
        xl[r] = make_uint4(l0v.x, l0v.y, l1v.x, l1v.y);
      }
      // the three piece products as three sweeps over the eight accumulators: consecutive MFMAs never share one
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(tk_f16x8, wf[nt][1]), __builtin_bit_cast(tk_f16x8, xh[r]), acc[r][nt], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(tk_f16x8, wf[nt][0]), __builtin_bit_cast(tk_f16x8, xl[r]), acc[r][nt], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(tk_f16x8, wf[nt][0]), __builtin_bit_cast(tk_f16x8, xh[r]), acc[r][nt], 0, 0, 0);
    }
    // D[row = output channel 16 nt + 4 lk + e][col = position li]
    if (d < d_end) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gh = h0 + 4 * wave + r, gw = w0 + li;
        if (gh < a.H && gw < a.W) {
          float* o = a.dst + ((((long)n * a.D + d) * a.H + gh) * a.W + gw) * a.dld + 4 * lk;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            float4 v = make_float4(acc[r][nt][0] * osc, acc[r][nt][1] * osc, acc[r][nt][2] * osc, acc[r][nt][3] * osc);
            float4* op = reinterpret_cast<float4*>(o + 16 * nt);
            if (a.accumulate) {
              const float4 e = *op;
              v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
            }
            *op = v;
          }
        }
      }
    }
    stage_store(d + 3, pre);   // slot (d + 3) & 7 held plane d - 5: its last readers passed two barriers ago
    __syncthreads();
  }
}

}  // namespace

int msk_gconv_tk_h2(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap) {
  if (ctx->conv_split != 2 || ctx->conv_impl == 25) return 0;  // 25 = A/B: the fp32-MFMA tight-K kernel
  if (!(g.kd == 5 && g.kh == 5 && g.kw == 5 && g.sd == 1 && g.sh == 1 && g.sw == 1 && g.pd == 2 && g.ph == 2 && g.pw == 2)) return 0;
  if (!(g.SD == g.DD && g.SH == g.DH && g.SW == g.DW)) return 0;
  if (g.CK < 1 || g.CK > 4 || g.CN != 32 || g.bias || g.prelu) return 0;
  if (g.DW < 12 || g.DH < 8 || g.DD < 4) return 0;
  if (g.dld % 4 || (((uintptr_t)g.dst) & 15)) return 0;
  unsigned short* wb = (unsigned short*)msk_workspace2(ctx, (size_t)kTKSteps * 2 * 2 * 64 * 8 * sizeof(unsigned short));
  if (!wb) return -1;
  const float* x_amax = g.in_amax ? g.in_amax : msk_absmax(ctx, g.src, g.sld, g.CK, (long)g.N * g.SD * g.SH * g.SW);
  const float* w_amax = msk_absmax(ctx, w_canon, 4, 4, (125L * g.CK * g.CN + 3) / 4);
  if (!x_amax || !w_amax) return -1;
  {
    msk_launch_scope ls(ctx, "pack_weights_tightk");
    hipLaunchKernelGGL(pack_tkh2_weights_k, dim3(100), dim3(256), 0, ctx->stream, w_canon, A, B, swap, g.transposed ? 1 : 0, g.CK, w_amax, wb);
    MSK_LAUNCH_CHECK(ctx);
  }
  TKH2Args a{};
  a.src = g.src; a.sld = g.sld; a.dst = g.dst; a.dld = g.dld;
  a.N = g.N; a.D = g.DD; a.H = g.DH; a.W = g.DW; a.CK = g.CK;
  a.wb = (const uint4*)wb; a.accumulate = g.accumulate;
  a.tiles_h = msk_cdiv(a.H, 16); a.tiles_w = msk_cdiv(a.W, 16);
  const long cols = (long)a.N * a.tiles_h * a.tiles_w;
  const long per_cu = ctx->foldn_wgs > 0 ? ctx->foldn_wgs : 3;  // tuning: option "foldn_wgs" (shared with conv_foldn_k)
  int segs = (int)((per_cu * ctx->num_cu + cols - 1) / cols);
  if (segs > a.D / 8) segs = a.D / 8;
  if (segs < 1) segs = 1;
  a.seg_len = msk_cdiv(a.D, segs);
  a.segs = msk_cdiv(a.D, a.seg_len);
  const long nblk = cols * a.segs;
  if (nblk > 0x7fffffff) return 0;
  a.nblk = (int)nblk;
  a.x_amax = x_amax; a.w_amax = w_amax;
  const char* tag = "conv_tk_h2";
  if (ctx->prof && ctx->prof_shapes) {
    char buf[160];
    snprintf(buf, sizeof(buf), "conv_tk_h2[ck=%d,cn=%d,n=%d,dhw=%dx%dx%d,acc=%d]", g.CK, g.CN, g.N, g.DD, g.DH, g.DW, g.accumulate);
    tag = msk_intern_tag(ctx, buf);
  }
  msk_launch_scope ls(ctx, tag);
  hipLaunchKernelGGL(conv_tk_h2_k, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, a);
  MSK_LAUNCH_CHECK(ctx);
  return 1;
}

int msk_gconv_halo_tightk(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap) {
  const int ks = g.kd;
  if (!(g.kd == g.kh && g.kh == g.kw && ks == 5)) return 0;
  if (!(g.sd == 1 && g.sh == 1 && g.sw == 1 && g.pd == 2 && g.ph == 2 && g.pw == 2)) return 0;
  if (!(g.SD == g.DD && g.SH == g.DH && g.SW == g.DW)) return 0;
  // CK = 1 (in_tr.conv1, N = 16 half-padded) measured 0.39 ms here vs 0.35 ms on the VALU kernel; a tiny CN as well
  // belongs to the VALU kernel
  if (g.CK < 2 || g.CK > 4 || g.CN < 8) return 0;
  const int taps = ks * ks * ks, HT = (taps + 1) / 2;
  const int npad = ((g.CN + 31) / 32) * 32;
  float* wt = (float*)msk_workspace2(ctx, (size_t)HT * g.CK * 2 * npad * sizeof(float));
  if (!wt) return -1;
  {
    msk_launch_scope ls(ctx, "pack_weights_tightk");
    hipLaunchKernelGGL(pack_tightk_weights_k, dim3(64), dim3(256), 0, ctx->stream, w_canon, A, B, ks, swap,
                       g.transposed ? 1 : 0, g.CK, g.CN, npad, wt);
    MSK_LAUNCH_CHECK(ctx);
  }
  TKArgs a{};
  a.src = g.src; a.sld = g.sld; a.dst = g.dst; a.dld = g.dld;
  a.N = g.N; a.D = g.DD; a.H = g.DH; a.W = g.DW;
  a.CN = g.CN; a.npad = npad; a.wt = wt; a.bias = g.bias; a.accumulate = g.accumulate;
  a.wt_bytes = (unsigned)((size_t)HT * g.CK * 2 * npad * sizeof(float));
  {
    const size_t sb = (size_t)g.N * g.SD * g.SH * g.SW * g.sld * sizeof(float);
    if (sb >= 0xFFFFFFF0ull) return 0;
    a.src_bytes = (unsigned)sb;
  }
  const char* tag = "conv_halo_tightk";
  if (ctx->prof && ctx->prof_shapes) {
    char buf[160];
    snprintf(buf, sizeof(buf), "conv_halo_tightk[ck=%d,cn=%d,n=%d,dhw=%dx%dx%d,acc=%d]", g.CK, g.CN, g.N, g.DD, g.DH, g.DW,
             g.accumulate);
    tag = msk_intern_tag(ctx, buf);
  }
  int rc;
  if (g.DW >= 32 && g.DW % 32 < 1) rc = launch_tk<2, 4, 32, 5>(ctx, a, g.CK, npad / 32, tag);
  else if (g.DW >= 16 && g.DW % 16 < 1) rc = launch_tk<2, 8, 16, 5>(ctx, a, g.CK, npad / 32, tag);
  else rc = launch_tk<4, 8, 8, 5>(ctx, a, g.CK, npad / 32, tag);
  return rc == 0 ? 1 : rc;
}
