// MFMA (fp32-input v_mfma_f32_32x32x2_f32) convolution kernels for gfx950.
//
// (1) conv_halo_mfma_k -- 'same' stride-1 odd-kernel convolutions (the 5x5x5 LUConv
//     layers = 99.8% of VNet FLOPs; also their data gradients, which are the same
//     convolution with flipped/transposed weights).  Implicit GEMM
//        M = output voxels (32 per MFMA tile), N = out channels (32), K = taps x Cin.
//     Per workgroup: a TDxTHxTW output tile; for each 8-channel K-chunk the input halo
//     tile is staged ONCE in LDS as [quad][voxel][4] (conflict-free ds_read_b128 with
//     an immediate per-tap offset) and reused by all k^3 taps.  Weights are pre-packed
//     so a wavefront's B fragment is one contiguous 1 KiB global load (L2 resident,
//     identical for every workgroup), double-buffered one tap-row ahead.
//     K order inside a chunk is (h, q): lane (i, h) feeds channel 8kc+4h+q to the q-th
//     MFMA -- A and B agree, and any K permutation is legal in a GEMM sum.
//
// (2) wgrad_mfma_k -- weight gradients: M = Cin tile (32), N = Cout tile (32),
//     K = voxels.  Both operands are one coalesced dword per lane straight from global
//     (lanes run along channels in NDHWC), the dy fragment is reused across the kw taps
//     of a tap-row, accumulators (kw x 16 VGPRs) stay in registers for the whole voxel
//     range, split-K partials are reduced deterministically by wgrad_reduce_k.
#include "msk_conv.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct HaloArgs {
  const float* src;
  int sld;
  float* dst;
  int dld;
  int N, D, H, W;
  int CK, CN;
  const float4* wm;
  int KC, npad;
  const float* bias;
  int accumulate;
  int tiles_d, tiles_h, tiles_w;
  int nblk;
  int vec;
  // deterministic split-K over the 8-channel chunks (small-M layers: 256ch @ 8^3 has only 32
  // (M,N) tiles for 256 CUs): blockIdx.z owns chunks [z*kc_per, (z+1)*kc_per) and writes an
  // fp32 slab partial[z][voxel][CN]; splitk_reduce_k adds the slabs in fixed order.
  int ksplit, kc_per;
  float* partial;
};

__device__ __forceinline__ int xcd_remap(int bid, int nb) {
  // bijective "block b runs on XCD b%8" -> contiguous chunk per XCD (guide T1)
  const int q = nb >> 3, r = nb & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int TD, int TH, int TW, int KS>
__global__ void __launch_bounds__(256, 2) conv_halo_mfma_k(HaloArgs a) {
  constexpr int P = KS / 2;
  constexpr int HD = TD + 2 * P, HH = TH + 2 * P, HW = TW + 2 * P;
  constexpr int NV = HD * HH * HW;
  constexpr int NVP = NV | 1;  // odd plane pitch
  constexpr int MR = TD * TH * TW / 128;
  static_assert(TD * TH * TW % 128 == 0, "tile must hold 4 waves x MR x 32 voxels");
  __shared__ float4 lds[2 * NVP];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  int tile = xcd_remap(blockIdx.x, a.nblk);
  const int twi = tile % a.tiles_w;
  tile /= a.tiles_w;
  const int thi = tile % a.tiles_h;
  tile /= a.tiles_h;
  const int tdi = tile % a.tiles_d;
  const int n = tile / a.tiles_d;
  const int d0 = tdi * TD, h0 = thi * TH, w0 = twi * TW;
  const int nt = blockIdx.y;

  int abase[MR];
#pragma unroll
  for (int r = 0; r < MR; ++r) {
    const int l = (wave * MR + r) * 32 + li;
    const int dz = l / (TH * TW), hy = (l / TW) % TH, wx = l % TW;
    abase[r] = (dz * HH + hy) * HW + wx + lh * NVP;
  }
  f32x16 acc[MR];
#pragma unroll
  for (int r = 0; r < MR; ++r)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[r][j] = 0.f;

  const long tapstride = (long)a.KC * 2 * a.npad;  // float4 units
  const float4* wlane = a.wm + ((long)lh * a.npad + nt * 32 + li);

  const int kc_begin = a.ksplit > 1 ? (int)blockIdx.z * a.kc_per : 0;
  const int kc_end = a.ksplit > 1 ? min(a.KC, kc_begin + a.kc_per) : a.KC;
  for (int kc = kc_begin; kc < kc_end; ++kc) {
    __syncthreads();
    // ---- stage the halo tile for channels [8kc, 8kc+8): groups of 7 independent float4 loads in
    // flight per thread (a plain load->store loop serialises ~14 global round trips per chunk) ----
    constexpr int SG = 7;
    for (int base = 0; base < NV * 2; base += SG * 256) {
      float4 tmp[SG];
#pragma unroll
      for (int i = 0; i < SG; ++i) {
        const int it = base + tid + i * 256;
        const int hv = it >> 1, q = it & 1;
        const int hd = hv / (HH * HW), rem = hv % (HH * HW), hh = rem / HW, hw = rem % HW;
        const int gd = d0 - P + hd, gh = h0 - P + hh, gw = w0 - P + hw;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c0 = kc * 8 + q * 4;
        if (it < NV * 2 && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W && c0 < a.CK) {
          const float* p = a.src + ((((long)n * a.D + gd) * a.H + gh) * a.W + gw) * a.sld + c0;
          if (a.vec) {
            v = *reinterpret_cast<const float4*>(p);
          } else {
            v.x = p[0];
            if (c0 + 1 < a.CK) v.y = p[1];
            if (c0 + 2 < a.CK) v.z = p[2];
            if (c0 + 3 < a.CK) v.w = p[3];
          }
        }
        tmp[i] = v;
      }
#pragma unroll
      for (int i = 0; i < SG; ++i) {
        const int it = base + tid + i * 256;
        if (it < NV * 2) lds[(it & 1) * NVP + (it >> 1)] = tmp[i];
      }
    }
    __syncthreads();

    const float4* wk = wlane + (long)kc * 2 * a.npad;
    float4 bcur[KS], bnxt[KS];
#pragma unroll
    for (int kw = 0; kw < KS; ++kw) bcur[kw] = wk[(long)kw * tapstride];
    for (int rr = 0; rr < KS * KS; ++rr) {
      if (rr + 1 < KS * KS) {
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) bnxt[kw] = wk[(long)((rr + 1) * KS + kw) * tapstride];
      }
      const int kd = rr / KS, kh = rr % KS;
      const int rowoff = (kd * HH + kh) * HW;
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) {
        float4 av[MR];
#pragma unroll
        for (int r = 0; r < MR; ++r) av[r] = lds[abase[r] + rowoff + kw];
        const float4 b = bcur[kw];
#pragma unroll
        for (int r = 0; r < MR; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r].x, b.x, acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < MR; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r].y, b.y, acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < MR; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r].z, b.z, acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < MR; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r].w, b.w, acc[r], 0, 0, 0);
      }
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) bcur[kw] = bnxt[kw];
    }
  }

  // ---- epilogue: D[row = voxel][col = out channel] ----
  const int co = nt * 32 + li;
  if (co < a.CN) {
    const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
    for (int r = 0; r < MR; ++r) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = (j & 3) + 8 * (j >> 2) + 4 * lh;
        const int l = (wave * MR + r) * 32 + row;
        const int dz = l / (TH * TW), hy = (l / TW) % TH, wx = l % TW;
        const int gd = d0 + dz, gh = h0 + hy, gw = w0 + wx;
        if (gd < a.D && gh < a.H && gw < a.W) {
          const long vox = (((long)n * a.D + gd) * a.H + gh) * a.W + gw;
          if (a.ksplit > 1) {
            a.partial[((long)blockIdx.z * ((long)a.N * a.D * a.H * a.W) + vox) * a.CN + co] = acc[r][j];
          } else {
            float* o = a.dst + vox * a.dld + co;
            float v = acc[r][j] + bv;
            if (a.accumulate) v += *o;
            *o = v;
          }
        }
      }
    }
  }
}

__global__ void __launch_bounds__(256)
splitk_reduce_k(const float* __restrict__ partial, int ksplit, long voxels, int CN, const float* __restrict__ bias,
                float* __restrict__ dst, int dld, int accumulate) {
  const long total = voxels * CN;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long v = i / CN;
    const int c = (int)(i - v * CN);
    float s = bias ? bias[c] : 0.f;
    for (int z = 0; z < ksplit; ++z) s += partial[(long)z * total + i];  // fixed order
    float* o = dst + v * dld + c;
    *o = accumulate ? *o + s : s;
  }
}

// Same halo staging, VALU inner loop: for layers with a tiny channel count on one side
// (in_tr 1->16, out_tr 32->ncls and its data gradient ncls->32) a 32x32 MFMA tile would be
// mostly padding (10x waste at ncls=3).  fp32 VALU FMA has the same 64 FLOP/clk/SIMD peak as
// the fp32 MFMA, so one thread per output voxel with CN accumulators, the input quad from LDS
// (conflict-free ds_read_b128) and the weights as wave-uniform scalar loads runs these layers
// at their useful FLOP count.
template <int TD, int TH, int TW, int KS, int CK, int CN>
__global__ void __launch_bounds__(256) conv_halo_valu_k(HaloArgs a, const float* __restrict__ wp /*[tap][CK][CN]*/) {
  constexpr int P = KS / 2;
  constexpr int HD = TD + 2 * P, HH = TH + 2 * P, HW = TW + 2 * P;
  constexpr int NV = HD * HH * HW;
  // ONE channel quad per staged chunk: 27.7 KiB of LDS per workgroup -> 5 workgroups per CU, so the
  // scalar-load latency of the wave-uniform weights is covered by occupancy (with 8-channel chunks
  // and 2 workgroups per CU the kernel ran 5.6x off its VALU bound)
  constexpr int QC = (CK + 3) / 4;
  static_assert(TD * TH * TW == 256, "one thread per output voxel");
  __shared__ float4 lds[NV];

  const int tid = threadIdx.x;
  int tile = xcd_remap(blockIdx.x, a.nblk);
  const int twi = tile % a.tiles_w;
  tile /= a.tiles_w;
  const int thi = tile % a.tiles_h;
  tile /= a.tiles_h;
  const int tdi = tile % a.tiles_d;
  const int n = tile / a.tiles_d;
  const int d0 = tdi * TD, h0 = thi * TH, w0 = twi * TW;
  const int dz = tid / (TH * TW), hy = (tid / TW) % TH, wx = tid % TW;
  const int base = (dz * HH + hy) * HW + wx;

  float acc[CN];
#pragma unroll
  for (int j = 0; j < CN; ++j) acc[j] = 0.f;

#pragma unroll 1
  for (int qc = 0; qc < QC; ++qc) {
    const int c0 = qc * 4;
    __syncthreads();
    constexpr int SG = 7;
    for (int sb = 0; sb < NV; sb += SG * 256) {
      float4 tmp[SG];
#pragma unroll
      for (int i = 0; i < SG; ++i) {
        const int hv = sb + tid + i * 256;
        const int hd = hv / (HH * HW), rem = hv % (HH * HW), hh = rem / HW, hw = rem % HW;
        const int gd = d0 - P + hd, gh = h0 - P + hh, gw = w0 - P + hw;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (hv < NV && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W) {
          const float* p = a.src + ((((long)n * a.D + gd) * a.H + gh) * a.W + gw) * a.sld + c0;
          if (a.vec) {
            v = *reinterpret_cast<const float4*>(p);
          } else {
            v.x = p[0];
            if (c0 + 1 < CK) v.y = p[1];
            if (c0 + 2 < CK) v.z = p[2];
            if (c0 + 3 < CK) v.w = p[3];
          }
        }
        tmp[i] = v;
      }
#pragma unroll
      for (int i = 0; i < SG; ++i) {
        const int hv = sb + tid + i * 256;
        if (hv < NV) lds[hv] = tmp[i];
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int rr = 0; rr < KS * KS; ++rr) {
      const int rowoff = ((rr / KS) * HH + (rr % KS)) * HW;
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) {
        const float4 av = lds[base + rowoff + kw];
        const float* w = wp + ((long)(rr * KS + kw) * CK + c0) * CN;  // wave-uniform -> scalar loads
        const float xs[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (c0 + c < CK) {
#pragma unroll
            for (int j = 0; j < CN; ++j) acc[j] = fmaf(xs[c], w[c * CN + j], acc[j]);
          }
        }
      }
    }
  }

  const int gd = d0 + dz, gh = h0 + hy, gw = w0 + wx;
  if (gd < a.D && gh < a.H && gw < a.W) {
    float* o = a.dst + ((((long)n * a.D + gd) * a.H + gh) * a.W + gw) * a.dld;
#pragma unroll
    for (int j = 0; j < CN; ++j) {
      float v = acc[j] + (a.bias ? a.bias[j] : 0.f);
      if (a.accumulate) v += o[j];
      o[j] = v;
    }
  }
}

template <int KS, int CK, int CN>
int launch_halo_valu(msk_ctx* ctx, HaloArgs& a, const float* wp) {
  // one thread per voxel: tiles of 256 voxels
  const int TW = a.W >= 32 ? 32 : (a.W >= 16 ? 16 : 8);
  const int TD = TW == 8 ? 4 : 2, TH = TW == 32 ? 4 : 8;
  a.tiles_d = msk_cdiv(a.D, TD);
  a.tiles_h = msk_cdiv(a.H, TH);
  a.tiles_w = msk_cdiv(a.W, TW);
  const long nblk = (long)a.N * a.tiles_d * a.tiles_h * a.tiles_w;
  if (nblk > 0x7fffffff) return msk_fail(ctx, __FILE__, __LINE__, "conv_halo_valu", "grid too large");
  a.nblk = (int)nblk;
  const char* tag = "conv_halo_valu";
  if (ctx->prof && ctx->prof_shapes) {
    char buf[160];
    snprintf(buf, sizeof(buf), "conv_halo_valu[ck=%d,cn=%d,n=%d,dhw=%dx%dx%d,acc=%d]", CK, CN, a.N, a.D, a.H, a.W, a.accumulate);
    tag = msk_intern_tag(ctx, buf);
  }
  msk_launch_scope ls(ctx, tag);
  if (TW == 32)
    hipLaunchKernelGGL((conv_halo_valu_k<2, 4, 32, KS, CK, CN>), dim3((unsigned)nblk), dim3(256), 0, ctx->stream, a, wp);
  else if (TW == 16)
    hipLaunchKernelGGL((conv_halo_valu_k<2, 8, 16, KS, CK, CN>), dim3((unsigned)nblk), dim3(256), 0, ctx->stream, a, wp);
  else
    hipLaunchKernelGGL((conv_halo_valu_k<4, 8, 8, KS, CK, CN>), dim3((unsigned)nblk), dim3(256), 0, ctx->stream, a, wp);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

template <int TD, int TH, int TW, int KS>
int launch_halo(msk_ctx* ctx, HaloArgs& a, int ntiles_n) {
  a.tiles_d = msk_cdiv(a.D, TD);
  a.tiles_h = msk_cdiv(a.H, TH);
  a.tiles_w = msk_cdiv(a.W, TW);
  const long nblk = (long)a.N * a.tiles_d * a.tiles_h * a.tiles_w;
  if (nblk > 0x7fffffff) return msk_fail(ctx, __FILE__, __LINE__, "conv_halo", "grid too large");
  a.nblk = (int)nblk;
  const char* tag = "conv_halo_mfma_k";
  if (ctx->prof) {  // same spelling as the demangled kernel name in rocprofv3's kernel stats
    char buf[200];
    int n = snprintf(buf, sizeof(buf), "conv_halo_mfma_k<%d, %d, %d, %d>", TD, TH, TW, KS);
    if (ctx->prof_shapes)
      snprintf(buf + n, sizeof(buf) - n, "[ck=%d,cn=%d,n=%d,dhw=%dx%dx%d,acc=%d]", a.CK, a.CN, a.N, a.D, a.H, a.W, a.accumulate);
    tag = msk_intern_tag(ctx, buf);
  }
  // split K when the (M, N) tiling alone cannot fill the chip (~4 workgroups per CU wanted)
  a.ksplit = 1;
  a.kc_per = a.KC;
  a.partial = nullptr;
  const long mn_blocks = nblk * ntiles_n;
  if (mn_blocks < 2L * ctx->num_cu && a.KC >= 2) {
    long want = (4L * ctx->num_cu + mn_blocks - 1) / mn_blocks;
    if (want > a.KC) want = a.KC;
    a.kc_per = (int)((a.KC + want - 1) / want);
    a.ksplit = (a.KC + a.kc_per - 1) / a.kc_per;
  }
  const long voxels = (long)a.N * a.D * a.H * a.W;
  if (a.ksplit > 1) {
    a.partial = (float*)msk_workspace(ctx, (size_t)a.ksplit * voxels * a.CN * sizeof(float));
    if (!a.partial) return -1;
  }
  {
    msk_launch_scope ls(ctx, tag);
    hipLaunchKernelGGL((conv_halo_mfma_k<TD, TH, TW, KS>), dim3((unsigned)nblk, ntiles_n, a.ksplit), dim3(256), 0,
                       ctx->stream, a);
    MSK_LAUNCH_CHECK(ctx);
  }
  if (a.ksplit > 1) {
    msk_launch_scope ls(ctx, "conv_splitk_reduce");
    long blocks = (voxels * a.CN + 255) / 256;
    if (blocks > 8L * ctx->num_cu) blocks = 8L * ctx->num_cu;
    hipLaunchKernelGGL(splitk_reduce_k, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const float*)a.partial,
                       a.ksplit, voxels, a.CN, a.bias, a.dst, a.dld, a.accumulate);
    MSK_LAUNCH_CHECK(ctx);
  }
  return 0;
}

// ---------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------
// All operand loads are raw buffer loads: a 32-bit byte offset per lane and the hardware
// range check (offset >= num_records -> 0) replace every bounds branch, so the K loop is
// straight-line code: offsets -> loads (next step) -> MFMAs (this step).
constexpr unsigned kOOB = 0xFFFFFFF0u;

__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

struct VoxCursor {  // (n, d, h, w) of a linear voxel index, advanced without divisions
  int n, d, h, w;
  __device__ __forceinline__ void init(long m, int D, int H, int W) {
    w = (int)(m % W);
    h = (int)((m / W) % H);
    d = (int)((m / ((long)W * H)) % D);
    n = (int)(m / ((long)W * H * D));
  }
  __device__ __forceinline__ void wrap(int D, int H, int W) {
    if (w >= W) {
      w -= W;
      if (++h >= H) {
        h = 0;
        if (++d >= D) {
          d = 0;
          ++n;
        }
      }
    }
  }
  __device__ __forceinline__ void advance2(int D, int H, int W) {
    w += 2;
    wrap(D, H, W);
    wrap(D, H, W);  // W == 1 needs two carries
  }
};

template <int KW, int U>
__global__ void __launch_bounds__(256, 2)
wgrad_mfma_k(WGrad g, int splits, float* __restrict__ partial, unsigned a_bytes, unsigned b_bytes) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int ca_tiles = (g.CA + 31) >> 5, cb_tiles = (g.CB + 31) >> 5;
  int b = blockIdx.x;
  const int cbt = b % cb_tiles;
  b /= cb_tiles;
  const int cat = b % ca_tiles;
  const int rr = b / ca_tiles;  // tap row = kd*kh_count + kh
  const int kd = rr / g.kh, kh = rr % g.kh;
  const int split = blockIdx.y * 4 + wave;
  if (split >= splits) return;

  const long M = (long)g.N * g.BD * g.BH * g.BW;
  long per = (M + splits - 1) / splits;
  per = (per + 1) & ~1L;  // even, so lane halves stay in step
  const long m0 = (long)split * per;
  long m1 = m0 + per;
  if (m1 > M) m1 = M;

  const int ca = cat * 32 + li, cb = cbt * 32 + li;
  const bool ca_ok = ca < g.CA, cb_ok = cb < g.CB;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, b_bytes, 0x00020000);

  f32x16 acc[KW];
#pragma unroll
  for (int k = 0; k < KW; ++k)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;

  VoxCursor c;
  long m = m0 + lh;
  c.init(m, g.BD, g.BH, g.BW);

  // Batches of U steps: all (KW+1)*U loads of the NEXT batch are issued before the KW*U MFMAs
  // of the current one, so each wave has U*KW*64 cycles of matrix work per memory round trip
  // (with U = 1 the loop ran at min(1, 4 waves * 320 cycles / load latency) ~ 45% of peak).
  float av_n[U][KW], bv_n[U];
  auto load_batch = [&](long mm) {  // loads steps mm, mm+2, ...; leaves the cursor after the batch
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long mu = mm + 2 * u;
      const bool live = mu < m1;
      const int id = c.d * g.sd - g.pd + kd, ih = c.h * g.sh - g.ph + kh;
      const bool rowok = live && ca_ok && (unsigned)id < (unsigned)g.AD && (unsigned)ih < (unsigned)g.AH;
      const int rowbase = ((c.n * g.AD + id) * g.AH + ih) * g.AW;  // voxel index of (n, id, ih, 0)
      const int iw0 = c.w * g.sw - g.pw;
#pragma unroll
      for (int k = 0; k < KW; ++k) {
        const int iw = iw0 + k;
        const unsigned off = ((unsigned)(rowbase + iw) * (unsigned)g.ald + (unsigned)ca) * 4u;
        av_n[u][k] = buf_load(ra, (rowok && (unsigned)iw < (unsigned)g.AW) ? off : kOOB);
      }
      const unsigned offb = ((unsigned)mu * (unsigned)g.bld + (unsigned)cb) * 4u;
      bv_n[u] = buf_load(rb, (live && cb_ok) ? offb : kOOB);
      c.advance2(g.BD, g.BH, g.BW);
    }
  };
  load_batch(m);
  for (; m - lh < m1; m += 2 * U) {  // uniform trip count across the wave
    float av[U][KW], bv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int k = 0; k < KW; ++k) av[u][k] = av_n[u][k];
      bv[u] = bv_n[u];
    }
    load_batch(m + 2 * U);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < KW; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][k], bv[u], acc[k], 0, 0, 0);
  }

  const int taps = g.kd * g.kh * g.kw;
  if (cb_ok) {
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      const int tap = rr * g.kw + k;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = (j & 3) + 8 * (j >> 2) + 4 * lh;
        const int oca = cat * 32 + row;
        if (oca < g.CA) partial[(((long)split * taps + tap) * g.CA + oca) * g.CB + cb] = acc[k][j];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// General gather convolution on MFMA, operands straight from global memory.
// Used for the strided 2x2x2 (or anisotropic) down/up convolutions and the 1x1x1 head:
// HBM-bound layers (arithmetic intensity ~13-20 flop/B) where an LDS halo buys nothing --
// every input voxel is used by exactly one output voxel when k == s.
//   forward mode    spos = dpos*s - p + k
//   transposed mode spos = (dpos + p - k)/s when divisible: dst voxels are tiled per PARITY
//                   CLASS (dpos mod s), so inside a tile the set of contributing taps is
//                   uniform and no MFMA work is spent on masked taps.
// M = 32 dst voxels per wave, N = NR x 32 output channels in registers, K = taps x channels
// in chunks of 8 with the same packed weights as the halo kernel.
template <int NR>
__global__ void __launch_bounds__(256, NR <= 2 ? 4 : (NR <= 4 ? 3 : 2))  // latency bound: as many wavefronts as the accumulators allow
gconv_gather_mfma_k(GConv g, const float4* __restrict__ wm, int KC, int npad, unsigned src_bytes, int vec,
                    int tiles_max) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  // parity class of this block (forward mode: a single class with "stride" 1)
  const int cs_d = g.transposed ? g.sd : 1, cs_h = g.transposed ? g.sh : 1, cs_w = g.transposed ? g.sw : 1;
  int cls = blockIdx.y;
  const int rw = cls % cs_w;
  cls /= cs_w;
  const int rh = cls % cs_h;
  const int rd = cls / cs_h;
  // coarse grid of this class: dpos = q*cs + r
  const int QD = (g.DD - rd + cs_d - 1) / cs_d, QH = (g.DH - rh + cs_h - 1) / cs_h, QW = (g.DW - rw + cs_w - 1) / cs_w;
  const long Mq = (long)g.N * QD * QH * QW;
  const long m = ((long)blockIdx.x * 4 + wave) * 32 + li;
  if (((long)blockIdx.x * 4 + wave) * 32 >= Mq || QD <= 0 || QH <= 0 || QW <= 0) return;
  const bool mok = m < Mq;
  const long mm = mok ? m : 0;
  const unsigned mmu = (unsigned)mm, mt1 = mmu / (unsigned)QW, mt2 = mt1 / (unsigned)QH;  // Mq < 2^31
  const int qw = (int)(mmu - mt1 * (unsigned)QW), qh = (int)(mt1 - mt2 * (unsigned)QH);
  const int n = (int)(mt2 / (unsigned)QD), qd = (int)(mt2 - (unsigned)n * (unsigned)QD);
  const int od = qd * cs_d + rd, oh = qh * cs_h + rh, ow = qw * cs_w + rw;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g.src, 0, src_bytes, 0x00020000);
  const int nt0 = blockIdx.z * NR;

  f32x16 acc[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[r][j] = 0.f;

  const long tapstride = (long)KC * 2 * npad;
  // packed weights through a raw buffer resource: (tap, chunk, N tile) in the scalar offset, one constant lane offset
  const unsigned wbytes = (unsigned)((long)g.kd * g.kh * g.kw * KC * 2 * npad * 16);
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)wm, 0, wbytes, 0x00020000);
  const unsigned wlane_b = (unsigned)(lh * npad + nt0 * 32 + li) * 16u;
  for (int a = 0; a < g.kd; ++a) {
    int id;
    if (!g.transposed) {
      id = od * g.sd - g.pd + a;
    } else {
      const int t = od + g.pd - a;
      if ((rd + g.pd - a) % g.sd != 0) continue;  // uniform over the class
      id = t >= 0 ? t / g.sd : -1;
    }
    for (int b = 0; b < g.kh; ++b) {
      int ih;
      if (!g.transposed) {
        ih = oh * g.sh - g.ph + b;
      } else {
        const int t = oh + g.ph - b;
        if ((rh + g.ph - b) % g.sh != 0) continue;
        ih = t >= 0 ? t / g.sh : -1;
      }
      for (int c = 0; c < g.kw; ++c) {
        int iw;
        if (!g.transposed) {
          iw = ow * g.sw - g.pw + c;
        } else {
          const int t = ow + g.pw - c;
          if ((rw + g.pw - c) % g.sw != 0) continue;
          iw = t >= 0 ? t / g.sw : -1;
        }
        const bool ok = mok && (unsigned)id < (unsigned)g.SD && (unsigned)ih < (unsigned)g.SH && (unsigned)iw < (unsigned)g.SW;
        const unsigned vbase = (unsigned)(((n * g.SD + id) * g.SH + ih) * g.SW + iw) * (unsigned)g.sld;
        const int tap = (a * g.kh + b) * g.kw + c;
        const unsigned wtap_b = (unsigned)(tap * tapstride * 16);
        // K chunks in batches of KB: every load of the batch is issued before its MFMAs (these
        // layers have only a handful of MFMAs per wave; a load->MFMA chain per chunk was pure latency)
        constexpr int KB = NR >= 4 ? 2 : 4;
        for (int kc0 = 0; kc0 < KC; kc0 += KB) {
          float av[KB][4];
          float4 bv[KB][NR];
#pragma unroll
          for (int u = 0; u < KB; ++u) {
            const int kc = kc0 + u;
            const int c0 = kc * 8 + lh * 4;
            const bool kin = kc < KC;
            if (vec) {  // CK % 4 == 0: one validity test per quad (the compiler merges the 4 dwords)
              const unsigned off = (ok && kin && c0 < g.CK) ? (vbase + (unsigned)c0) * 4u : kOOB;
#pragma unroll
              for (int q = 0; q < 4; ++q) av[u][q] = buf_load(rs, off == kOOB ? kOOB : off + 4u * q);
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q)
                av[u][q] = buf_load(rs, (ok && kin && c0 + q < g.CK) ? (vbase + (unsigned)(c0 + q)) * 4u : kOOB);
            }
#pragma unroll
            for (int r = 0; r < NR; ++r)
              bv[u][r] = kin ? __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                   wres, (int)wlane_b, (int)(wtap_b + (unsigned)(kc * 2 * npad + r * 32) * 16u), 0))
                             : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int u = 0; u < KB; ++u) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
              acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][0], bv[u][r].x, acc[r], 0, 0, 0);
              acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][1], bv[u][r].y, acc[r], 0, 0, 0);
              acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][2], bv[u][r].z, acc[r], 0, 0, 0);
              acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][3], bv[u][r].w, acc[r], 0, 0, 0);
            }
          }
        }
      }
    }
  }

  // epilogue: rows = the tile's 32 dst voxels (decoded per row), cols = output channels
  // The lane's first row is decoded with three 32-bit divisions (Mq < 2^31: tensors are chunked below 4 GiB), the
  // other 15 follow by carries (rows advance by 1, 1, 1, 5).  The 64-bit divisions by run-time values that stood
  // here for every row and N tile cost more instructions than the whole MFMA loop of these k = s layers.
  const long mbase = ((long)blockIdx.x * 4 + wave) * 32;
  long rowoff[16];
  bool rowok[16];
  {
    const long m0 = mbase + 4 * lh;
    const unsigned mu = (unsigned)(m0 < Mq ? m0 : 0);
    const unsigned t1 = mu / (unsigned)QW, t2 = t1 / (unsigned)QH;
    unsigned w_ = mu - t1 * (unsigned)QW, h_ = t1 - t2 * (unsigned)QH;
    unsigned n_ = t2 / (unsigned)QD, d_ = t2 - n_ * (unsigned)QD;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j > 0) {
        w_ += (j & 3) ? 1u : 5u;
        while (w_ >= (unsigned)QW) {
          w_ -= (unsigned)QW;
          if (++h_ >= (unsigned)QH) {
            h_ = 0;
            if (++d_ >= (unsigned)QD) {
              d_ = 0;
              ++n_;
            }
          }
        }
      }
      rowok[j] = mbase + (j & 3) + 8 * (j >> 2) + 4 * lh < Mq;
      rowoff[j] = ((((long)n_ * g.DD + d_ * cs_d + rd) * g.DH + h_ * cs_h + rh) * g.DW + w_ * cs_w + rw) * g.dld;
    }
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int co = (nt0 + r) * 32 + li;
    if (co >= g.CN) continue;
    const float bvs = g.bias ? g.bias[co] : 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (rowok[j]) {
        float* o = g.dst + rowoff[j] + co;
        float v = acc[r][j] + bvs;
        if (g.accumulate) v += *o;
        *o = v;
      }
    }
  }
  (void)tiles_max;
}

const char* wgrad_tag(msk_ctx* ctx, const char* base, const WGrad& g, int splits);

// LDS-staged weight gradient for stride-1 'same' convolutions (all 5x5x5 LUConv layers).
// The global-direct kernel above re-reads every x row 5x per tap-row and thrashes the 32 KiB L1
// (PMC: MFMA pipe 42% busy, TCP pending-stall dominant).  Here a workgroup of KS waves owns one
// kd plane of taps (wave = kh, registers = kw): per chunk of R output rows x WS columns it stages
// the (R+2P) x (WS+2P) x-halo rows and the R x WS dy rows ONCE with coalesced float4 loads, then
// every MFMA operand is a conflict-free ds_read_b32 (32 lanes = 32 consecutive channels).
template <int KS, int R, int WS>
__global__ void __launch_bounds__(256, 3)
wgrad_lds_mfma_k(WGrad g, int splits, int chunks_total, int chunks_per_split, float* __restrict__ partial) {
  constexpr int P = KS / 2;
  constexpr int XR = R + 2 * P, XW = WS + 2 * P;
  constexpr int NT = 256;
  // 4 waves (one per SIMD: odd wave counts leave the doubly-loaded SIMD as the occupancy
  // limiter -- measured 1.2 waves/SIMD with 5-wave workgroups).  The KS*KS taps of the kd plane
  // are dealt round-robin: wave w owns taps w, w+4, ... (6 each for 5x5) and the LAST tap is
  // shared -- each wave accumulates it for a quarter of the voxel pairs and the four partial
  // accumulators are summed through LDS in a fixed order at the end (7/6/6/6 -> 6.25 each).
  static_assert((KS * KS - 1) % 4 == 0, "own taps must divide evenly over 4 waves");
  constexpr int TPW = (KS * KS - 1) / 4;
  constexpr int SHARED_OFF = ((KS - 1) * XW + (KS - 1)) * 32;  // tap (kh, kw) = (KS-1, KS-1)
  __shared__ float xs[XR * XW * 32];
  __shared__ float dys[R * WS * 32];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int ca_tiles = (g.CA + 31) >> 5, cb_tiles = (g.CB + 31) >> 5;
  int b = blockIdx.x;
  const int cbt = b % cb_tiles;
  b /= cb_tiles;
  const int cat = b % ca_tiles;
  const int kd = b / ca_tiles;
  const int split = blockIdx.y;
  const int D = g.BD, H = g.BH, W = g.BW;
  const int hblocks = (H + R - 1) / R, wblocks = (W + WS - 1) / WS;

  int toff[TPW];  // LDS float offset of tap j relative to (row r, col 2p)
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int t = wave + 4 * j;
    toff[j] = ((t / KS) * XW + (t % KS)) * 32;
  }
  f32x16 acc[TPW], acc_sh;
#pragma unroll
  for (int k = 0; k < TPW; ++k)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) acc_sh[j] = 0.f;

  const int c_begin = split * chunks_per_split;
  int c_end = c_begin + chunks_per_split;
  if (c_end > chunks_total) c_end = chunks_total;

  // Staging: groups of 4 float4 loads in flight per thread, written straight to LDS (the 7x16
  // accumulator registers leave no room for a whole-chunk register prefetch without spills;
  // with 3 workgroups per CU another workgroup's MFMAs cover this phase).
  constexpr int XITEMS = XR * XW * 8, DITEMS = R * WS * 8;
  auto next_valid = [&](int ch) {  // first chunk >= ch whose input depth d + kd - P is inside the volume
    while (ch < c_end) {
      const int d = (ch / (wblocks * hblocks)) % D;
      if ((unsigned)(d + kd - P) < (unsigned)D) break;
      ++ch;
    }
    return ch;
  };
  auto stage = [&](int ch) {
    int t = ch;
    const int wb = t % wblocks;
    t /= wblocks;
    const int hb = t % hblocks;
    t /= hblocks;
    const int d = t % D;
    const int n = t / D;
    const int id = d + kd - P;
    const int h0 = hb * R, w0 = wb * WS;
    for (int base = 0; base < XITEMS; base += 4 * NT) {
      float4 tmp[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int it = base + tid + i * NT;
        const int q = it & 7, v = it >> 3;
        const int row = v / XW, col = v % XW;
        const int ih = h0 - P + row, iw = w0 - P + col;
        const int c0 = cat * 32 + q * 4;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (it < XITEMS && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W && c0 < g.CA)
          val = *reinterpret_cast<const float4*>(g.A + ((((long)n * D + id) * H + ih) * W + iw) * g.ald + c0);
        tmp[i] = val;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int it = base + tid + i * NT;
        if (it < XITEMS) *reinterpret_cast<float4*>(&xs[it * 4]) = tmp[i];  // (v*32 + q*4) == it*4
      }
    }
    for (int base = 0; base < DITEMS; base += 4 * NT) {
      float4 tmp[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int it = base + tid + i * NT;
        const int q = it & 7, v = it >> 3;
        const int row = v / WS, col = v % WS;
        const int oh = h0 + row, ow = w0 + col;
        const int c0 = cbt * 32 + q * 4;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (it < DITEMS && oh < H && ow < W && c0 < g.CB)
          val = *reinterpret_cast<const float4*>(g.B + ((((long)n * D + d) * H + oh) * W + ow) * g.bld + c0);
        tmp[i] = val;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int it = base + tid + i * NT;
        if (it < DITEMS) *reinterpret_cast<float4*>(&dys[it * 4]) = tmp[i];
      }
    }
  };

  int ch = next_valid(c_begin);
  while (ch < c_end) {
    __syncthreads();  // every wave finished reading the previous chunk
    stage(ch);
    __syncthreads();
    const int nxt = next_valid(ch + 1);
    // ---- MFMA: K runs over the chunk's voxels, 2 per instruction (lane half) ----
    // voxel pairs are consumed in groups of 4 (one shared-tap quarter per wave); rows narrower
    // than 4 pairs (WS = 4, 2: the MRI slabs' W = 12, 9, 4, 2) take their group from 2 or 4 rows
    constexpr int PP = WS / 2;
    constexpr int RSTEP = PP >= 4 ? 1 : 4 / PP;
    static_assert(R % RSTEP == 0 && (PP >= 4 ? PP % 4 == 0 : 4 % PP == 0), "pair groups must tile the chunk");
#pragma unroll 1
    for (int r = 0; r < R; r += RSTEP) {
      const float* xrow = &xs[(r * XW + lh) * 32 + li];
      const float* drow = &dys[(r * WS + lh) * 32 + li];
      for (int p4 = 0; p4 < (PP >= 4 ? PP : 4); p4 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = p4 + u;
          const int xo = PP >= 4 ? e * 64 : (e / PP) * XW * 32 + (e % PP) * 64;
          const int yo = PP >= 4 ? e * 64 : (e / PP) * WS * 32 + (e % PP) * 64;
          const float bv = drow[yo];
          float av[TPW];
#pragma unroll
          for (int k = 0; k < TPW; ++k) av[k] = xrow[xo + toff[k]];
#pragma unroll
          for (int k = 0; k < TPW; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k], bv, acc[k], 0, 0, 0);
          if (u == wave)  // wave-uniform: this wave's quarter of the shared tap
            acc_sh = __builtin_amdgcn_mfma_f32_32x32x2f32(xrow[xo + SHARED_OFF], bv, acc_sh, 0, 0, 0);
        }
      }
    }
    ch = nxt;
  }
  // ---- shared tap: sum the four waves' partial accumulators in a fixed order through LDS ----
  __syncthreads();
  float* red = xs;  // 4 waves x 16 regs x 64 lanes floats = 16 KiB <= sizeof(xs)
#pragma unroll
  for (int j = 0; j < 16; ++j) red[(wave * 16 + j) * 64 + lane] = acc_sh[j];
  __syncthreads();

  const int taps = KS * KS * KS;
  const int cb = cbt * 32 + li;
  if (cb < g.CB) {
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
      const int tap = kd * KS * KS + wave + 4 * k;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = (j & 3) + 8 * (j >> 2) + 4 * lh;
        const int oca = cat * 32 + row;
        if (oca < g.CA) partial[(((long)split * taps + tap) * g.CA + oca) * g.CB + cb] = acc[k][j];
      }
    }
    if (wave == 0) {
      const int tap = kd * KS * KS + KS * KS - 1;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = (j & 3) + 8 * (j >> 2) + 4 * lh;
        const int oca = cat * 32 + row;
        const float v = ((red[(0 * 16 + j) * 64 + lane] + red[(1 * 16 + j) * 64 + lane]) +
                         (red[(2 * 16 + j) * 64 + lane] + red[(3 * 16 + j) * 64 + lane]));
        if (oca < g.CA) partial[(((long)split * taps + tap) * g.CA + oca) * g.CB + cb] = v;
      }
    }
  }
}

template <int KS, int R, int WS>
int launch_wgrad_lds(msk_ctx* ctx, const WGrad& g, int num_cu) {
  const int taps = KS * KS * KS;
  const int ca_tiles = (g.CA + 31) / 32, cb_tiles = (g.CB + 31) / 32;
  const long chunks = (long)g.N * g.BD * ((g.BH + R - 1) / R) * ((g.BW + WS - 1) / WS);
  const long tasks = (long)KS * ca_tiles * cb_tiles;
  // ~3 rounds of 2-3 resident workgroups per CU
  long splits = ((long)num_cu * ctx->wgrad_rounds + tasks - 1) / tasks;
  if (splits > chunks) splits = chunks;
  if (splits < 1) splits = 1;
  const size_t per = (size_t)taps * g.CA * g.CB * sizeof(float);
  while (splits > 1 && splits * per > ((size_t)1 << 30)) --splits;
  const int cps = (int)((chunks + splits - 1) / splits);
  splits = (chunks + cps - 1) / cps;  // no empty splits
  float* partial = (float*)msk_workspace(ctx, (size_t)splits * per);
  if (!partial) return -1;
  {
    msk_launch_scope ls(ctx, wgrad_tag(ctx, "wgrad_lds_mfma", g, (int)splits));
    hipLaunchKernelGGL((wgrad_lds_mfma_k<KS, R, WS>), dim3((unsigned)tasks, (unsigned)splits), dim3(256), 0,
                       ctx->stream, g, (int)splits, (int)chunks, cps, partial);
    MSK_LAUNCH_CHECK(ctx);
  }
  return msk_wgrad_reduce(ctx, partial, (int)splits, taps, g.CA, g.CB, g.dw, g.accumulate);
}

// Folded-taps variant for layers with a tiny channel count on one side (stride 1, same grid):
//   FOLD_ROWS (in_tr, CA small): MFMA rows = (tap, ca) pairs, cols = cb.
//        dW[cb][ca][tap] = sum_m A[m + tap - p][ca] * B[m][cb]     (A gathered, B streamed)
//   !FOLD_ROWS (out_tr, CB small): rows = ca, MFMA cols = (tap, cb) pairs.
//        dW[cb][ca][tap] = sum_u A[u][ca] * B[u - tap + p][cb]     (A streamed, B gathered)
// so the padded MFMA dimension holds useful (tap, channel) pairs instead of zeros.
template <bool FOLD_ROWS, int NT>
__global__ void __launch_bounds__(256, 2)
wgrad_fold_mfma_k(WGrad g, int splits, float* __restrict__ partial, unsigned a_bytes, unsigned b_bytes, int groups) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int CF = FOLD_ROWS ? g.CA : g.CB;  // folded (small) channel count
  const int CS = FOLD_ROWS ? g.CB : g.CA;  // streamed channel count
  const int taps = g.kd * g.kh * g.kw;
  const int Q = taps * CF;
  int b = blockIdx.x;
  const int grp = b % groups;  // group of NT folded tiles
  const int st = b / groups;   // tile of 32 streamed channels
  const int split = blockIdx.y * 4 + wave;
  if (split >= splits) return;

  // stride-1 same-size grids: one voxel index space for both tensors
  const int D = g.BD, H = g.BH, W = g.BW;
  const long M = (long)g.N * D * H * W;
  long per = (M + splits - 1) / splits;
  per = (per + 1) & ~1L;
  const long m0 = (long)split * per;
  long m1 = m0 + per;
  if (m1 > M) m1 = M;

  const int cs = st * 32 + li;
  const bool cs_ok = cs < CS;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rf = FOLD_ROWS ? ra : rb;   // gathered tensor
  const __amdgpu_buffer_rsrc_t rs = FOLD_ROWS ? rb : ra;   // streamed tensor
  const int ldf = FOLD_ROWS ? g.ald : g.bld, lds_ = FOLD_ROWS ? g.bld : g.ald;

  // per-lane folded pairs: voxel displacement (ed, eh, ew), linear delta, channel
  int ed[NT], eh[NT], ew[NT], delta[NT], cf[NT];
  bool qok[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int q = (grp * NT + t) * 32 + li;
    qok[t] = q < Q;
    const int tap = qok[t] ? q / CF : 0;
    cf[t] = qok[t] ? q % CF : 0;
    const int sgn = FOLD_ROWS ? 1 : -1;
    ed[t] = sgn * (tap / (g.kh * g.kw) - g.pd);
    eh[t] = sgn * ((tap / g.kw) % g.kh - g.ph);
    ew[t] = sgn * (tap % g.kw - g.pw);
    delta[t] = (ed[t] * H + eh[t]) * W + ew[t];
  }

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

  VoxCursor c;
  long m = m0 + lh;
  c.init(m, D, H, W);
  float fv_n[NT], sv_n;
  auto load_step = [&](long mm) {
    const bool live = mm < m1;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const bool ok = live && qok[t] && (unsigned)(c.d + ed[t]) < (unsigned)D &&
                      (unsigned)(c.h + eh[t]) < (unsigned)H && (unsigned)(c.w + ew[t]) < (unsigned)W;
      const unsigned off = ((unsigned)((int)mm + delta[t]) * (unsigned)ldf + (unsigned)cf[t]) * 4u;
      fv_n[t] = buf_load(rf, ok ? off : kOOB);
    }
    const unsigned offs = ((unsigned)mm * (unsigned)lds_ + (unsigned)cs) * 4u;
    sv_n = buf_load(rs, (live && cs_ok) ? offs : kOOB);
  };
  load_step(m);
  for (; m - lh < m1; m += 2) {
    float fv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) fv[t] = fv_n[t];
    const float sv = sv_n;
    c.advance2(D, H, W);
    load_step(m + 2);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if constexpr (FOLD_ROWS)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fv[t], sv, acc[t], 0, 0, 0);  // rows = pairs, cols = cb
      else
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv, fv[t], acc[t], 0, 0, 0);  // rows = ca, cols = pairs
    }
  }

  // D[row][col]: row = (j&3) + 8*(j>>2) + 4*lh, col = li
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int row = (j & 3) + 8 * (j >> 2) + 4 * lh;
      int q, cstream;
      if constexpr (FOLD_ROWS) {
        q = (grp * NT + t) * 32 + row;
        cstream = st * 32 + li;
      } else {
        q = (grp * NT + t) * 32 + li;
        cstream = st * 32 + row;
      }
      if (q < Q && cstream < CS) {
        const int tap = q / CF, cfold = q % CF;
        const int oca = FOLD_ROWS ? cfold : cstream, ocb = FOLD_ROWS ? cstream : cfold;
        partial[(((long)split * taps + tap) * g.CA + oca) * g.CB + ocb] = acc[t][j];
      }
    }
  }
}

const char* wgrad_tag(msk_ctx* ctx, const char* base, const WGrad& g, int splits) {
  if (!(ctx->prof && ctx->prof_shapes)) return base;
  char buf[160];
  snprintf(buf, sizeof(buf), "%s[ca=%d,cb=%d,k=%dx%dx%d,M=%ld,splits=%d]", base, g.CA, g.CB, g.kd, g.kh, g.kw,
           (long)g.N * g.BD * g.BH * g.BW, splits);
  return msk_intern_tag(ctx, buf);
}

template <int KW>
int launch_wgrad(msk_ctx* ctx, const WGrad& g, int splits, float* partial, unsigned a_bytes, unsigned b_bytes) {
  const int ca_tiles = (g.CA + 31) / 32, cb_tiles = (g.CB + 31) / 32;
  const int rows = g.kd * g.kh;
  msk_launch_scope ls(ctx, wgrad_tag(ctx, "wgrad_mfma", g, splits));
  hipLaunchKernelGGL((wgrad_mfma_k<KW, (KW >= 4 ? 4 : 8)>), dim3(rows * ca_tiles * cb_tiles, (splits + 3) / 4), dim3(256), 0, ctx->stream,
                     g, splits, partial, a_bytes, b_bytes);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

long pick_splits(long tasks, long M, size_t per_bytes) {
  // ~3 rounds of (256 CUs x 16 resident waves): short waves keep the last-round tail small
  // (exactly 1025 blocks on 1024 slots doubled the kernel time in the first measurement)
  long splits = (12288 + tasks - 1) / tasks;
  const long maxs = M / 256 > 0 ? M / 256 : 1;
  if (splits > maxs) splits = maxs;
  if (splits < 1) splits = 1;
  splits = (splits + 3) & ~3L;
  while (splits > 4 && splits * per_bytes > ((size_t)1 << 30)) splits -= 4;  // partial slab <= 1 GiB
  return splits;
}

int used_splits(long M, long splits) {
  long per_vox = (M + splits - 1) / splits;
  per_vox = (per_vox + 1) & ~1L;
  return (int)((M + per_vox - 1) / per_vox);
}

}  // namespace

int msk_gconv_halo_mfma(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap) {
  // eligibility: cubic odd kernel (3 or 5), stride 1, 'same' padding, same spatial dims
  const int ks = g.kd;
  if (!(g.kd == g.kh && g.kh == g.kw && (ks == 3 || ks == 5))) return 0;
  if (!(g.sd == 1 && g.sh == 1 && g.sw == 1)) return 0;
  if (!(g.pd == ks / 2 && g.ph == ks / 2 && g.pw == ks / 2)) return 0;
  if (!(g.SD == g.DD && g.SH == g.DH && g.SW == g.DW)) return 0;
  // With stride 1 and p = k/2 the transposed gather equals a forward gather with flipped taps.
  const int flip = g.transposed ? 1 : 0;
  const int taps = ks * ks * ks;

  // tiny-channel 5^3 layers of VNet: VALU halo kernel (useful FLOPs only)
  const bool valu_in = ks == 5 && g.CK == 1 && g.CN == 16;                 // in_tr.conv1 forward
  const bool valu_out = ks == 5 && g.CK == 32 && g.CN >= 1 && g.CN <= 4;   // out_tr.conv1 forward
  const bool valu_outT = ks == 5 && g.CN == 32 && g.CK >= 1 && g.CK <= 4;  // out_tr.conv1 data gradient
  if (valu_in || valu_out || valu_outT) {
    const float* wp = msk_pack_weights_get(ctx, w_canon, A, B, taps, swap, flip, ks, ks, ks, 0, g.CK, g.CN, 0, 0);
    if (!wp) return -1;
    HaloArgs a{};
    a.src = g.src; a.sld = g.sld; a.dst = g.dst; a.dld = g.dld;
    a.N = g.N; a.D = g.DD; a.H = g.DH; a.W = g.DW;
    a.CK = g.CK; a.CN = g.CN;
    a.bias = g.bias; a.accumulate = g.accumulate;
    a.vec = (g.CK % 4 == 0) && (g.sld % 4 == 0) && (((uintptr_t)g.src) % 16 == 0);
    int rc = 0;
    if (valu_in) rc = launch_halo_valu<5, 1, 16>(ctx, a, wp);
    else if (valu_out) {
      switch (g.CN) {
        case 1: rc = launch_halo_valu<5, 32, 1>(ctx, a, wp); break;
        case 2: rc = launch_halo_valu<5, 32, 2>(ctx, a, wp); break;
        case 3: rc = launch_halo_valu<5, 32, 3>(ctx, a, wp); break;
        default: rc = launch_halo_valu<5, 32, 4>(ctx, a, wp); break;
      }
    } else {
      switch (g.CK) {
        case 1: rc = launch_halo_valu<5, 1, 32>(ctx, a, wp); break;
        case 2: rc = launch_halo_valu<5, 2, 32>(ctx, a, wp); break;
        case 3: rc = launch_halo_valu<5, 3, 32>(ctx, a, wp); break;
        default: rc = launch_halo_valu<5, 4, 32>(ctx, a, wp); break;
      }
    }
    return rc == 0 ? 1 : rc;
  }

  const int KC = (g.CK + 7) / 8;
  const int npad = ((g.CN + 31) / 32) * 32;
  const size_t wbytes = (size_t)taps * KC * 2 * npad * 4 * sizeof(float);
  const float* wm = msk_pack_weights_get(ctx, w_canon, A, B, taps, swap, flip, ks, ks, ks, 1, g.CK, g.CN, KC, npad);
  if (!wm) return -1;

  HaloArgs a{};
  a.src = g.src; a.sld = g.sld; a.dst = g.dst; a.dld = g.dld;
  a.N = g.N; a.D = g.DD; a.H = g.DH; a.W = g.DW;
  a.CK = g.CK; a.CN = g.CN;
  a.wm = reinterpret_cast<const float4*>(wm);
  a.KC = KC; a.npad = npad;
  a.bias = g.bias; a.accumulate = g.accumulate;
  a.vec = (g.CK % 4 == 0) && (g.sld % 4 == 0) && (((uintptr_t)g.src) % 16 == 0);
  const int ntn = npad / 32;
  // 256-voxel tile shape: the one that wastes the fewest padded voxels on this volume (the MRI
  // slabs have W = 12, 9, 8, 4, 2: a W=8 tile is 56% full at W=9, 50% at W=4); ties go to the
  // widest W (longest contiguous runs in the staging loads).
  static const int kTiles[5][3] = {{2, 4, 32}, {2, 8, 16}, {4, 8, 8}, {4, 16, 4}, {8, 16, 2}};
  // ties go to the first in this order: measured on 32ch@128^3 (tools/bench_conv.py) the compact <4,8,8> tile
  // (smallest halo, 3 workgroups per CU) beats <2,4,32> by 3% forward and 5% on the data gradient
  static const int kOrder[5] = {2, 1, 0, 3, 4};
  int best = 0;
  double best_util = -1.0;
  for (int oi = 0; oi < 5; ++oi) {
    const int i = kOrder[oi];
    const double padded = (double)msk_cdiv(g.DD, kTiles[i][0]) * kTiles[i][0] * msk_cdiv(g.DH, kTiles[i][1]) * kTiles[i][1] *
                          msk_cdiv(g.DW, kTiles[i][2]) * kTiles[i][2];
    const double util = (double)g.DD * g.DH * g.DW / padded;
    if (util > best_util * 1.02) { best_util = util; best = i; }
  }
  if (ctx->halo_tile >= 0 && ctx->halo_tile < 5) best = ctx->halo_tile;
  int rc;
  if (ks == 5) {
    switch (best) {
      case 0: rc = launch_halo<2, 4, 32, 5>(ctx, a, ntn); break;
      case 1: rc = launch_halo<2, 8, 16, 5>(ctx, a, ntn); break;
      case 2: rc = launch_halo<4, 8, 8, 5>(ctx, a, ntn); break;
      case 3: rc = launch_halo<4, 16, 4, 5>(ctx, a, ntn); break;
      default: rc = launch_halo<8, 16, 2, 5>(ctx, a, ntn); break;
    }
  } else {
    switch (best) {
      case 0: rc = launch_halo<2, 4, 32, 3>(ctx, a, ntn); break;
      case 1: rc = launch_halo<2, 8, 16, 3>(ctx, a, ntn); break;
      case 2: rc = launch_halo<4, 8, 8, 3>(ctx, a, ntn); break;
      case 3: rc = launch_halo<4, 16, 4, 3>(ctx, a, ntn); break;
      default: rc = launch_halo<8, 16, 2, 3>(ctx, a, ntn); break;
    }
  }
  return rc == 0 ? 1 : rc;
}

int msk_gconv_gather_mfma(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap) {
  const size_t sbytes = (size_t)g.N * g.SD * g.SH * g.SW * g.sld * sizeof(float);
  if (sbytes >= 0xFFFFFFF0ull) return 0;  // 32-bit byte offsets in the kernel
  const int taps = g.kd * g.kh * g.kw;
  const int KC = (g.CK + 7) / 8;
  const int npad = ((g.CN + 31) / 32) * 32;
  const float* wm = msk_pack_weights_get(ctx, w_canon, A, B, taps, swap, 0, g.kd, g.kh, g.kw, 1, g.CK, g.CN, KC, npad);
  if (!wm) return -1;
  const int ntn = npad / 32;
  const int NR = (ntn == 1 || ntn == 2 || ntn == 4 || ntn == 8) ? ntn : 1;
  const int cs_d = g.transposed ? g.sd : 1, cs_h = g.transposed ? g.sh : 1, cs_w = g.transposed ? g.sw : 1;
  long tiles_max = 1;
  for (int rd = 0; rd < cs_d; ++rd)
    for (int rh = 0; rh < cs_h; ++rh)
      for (int rw = 0; rw < cs_w; ++rw) {
        const long QD = (g.DD - rd + cs_d - 1) / cs_d, QH = (g.DH - rh + cs_h - 1) / cs_h, QW = (g.DW - rw + cs_w - 1) / cs_w;
        if (QD <= 0 || QH <= 0 || QW <= 0) continue;
        const long t = ((long)g.N * QD * QH * QW + 127) / 128;
        if (t > tiles_max) tiles_max = t;
      }
  // deep levels have few dst voxels: trade A re-reads (tiny tensors) for parallelism over N
  int NRsel = NR;
  while (NRsel > 1 && tiles_max * cs_d * cs_h * cs_w * (ntn / NRsel) < 2L * ctx->num_cu) NRsel >>= 1;
  const int vec = (g.CK % 4 == 0) && (g.sld % 4 == 0) && (((uintptr_t)g.src) % 16 == 0);
  const char* tag = "gconv_gather_mfma";
  if (ctx->prof && ctx->prof_shapes) {
    char buf[160];
    snprintf(buf, sizeof(buf), "gconv_gather_mfma[ck=%d,cn=%d,k=%dx%dx%d,T=%d,dst=%dx%dx%dx%d]", g.CK, g.CN, g.kd, g.kh,
             g.kw, g.transposed, g.N, g.DD, g.DH, g.DW);
    tag = msk_intern_tag(ctx, buf);
  }
  msk_launch_scope ls(ctx, tag);
  dim3 grid((unsigned)tiles_max, cs_d * cs_h * cs_w, ntn / NRsel);
  const float4* w4 = reinterpret_cast<const float4*>(wm);
  switch (NRsel) {
    case 8: hipLaunchKernelGGL((gconv_gather_mfma_k<8>), grid, dim3(256), 0, ctx->stream, g, w4, KC, npad, (unsigned)sbytes, vec, (int)tiles_max); break;
    case 4: hipLaunchKernelGGL((gconv_gather_mfma_k<4>), grid, dim3(256), 0, ctx->stream, g, w4, KC, npad, (unsigned)sbytes, vec, (int)tiles_max); break;
    case 2: hipLaunchKernelGGL((gconv_gather_mfma_k<2>), grid, dim3(256), 0, ctx->stream, g, w4, KC, npad, (unsigned)sbytes, vec, (int)tiles_max); break;
    default: hipLaunchKernelGGL((gconv_gather_mfma_k<1>), grid, dim3(256), 0, ctx->stream, g, w4, KC, npad, (unsigned)sbytes, vec, (int)tiles_max); break;
  }
  MSK_LAUNCH_CHECK(ctx);
  return 1;
}

int msk_wgrad_mfma(msk_ctx* ctx, const WGrad& g) {
  const int taps = g.kd * g.kh * g.kw;
  const long M = (long)g.N * g.BD * g.BH * g.BW;
  // 32-bit byte offsets inside the kernels: both tensors must stay below 4 GiB
  const size_t abytes = (size_t)g.N * g.AD * g.AH * g.AW * g.ald * sizeof(float);
  const size_t bbytes = (size_t)M * g.bld * sizeof(float);
  if (abytes >= 0xFFFFFFF0ull || bbytes >= 0xFFFFFFF0ull) return 0;
  const size_t per = (size_t)taps * g.CA * g.CB * sizeof(float);

  // tiny channel count on one side + stride-1 same grid -> fold the taps into the MFMA tile
  const bool same_grid = g.sd == 1 && g.sh == 1 && g.sw == 1 && g.AD == g.BD && g.AH == g.BH && g.AW == g.BW;
  const bool fold_rows = same_grid && taps > 1 && g.CA <= 8 && g.CA < g.CB;
  const bool fold_cols = same_grid && taps > 1 && g.CB <= 8 && !fold_rows;
  if (fold_rows || fold_cols) {
    constexpr int NT = 4;
    const int CF = fold_rows ? g.CA : g.CB, CS = fold_rows ? g.CB : g.CA;
    const int tiles = (taps * CF + 31) / 32;
    const int groups = (tiles + NT - 1) / NT;
    const int stiles = (CS + 31) / 32;
    const long splits = pick_splits((long)groups * stiles, M, per);
    float* partial = (float*)msk_workspace(ctx, (size_t)splits * per);
    if (!partial) return -1;
    {
      msk_launch_scope ls(ctx, wgrad_tag(ctx, fold_rows ? "wgrad_fold_rows_mfma" : "wgrad_fold_cols_mfma", g, (int)splits));
      dim3 grid(groups * stiles, (unsigned)((splits + 3) / 4));
      if (fold_rows)
        hipLaunchKernelGGL((wgrad_fold_mfma_k<true, NT>), grid, dim3(256), 0, ctx->stream, g, (int)splits, partial,
                           (unsigned)abytes, (unsigned)bbytes, groups);
      else
        hipLaunchKernelGGL((wgrad_fold_mfma_k<false, NT>), grid, dim3(256), 0, ctx->stream, g, (int)splits, partial,
                           (unsigned)abytes, (unsigned)bbytes, groups);
      MSK_LAUNCH_CHECK(ctx);
    }
    int rc = msk_wgrad_reduce(ctx, partial, used_splits(M, splits), taps, g.CA, g.CB, g.dw, g.accumulate);
    return rc == 0 ? 1 : rc;
  }

  // stride-1 'same' cubic 3^3 / 5^3 with vector-friendly channels: LDS-staged kernel
  const bool cubic = g.kd == g.kh && g.kh == g.kw && (g.kd == 3 || g.kd == 5) && g.pd == g.kd / 2 &&
                     g.ph == g.kd / 2 && g.pw == g.kd / 2;
  const bool vec_ok = g.ald % 4 == 0 && g.bld % 4 == 0 && g.CA % 4 == 0 && g.CB % 4 == 0 &&
                      ((uintptr_t)g.A) % 16 == 0 && ((uintptr_t)g.B) % 16 == 0;
  if (same_grid && cubic && vec_ok && ctx->conv_impl != 5) {
    // chunk shape (R rows x WS columns) with the fewest padded voxels on this plane; ties -> widest
    static const int kChunks[5][2] = {{4, 32}, {8, 16}, {16, 8}, {32, 4}, {32, 2}};
    int best = 0;
    double best_util = -1.0;
    for (int i = 0; i < 5; ++i) {
      const double padded = (double)msk_cdiv(g.BH, kChunks[i][0]) * kChunks[i][0] * msk_cdiv(g.BW, kChunks[i][1]) * kChunks[i][1];
      const double util = (double)g.BH * g.BW / padded;
      if (util > best_util * 1.02) { best_util = util; best = i; }
    }
    if (ctx->wgrad_chunk >= 0 && ctx->wgrad_chunk < 5) best = ctx->wgrad_chunk;
    int rc;
    if (g.kd == 5) {
      switch (best) {
        case 0: rc = launch_wgrad_lds<5, 4, 32>(ctx, g, ctx->num_cu); break;
        case 1: rc = launch_wgrad_lds<5, 8, 16>(ctx, g, ctx->num_cu); break;
        case 2: rc = launch_wgrad_lds<5, 16, 8>(ctx, g, ctx->num_cu); break;
        case 3: rc = launch_wgrad_lds<5, 32, 4>(ctx, g, ctx->num_cu); break;
        default: rc = launch_wgrad_lds<5, 32, 2>(ctx, g, ctx->num_cu); break;
      }
    } else {
      switch (best) {
        case 0: rc = launch_wgrad_lds<3, 4, 32>(ctx, g, ctx->num_cu); break;
        case 1: rc = launch_wgrad_lds<3, 8, 16>(ctx, g, ctx->num_cu); break;
        case 2: rc = launch_wgrad_lds<3, 16, 8>(ctx, g, ctx->num_cu); break;
        case 3: rc = launch_wgrad_lds<3, 32, 4>(ctx, g, ctx->num_cu); break;
        default: rc = launch_wgrad_lds<3, 32, 2>(ctx, g, ctx->num_cu); break;
      }
    }
    return rc == 0 ? 1 : rc;
  }

  if (g.kw < 1 || g.kw > 5) return 0;
  const int ca_tiles = (g.CA + 31) / 32, cb_tiles = (g.CB + 31) / 32;
  const long splits = pick_splits((long)g.kd * g.kh * ca_tiles * cb_tiles, M, per);
  float* partial = (float*)msk_workspace(ctx, (size_t)splits * per);
  if (!partial) return -1;
  int rc;
  switch (g.kw) {
    case 1: rc = launch_wgrad<1>(ctx, g, (int)splits, partial, (unsigned)abytes, (unsigned)bbytes); break;
    case 2: rc = launch_wgrad<2>(ctx, g, (int)splits, partial, (unsigned)abytes, (unsigned)bbytes); break;
    case 3: rc = launch_wgrad<3>(ctx, g, (int)splits, partial, (unsigned)abytes, (unsigned)bbytes); break;
    case 4: rc = launch_wgrad<4>(ctx, g, (int)splits, partial, (unsigned)abytes, (unsigned)bbytes); break;
    default: rc = launch_wgrad<5>(ctx, g, (int)splits, partial, (unsigned)abytes, (unsigned)bbytes); break;
  }
  if (rc != 0) return rc;
  rc = msk_wgrad_reduce(ctx, partial, used_splits(M, splits), taps, g.CA, g.CB, g.dw, g.accumulate);
  return rc == 0 ? 1 : rc;
}
