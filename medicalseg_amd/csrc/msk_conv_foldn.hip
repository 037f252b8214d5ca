// 'Same' 5^3 convolution with a TINY output-channel count (5*CN <= 16: out_tr.conv1, 32 -> ncls <= 3, vnet.py:165)
// on the fp32 matrix pipe with the kd taps FOLDED INTO THE MFMA COLUMNS, marching along D.
//
// Padding CN = 3 to a 16-wide MFMA tile wastes 13 of 16 columns; the channel-quad VALU kernel this replaces
// (conv_halo_valu2_k, 1.7 ms for 32 -> 3 @ 2x128^3) staged ONE channel quad of a 3-D halo tile at a time, i.e. it
// fetched 16 of the 128 bytes of a voxel per pass, eight passes per tile: round-1 counters showed 19x the algorithmic
// traffic on the fabric.  Here
//   * the N dimension of v_mfma_f32_16x16x4_f32 enumerates (co, kd) pairs (15 of 16 columns for CN = 3): one INPUT
//     plane d' contributes to the five output planes d' + 2 - kd at once,
//         Z[w][(co, kd)] = sum_{kh, kw, ci} x[d', h + kh - 2, w + kw - 2, ci] * W[kd][kh][kw][ci][co],
//     K = 25 (kh, kw) taps x channel quads, M = 16 consecutive output positions along W (no halo rows in M);
//   * a workgroup owns an (8 x 16) column of (h, w) and walks along D.  The partial sums ride in the accumulator
//     registers: before plane d' + 1 is accumulated every column moves one kd to the right inside its 16-lane row
//     (DPP row_shr:1), so column (co, kd) always holds the partial sum of output plane d' + 2 - kd; what leaves
//     kd = 4 is a finished output plane.  No 3-D halo: each input plane is staged ONCE per column, whole 128-byte
//     voxels (all channels), halo only in (h, w): 12 x 20 / (8 x 16) = 1.9x, served by L2 between neighbouring columns.
//   * LDS holds one input plane (12 x 20 voxels x 32 channels, bank-spread layout below: conflict-free ds_read_b32 for
//     the A operand); the next plane is prefetched into registers during the 200 MFMAs of
//     the current one.  B operands (weights, 51 KB packed per lane) stream from L1/L2 as 16-byte loads.
#include "msk_conv.h"
#include "msk_wbf.h"

typedef _Float16 fn_f16x8 __attribute__((ext_vector_type(8)));

namespace {

struct FNArgs {
  const float* src;
  int sld;
  float* dst;
  int dld;
  int N, D, H, W, CN;
  const float4* wb;  // [tap2d = kh*5 + kw][quad group][lane][4]: B operand of quad 4*group + e for lane (k = lane/16, col = lane%16)
  const float* bias;
  const float* prelu;
  int accumulate;
  int tiles_h, tiles_w, segs, seg_len, nblk;
  unsigned src_bytes;
};

__device__ __forceinline__ int xcd_remap_fn(int bid, int nb) {
  const int q = nb >> 3, r = nb & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// wb[((tap2d*QG + g)*64 + lane)*4 + e] = W[kd][kh][kw][ci = 4*(4g + e) + lane/16][co], (co, kd) = divmod(lane % 16, 5)
__global__ void __launch_bounds__(256)
pack_foldn_weights_k(const float* __restrict__ w, int A, int B, int swap, int flip, int CK, int CN, int QG,
                     float* __restrict__ out) {
  const int total = 25 * QG * 64 * 4;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int e = idx & 3, lane = (idx >> 2) & 63, g = (idx >> 8) % QG, tap2 = (idx >> 8) / QG;
    const int j = lane & 15, k = lane >> 4;
    const int ci = 4 * (4 * g + e) + k;
    float v = 0.f;
    if (j < 5 * CN && ci < CK) {
      const int co = j / 5, kd = j % 5;
      int tap = kd * 25 + tap2;
      if (flip) tap = 124 - tap;
      const int ia = swap ? co : ci, ib = swap ? ci : co;
      v = w[((long)ia * B + ib) * 125 + tap];
    }
    out[idx] = v;
  }
}

template <int Q>  // channel quads (CK = 4 Q, Q % 4 == 0)
__global__ void __launch_bounds__(256, 3)  // one wavefront per SIMD and workgroup: three workgroups per CU hide the staging
conv_foldn_k(FNArgs a) {
  constexpr int TH = 8, TW = 16, HH = TH + 4, HW = TW + 4, NV = HH * HW;  // 12 x 20 = 240 voxels per plane
  constexpr int NLD = (NV * Q + 255) / 256;                               // 16-byte loads per thread and plane
  constexpr int QG = Q / 4;
  constexpr unsigned kOOB = 0xFFFFFFF0u;
  // plane in LDS as dwords [quad: QS][channel pair of the quad: HS][voxel][2]: ds_read_b32 is serviced in two groups of 32
  // lanes over 32 banks (MI355X_MICROARCH.md, LDS): the half-wave (k = 0, 1 | k = 2, 3) of an A read covers 32 consecutive
  // dwords; ds_write_b64 in four groups of 16 lanes = 8 quads x 2 voxels: QS = 4 (mod 32) puts quad q on banks 4q .. 4q+3.
  // (The first layout, [quad][voxel][4], measured 59 % of the LDS cycles in bank conflicts: a half-wave touched only
  // dwords 0, 1 (mod 4).)
  constexpr int HS = 2 * NV, QS = 2 * HS + 4;
  static_assert(QS % 32 == 4, "bank spreading");
  __shared__ __attribute__((aligned(16))) float lds[Q * QS + 64];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  int t = xcd_remap_fn(blockIdx.x, a.nblk);
  const int twi = t % a.tiles_w;
  t /= a.tiles_w;
  const int thi = t % a.tiles_h;
  t /= a.tiles_h;
  const int seg = t % a.segs;
  const int n = t / a.segs;
  const int h0 = thi * TH, w0 = twi * TW;
  const int d_begin = seg * a.seg_len;
  const int d_end = min(a.D, d_begin + a.seg_len);

  // staging map: 8 consecutive lanes (Q = 8) fetch the quads of one voxel = its whole 128-byte line
  const __amdgpu_buffer_rsrc_t sres = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
  unsigned st_off[NLD];
  int st_lds[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int e = j * 256 + tid;
    const int v = e / Q, q = e % Q;
    const int hh = v / HW, ww = v % HW;
    const int gh = h0 - 2 + hh, gw = w0 - 2 + ww;
    const bool ok = v < NV;
    const bool inb = ok && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
    st_off[j] = inb ? (unsigned)(((((long)n * a.D * a.H + gh) * a.W + gw) * a.sld + 4 * q) * 4) : kOOB;
    st_lds[j] = ok ? q * QS + v * 2 : -1;
  }
  const long plane_bytes = (long)a.H * a.W * a.sld * 4;

  float4 pre[NLD];
  auto fetch = [&](int dp) {
    const bool live = dp >= 0 && dp < a.D;  // uniform
    const unsigned soff = live ? (unsigned)(dp * plane_bytes) : 0u;
#pragma unroll
    for (int j = 0; j < NLD; ++j)
      pre[j] = live ? __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(sres, (int)st_off[j], (int)soff, 0))
                    : make_float4(0.f, 0.f, 0.f, 0.f);
  };

  const float* ldsf = lds;
  const int abase = (lk >> 1) * HS + (2 * wave * HW + li) * 2 + (lk & 1);  // this lane's A element of tap (0, 0), quad 0, first row of the wave
  const float4* wbl = a.wb + lane;

  const int col_co = li / 5, col_kd = li - col_co * 5;
  const bool col_live = li < 5 * a.CN;
  const bool col_first = col_kd == 0 || !col_live;
  f32x4 acc[2];  // the wave's two rows h0 + 2*wave + {0, 1}: two independent MFMA chains sharing every B operand
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) acc[rr] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int steps = (d_end - d_begin) + 4;
  fetch(d_begin - 2);
#pragma unroll 1
  for (int s = 0; s < steps; ++s) {
    const int dp = d_begin - 2 + s;
    const bool live = dp >= 0 && dp < a.D;
    __syncthreads();  // the previous plane's reads are done
    if (live) {
#pragma unroll
      for (int j = 0; j < NLD; ++j)
        if (st_lds[j] >= 0) {
          *reinterpret_cast<float2*>(lds + st_lds[j]) = make_float2(pre[j].x, pre[j].y);
          *reinterpret_cast<float2*>(lds + st_lds[j] + HS) = make_float2(pre[j].z, pre[j].w);
        }
    }
    __syncthreads();
    if (s + 1 < steps) fetch(dp + 1);

    // every partial sum moves one kd to the right (it now belongs to the same output plane seen from plane dp)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float cur = acc[rr][r];  // (a bit_cast applied straight to the vector element reads element 0: clang lvalue bug)
        const int sh = __builtin_amdgcn_update_dpp(0, __float_as_int(cur), 0x111 /* row_shr:1 */, 0xF, 0xF, true);
        acc[rr][r] = col_first ? 0.f : __int_as_float(sh);
      }
    if (live) {
      // A and B operands one (kh, kw) step ahead of their MFMAs: the waits then cover loads issued 16 MFMAs earlier (the
      // compiler's own schedule read LDS just in time).  The packed weights carry one padding tap and the LDS array a
      // padding row for the step behind the last.
      float4 bn[QG];
      float an[2][Q];
#pragma unroll
      for (int g = 0; g < QG; ++g) bn[g] = wbl[g * 64];
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        an[0][q] = ldsf[abase + q * QS];
        an[1][q] = ldsf[abase + q * QS + HW * 2];
      }
#pragma unroll 1
      for (int kh = 0; kh < 5; ++kh) {  // rolled: a fully unrolled plane (400 MFMAs) made the scheduler hoist loads into spills
        const float* arow = ldsf + abase + kh * HW * 2;
        const float4* brow = wbl + kh * 5 * QG * 64;
#pragma unroll
        for (int kw = 0; kw < 5; ++kw) {
          float4 b[QG];
          float av[2][Q];
          const int nxt = kw < 4 ? (kw + 1) * 2 : HW * 2;  // (kh, kw + 1) or (kh + 1, 0)
#pragma unroll
          for (int g = 0; g < QG; ++g) {
            b[g] = bn[g];
            bn[g] = brow[((kw + 1) * QG + g) * 64];
          }
#pragma unroll
          for (int q = 0; q < Q; ++q) {
            av[0][q] = an[0][q];
            av[1][q] = an[1][q];
            an[0][q] = arow[q * QS + nxt];
            an[1][q] = arow[q * QS + nxt + HW * 2];
          }
#pragma unroll
          for (int g = 0; g < QG; ++g) {
            const float bq[4] = {b[g].x, b[g].y, b[g].z, b[g].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][4 * g + e], bq[e], acc[0], 0, 0, 0);
              acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][4 * g + e], bq[e], acc[1], 0, 0, 0);
            }
          }
        }
      }
    }
    // column (co, 4) now holds output plane dp - 2 complete
    const int d = dp - 2;
    if (d >= d_begin && d < d_end && col_live && col_kd == 4) {
      const float bv = a.bias ? a.bias[col_co] : 0.f;
      const float sl = a.prelu ? a.prelu[col_co] : 1.f;
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int gh = h0 + 2 * wave + rr;
        if (gh < a.H) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int gw = w0 + 4 * lk + r;
            if (gw < a.W) {
              float* o = a.dst + ((((long)n * a.D + d) * a.H + gh) * a.W + gw) * a.dld + col_co;
              float v = acc[rr][r] + bv;
              if (a.accumulate) v += *o;
              if (v < 0.f) v *= sl;
              *o = v;
            }
          }
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// The same convolution with fp16 two-piece operands (msk_wbf.h, option "conv_split" 2): x * s = h + l in fp16, three
// v_mfma_f32_16x16x32_f16 per (tap, row) -- ALL 32 input channels in one instruction -- instead of eight fp32 MFMAs: 5.3x
// less matrix time, the kernel becomes LDS / L2 bound.  Differences to conv_foldn_k:
//   * the 25 (kh, kw) taps are SPLIT OVER THE FOUR WAVEFRONTS (7/6/6/6): a wavefront's weight fragments (two pieces per
//     tap) live in registers for the whole march -- no weight traffic at all after the prologue (streaming them per tap
//     would need 83 B/clk/CU from L1) -- and every wavefront runs all 8 rows of the tile for its taps.  The shift chain is
//     linear, so each wavefront carries its own partial sums through the planes; they meet only when an output plane
//     leaves the chain: 384 values per plane summed through LDS.
//   * LDS plane per piece [channel octet kg][voxel][8 x fp16]: a ds_read_b128 lane group then covers 16 consecutive
//     16-byte slots (conflict-free for every tap offset).
//   * x is scaled by the power of two of its device-side maximum (msk_absmax) while it is split on its way into LDS.
struct FNH2Args {
  const float* src;
  int sld;
  float* dst;
  int dld;
  int N, D, H, W, CN;
  const uint4* wb;  // [tap 25][piece 2][lane 64]: B fragment (8 fp16: ci = 8*(lane/16) .. +7) of column lane%16 = (co, kd)
  const float* bias;
  const float* prelu;
  int accumulate;
  int tiles_h, tiles_w, segs, seg_len, nblk;
  unsigned src_bytes;
  const float* x_amax;
  const float* w_amax;
};

// wb[((tap2d*2 + piece)*64 + lane)*8 + e] = piece of s_w * W[kd][kh][kw][ci = 8*(lane/16) + e][co], (co, kd) = divmod(lane%16, 5)
__global__ void __launch_bounds__(256)
pack_foldn_h2_weights_k(const float* __restrict__ w, int A, int B, int swap, int flip, int CN, const float* __restrict__ w_amax,
                        unsigned short* __restrict__ out) {
  const float sw = wbf_scale_of(w_amax);
  const int total = 25 * 64 * 8;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int e = idx & 7, lane = (idx >> 3) & 63, tap2 = idx >> 9;
    const int j = lane & 15, ci = 8 * (lane >> 4) + e;
    float v = 0.f;
    if (j < 5 * CN) {
      const int co = j / 5, kd = j % 5;
      int tap = kd * 25 + tap2;
      if (flip) tap = 124 - tap;
      const int ia = swap ? co : ci, ib = swap ? ci : co;
      v = w[((long)ia * B + ib) * 125 + tap] * sw;
    }
    const _Float16 h = (_Float16)v;
    out[((tap2 * 2 + 0) * 64 + lane) * 8 + e] = __builtin_bit_cast(unsigned short, h);
    out[((tap2 * 2 + 1) * 64 + lane) * 8 + e] = __builtin_bit_cast(unsigned short, (_Float16)(v - (float)h));
  }
}

__global__ void __launch_bounds__(256, 3)  // three workgroups per CU (24 spilled registers cost less than the third workgroup gains: 0.35 vs 0.36-0.44 ms)
conv_foldn_h2_k(FNH2Args a) {
  constexpr int TH = 8, TW = 16, HH = TH + 4, HW = TW + 4, NV = HH * HW;  // 12 x 20 = 240 voxels per plane (= 15 x 16)
  constexpr int NLD = (NV * 8 + 255) / 256;                               // 16-byte loads per thread and plane
  constexpr unsigned kOOB = 0xFFFFFFF0u;
  constexpr int NT0 = 7, NT1 = 6;                                         // taps of wavefront 0 / of the others
  constexpr int PP = 4 * NV;                                              // 16-byte slots per piece: [kg 4][voxel]
  __shared__ uint4 lds[2 * PP + 8];
  __shared__ float red[4][TH * TW * 4];                                   // partial outputs [wave][row][pos][co (padded to 4)]

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 15, lk = lane >> 4;
  int t = xcd_remap_fn(blockIdx.x, a.nblk);
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

  // staging map: 8 consecutive lanes fetch the quads of one voxel = its whole 128-byte line
  const __amdgpu_buffer_rsrc_t sres = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
  unsigned st_off[NLD];
  int st_lds[NLD];  // byte offset inside a piece, or -1
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int e = j * 256 + tid;
    const int v = e >> 3, q = e & 7;
    const int hh = v / HW, ww = v % HW;
    const int gh = h0 - 2 + hh, gw = w0 - 2 + ww;
    const bool ok = v < NV;
    const bool inb = ok && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
    st_off[j] = inb ? (unsigned)(((((long)n * a.D * a.H + gh) * a.W + gw) * a.sld + 4 * q) * 4) : kOOB;
    st_lds[j] = ok ? ((q >> 1) * NV + v) * 16 + (q & 1) * 8 : -1;
  }
  const long plane_bytes = (long)a.H * a.W * a.sld * 4;
  float4 pre[NLD];
  auto fetch = [&](int dp) {
    const bool live = dp >= 0 && dp < a.D;  // uniform
    const unsigned soff = live ? (unsigned)(dp * plane_bytes) : 0u;
#pragma unroll
    for (int j = 0; j < NLD; ++j)
      pre[j] = live ? __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(sres, (int)st_off[j], (int)soff, 0))
                    : make_float4(0.f, 0.f, 0.f, 0.f);
  };

  // this wavefront's taps [tap0, tap0 + ntap) and their weight fragments (both pieces) in registers
  const int tap0 = wave == 0 ? 0 : NT0 - NT1 + NT1 * wave, ntap = wave == 0 ? NT0 : NT1;
  uint4 bh[NT0], bl[NT0];
  int aoff[NT0];  // slot offset of the tap inside the plane
#pragma unroll
  for (int j = 0; j < NT0; ++j) {
    const int tp = min(tap0 + j, 24);
    bh[j] = a.wb[(tp * 2 + 0) * 64 + lane];
    bl[j] = a.wb[(tp * 2 + 1) * 64 + lane];
    aoff[j] = (tp / 5) * HW + (tp % 5);
  }
  const int abase = lk * NV + li;  // slot of (row 0, tap (0, 0)) for this lane: [kg = lk][voxel]

  const int col_co = li / 5, col_kd = li - col_co * 5;
  const bool col_live = li < 5 * a.CN;
  const bool col_first = col_kd == 0 || !col_live;
  f32x4 acc[TH];
#pragma unroll
  for (int r = 0; r < TH; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int steps = (d_end - d_begin) + 4;
  fetch(d_begin - 2);
#pragma unroll 1
  for (int s = 0; s < steps; ++s) {
    const int dp = d_begin - 2 + s;
    const bool live = dp >= 0 && dp < a.D;
    __syncthreads();  // the previous plane's reads (tile and partial outputs) are done
    if (live) {
      char* lb = reinterpret_cast<char*>(lds);
#pragma unroll
      for (int j = 0; j < NLD; ++j)
        if (st_lds[j] >= 0) {
          uint2 hv, lv;
          wbf_split2h_pair(pre[j].x * sx, pre[j].y * sx, hv.x, lv.x);
          wbf_split2h_pair(pre[j].z * sx, pre[j].w * sx, hv.y, lv.y);
          *reinterpret_cast<uint2*>(lb + st_lds[j]) = hv;
          *reinterpret_cast<uint2*>(lb + PP * 16 + st_lds[j]) = lv;
        }
    }
    __syncthreads();
    if (s + 1 < steps) fetch(dp + 1);

    // every partial sum moves one kd to the right (it now belongs to the same output plane seen from plane dp)
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float cur = acc[r][e];
        const int sh = __builtin_amdgcn_update_dpp(0, __float_as_int(cur), 0x111 /* row_shr:1 */, 0xF, 0xF, true);
        acc[r][e] = col_first ? 0.f : __int_as_float(sh);
      }
    if (live) {
#pragma unroll
      for (int j = 0; j < NT0; ++j) {
        if (j < ntap) {  // wave-uniform
          const uint4* ap = lds + abase + aoff[j];
#pragma unroll
          for (int r = 0; r < TH; ++r) {
            const uint4 ah = ap[r * HW], al = ap[PP + r * HW];
            acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(fn_f16x8, al), __builtin_bit_cast(fn_f16x8, bh[j]), acc[r], 0, 0, 0);
            acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(fn_f16x8, ah), __builtin_bit_cast(fn_f16x8, bl[j]), acc[r], 0, 0, 0);
            acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(fn_f16x8, ah), __builtin_bit_cast(fn_f16x8, bh[j]), acc[r], 0, 0, 0);
          }
        }
      }
    }
    // column (co, 4) of every wavefront now holds its share of output plane dp - 2: sum the four shares through LDS
    const int d = dp - 2;
    const bool out_plane = d >= d_begin && d < d_end;  // uniform
    if (out_plane && col_live && col_kd == 4) {
#pragma unroll
      for (int r = 0; r < TH; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][(r * TW + 4 * lk + e) * 4 + col_co] = acc[r][e];
    }
    if (out_plane) {
      __syncthreads();
      for (int i = tid; i < TH * TW * 3; i += 256) {
        const int co = i % 3, pos = i / 3;
        const int r = pos / TW, pw = pos - r * TW;
        const int gh = h0 + r, gw = w0 + pw;
        if (co < a.CN && gh < a.H && gw < a.W) {
          const int k = pos * 4 + co;
          float v = ((red[0][k] + red[1][k]) + (red[2][k] + red[3][k])) * osc + (a.bias ? a.bias[co] : 0.f);
          float* o = a.dst + ((((long)n * a.D + d) * a.H + gh) * a.W + gw) * a.dld + co;
          if (a.accumulate) v += *o;
          if (a.prelu && v < 0.f) v *= a.prelu[co];
          *o = v;
        }
      }
    }
  }
}

}  // namespace

// Will msk_conv3d_fwd_ex route this problem to conv_foldn_h2_k?  Then the caller may keep a kWbfXformHeader-byte "xform"
// whose first amax array receives max |x| for the layer's weight gradient (wgrad_cbs_h2_k).  Mirrors the checks below and
// the dispatch order in run_gconv_one (conv_impl 0 only).
bool msk_gconv_foldn_h2_accepts(const msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, int cout) {
  if (ctx->conv_split != 2 || ctx->conv_impl != 0) return false;
  if (!(cd.kd == 5 && cd.kh == 5 && cd.kw == 5 && cd.sd == 1 && cd.sh == 1 && cd.sw == 1 && cd.pd == 2 && cd.ph == 2 && cd.pw == 2)) return false;
  if (!(x.c == 32 && cout >= 1 && 5 * cout <= 16)) return false;
  if (x.w < 12 || x.d < 4) return false;
  if (x.ld % 4 || (((uintptr_t)x.p) & 15)) return false;
  if ((size_t)x.n * x.d * x.h * x.w * x.ld * sizeof(float) >= 0xFFFFFFF0ull) return false;
  return true;
}

int msk_gconv_halo_foldn(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap, bool* act_fused) {
  if (!(g.kd == 5 && g.kh == 5 && g.kw == 5)) return 0;
  if (!(g.sd == 1 && g.sh == 1 && g.sw == 1 && g.pd == 2 && g.ph == 2 && g.pw == 2)) return 0;
  if (!(g.SD == g.DD && g.SH == g.DH && g.SW == g.DW)) return 0;
  if (!(g.CK == 32 && g.CN >= 1 && 5 * g.CN <= 16)) return 0;
  if (g.DW < 12 || g.DD < 4) return 0;  // narrow slabs keep the VALU kernels
  if (!((g.sld % 4 == 0) && (((uintptr_t)g.src) % 16 == 0))) return 0;
  constexpr int Q = 8, QG = Q / 4;
  const size_t sb = (size_t)g.N * g.SD * g.SH * g.SW * g.sld * sizeof(float);
  if (sb >= 0xFFFFFFF0ull) return 0;
  if (ctx->conv_split == 2 && ctx->conv_impl != 24) {  // 24 = A/B: the fp32-MFMA form
    unsigned short* wb2 = (unsigned short*)msk_workspace2(ctx, (size_t)25 * 2 * 64 * 8 * sizeof(unsigned short));
    if (!wb2) return -1;
    // a caller-kept xform (msk_conv3d_fwd_ex): max |x| goes into its header for the layer's weight gradient
    const float* x_amax = g.in_amax ? g.in_amax
                                    : msk_absmax(ctx, g.src, g.sld, g.CK, (long)g.N * g.SD * g.SH * g.SW, g.xform ? (float*)g.xform : nullptr);
    const float* w_amax = msk_absmax(ctx, w_canon, 4, 4, (125L * g.CK * g.CN + 3) / 4);
    if (!x_amax || !w_amax) return -1;
    if (g.xform) {
      if (g.in_amax)  // the maximum came with the tensor (msk_conv3d_fwd_ex2): copy it into the header
        MSK_CHECK_HIP(ctx, hipMemcpyAsync(g.xform, g.in_amax, kWbfAmaxWays * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
      ctx->xform_written = true;
    }
    {
      msk_launch_scope ls(ctx, "pack_weights_foldn");
      hipLaunchKernelGGL(pack_foldn_h2_weights_k, dim3(50), dim3(256), 0, ctx->stream, w_canon, A, B, swap, g.transposed ? 1 : 0, g.CN,
                         w_amax, wb2);
      MSK_LAUNCH_CHECK(ctx);
    }
    FNH2Args a{};
    a.src = g.src; a.sld = g.sld; a.dst = g.dst; a.dld = g.dld;
    a.N = g.N; a.D = g.DD; a.H = g.DH; a.W = g.DW; a.CN = g.CN;
    a.wb = (const uint4*)wb2; a.bias = g.bias; a.prelu = g.prelu; a.accumulate = g.accumulate;
    a.tiles_h = msk_cdiv(a.H, 8); a.tiles_w = msk_cdiv(a.W, 16);
    const long cols = (long)a.N * a.tiles_h * a.tiles_w;
    const long per_cu = ctx->foldn_wgs > 0 ? ctx->foldn_wgs : 3;
    int segs = (int)((per_cu * ctx->num_cu + cols - 1) / cols);
    if (segs > a.D / 8) segs = a.D / 8;
    if (segs < 1) segs = 1;
    a.seg_len = msk_cdiv(a.D, segs);
    a.segs = msk_cdiv(a.D, a.seg_len);
    const long nblk = cols * a.segs;
    if (nblk > 0x7fffffff) return 0;
    a.nblk = (int)nblk;
    a.src_bytes = (unsigned)sb;
    a.x_amax = x_amax; a.w_amax = w_amax;
    const char* tag = "conv_foldn_h2";
    if (ctx->prof && ctx->prof_shapes) {
      char buf[160];
      snprintf(buf, sizeof(buf), "conv_foldn_h2[ck=%d,cn=%d,n=%d,dhw=%dx%dx%d,acc=%d]", g.CK, g.CN, g.N, g.DD, g.DH, g.DW, g.accumulate);
      tag = msk_intern_tag(ctx, buf);
    }
    msk_launch_scope ls(ctx, tag);
    hipLaunchKernelGGL(conv_foldn_h2_k, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, a);
    MSK_LAUNCH_CHECK(ctx);
    *act_fused = true;
    return 1;
  }
  float* wb = (float*)msk_workspace2(ctx, (size_t)26 * QG * 64 * 4 * sizeof(float));  // + one padding tap (prefetched, never used)
  if (!wb) return -1;
  {
    msk_launch_scope ls(ctx, "pack_weights_foldn");
    hipLaunchKernelGGL(pack_foldn_weights_k, dim3(25 * QG), dim3(256), 0, ctx->stream, w_canon, A, B, swap, g.transposed ? 1 : 0,
                       g.CK, g.CN, QG, wb);
    MSK_LAUNCH_CHECK(ctx);
  }
  FNArgs a{};
  a.src = g.src; a.sld = g.sld; a.dst = g.dst; a.dld = g.dld;
  a.N = g.N; a.D = g.DD; a.H = g.DH; a.W = g.DW; a.CN = g.CN;
  a.wb = (const float4*)wb; a.bias = g.bias; a.prelu = g.prelu; a.accumulate = g.accumulate;
  a.tiles_h = msk_cdiv(a.H, 8); a.tiles_w = msk_cdiv(a.W, 16);
  const long cols = (long)a.N * a.tiles_h * a.tiles_w;
  // D segments: enough workgroups for two per CU, each segment re-walks 4 planes
  const long per_cu = ctx->foldn_wgs > 0 ? ctx->foldn_wgs : 2;  // tuning: option "foldn_wgs"
  int segs = (int)((per_cu * ctx->num_cu + cols - 1) / cols);
  if (segs > a.D / 8) segs = a.D / 8;
  if (segs < 1) segs = 1;
  a.seg_len = msk_cdiv(a.D, segs);
  a.segs = msk_cdiv(a.D, a.seg_len);
  const long nblk = cols * a.segs;
  if (nblk > 0x7fffffff) return 0;
  a.nblk = (int)nblk;
  a.src_bytes = (unsigned)sb;
  const char* tag = "conv_foldn";
  if (ctx->prof && ctx->prof_shapes) {
    char buf[160];
    snprintf(buf, sizeof(buf), "conv_foldn[ck=%d,cn=%d,n=%d,dhw=%dx%dx%d,acc=%d]", g.CK, g.CN, g.N, g.DD, g.DH, g.DW, g.accumulate);
    tag = msk_intern_tag(ctx, buf);
  }
  msk_launch_scope ls(ctx, tag);
  hipLaunchKernelGGL((conv_foldn_k<Q>), dim3((unsigned)nblk), dim3(256), 0, ctx->stream, a);
  MSK_LAUNCH_CHECK(ctx);
  *act_fused = true;
  return 1;
}
