// Weight gradient of a 'same' 5^3 convolution with a TINY output-channel count (out_tr.conv1, 32 -> ncls <= 4,
// vnet.py:165):   dW[cb][ca][tap] = sum_u x[u][ca] * dy[u - tap + P][cb].
// The generic folded kernel (wgrad_fold_mfma_k<false>) gathers dy straight from global memory with one dword per lane
// and (tap, cb) pair: PMC MFMA pipe 41 % busy, 1.65 ms for 2 x 128^3.  Here, like wgrad_c1_mfma_k, the small-channel
// tensor lives in LDS: the dy halo of a 4 x 8 x 32 voxel tile as CB planes (CB x 13.8 KB), MFMA rows = (tap, cb) pairs
// (v_mfma_f32_16x16x4_f32: 16 pairs x 16 input channels x 4 voxels), the A operand is one ds_read_b32 at
// (voxel - tap offset) of the pair's plane, the B operand one coalesced dword of x (16 channels of 4 consecutive voxels).
// The four wavefronts of a workgroup own disjoint row tiles and all walk the whole voxel tile (x re-reads hit L1);
// workgroups are persistent and write one partial slab each.
#include "msk_conv.h"
#include "msk_wbf.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOOBc = 0xFFFFFFF0u;

__device__ __forceinline__ float cbs_load(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0));
}

// RTW = row tiles per wavefront, CT = column tiles (CA / 16)
template <int KS, int RTW, int CT>
__global__ void __launch_bounds__(256)
wgrad_cbs_mfma_k(WGrad g, int ntiles, int tiles_d, int tiles_h, int tiles_w, float* __restrict__ partial, unsigned a_bytes,
                 unsigned b_bytes) {
  constexpr int TD = 4, TH = 8, TW = 32, P = KS / 2;
  constexpr int HD = TD + 2 * P, HH = TH + 2 * P, HW = TW + 2 * P;
  constexpr int NV = HD * HH * HW, NVP = NV + 1;  // odd plane pitch
  constexpr int TAPS = KS * KS * KS;
  extern __shared__ float ys[];                    // [CB][NVP]

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 15, kq = lane >> 4;
  const int D = g.BD, H = g.BH, W = g.BW, CB = g.CB;
  const int Q = TAPS * CB;                         // (tap, cb) pairs, pair q = tap * CB + cb
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, b_bytes, 0x00020000);

  // this lane's pairs: LDS offset relative to the halo index of the CENTRE (u + P in every axis):
  //   plane(cb) * NVP - ((kd - P) * HH + (kh - P)) * HW - (kw - P)
  int po[RTW];
#pragma unroll
  for (int t = 0; t < RTW; ++t) {
    const int q = (wave * RTW + t) * 16 + r;
    if (q < Q) {
      const int tap = q / CB, cb = q - tap * CB;
      const int kd = tap / (KS * KS), kh = (tap / KS) % KS, kw = tap % KS;
      po[t] = cb * NVP - (((kd - P) * HH + (kh - P)) * HW + (kw - P));
    } else {
      po[t] = 0;                                   // padding pair: reads something valid, its rows are never stored
    }
  }

  f32x4 acc[RTW][CT];
#pragma unroll
  for (int t = 0; t < RTW; ++t)
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int t_ = tile;
    const int twi = t_ % tiles_w;
    t_ /= tiles_w;
    const int thi = t_ % tiles_h;
    t_ /= tiles_h;
    const int tdi = t_ % tiles_d;
    const int n = t_ / tiles_d;
    const int d0 = tdi * TD, h0 = thi * TH, w0 = twi * TW;
    __syncthreads();
    for (int hv = tid; hv < NV; hv += 256) {
      const int hd = hv / (HH * HW), rem = hv % (HH * HW), hh = rem / HW, hw = rem % HW;
      const int gd = d0 - P + hd, gh = h0 - P + hh, gw = w0 - P + hw;
      const bool in = (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
      const unsigned off = in ? (unsigned)((((n * D + gd) * H + gh) * W + gw) * g.bld) * 4u : kOOBc;
      for (int cb = 0; cb < CB; ++cb) ys[cb * NVP + hv] = cbs_load(rb, off == kOOBc ? kOOBc : off + 4u * cb);
    }
    __syncthreads();
    constexpr int UB = 8;
#pragma unroll 1
    for (int s0 = 0; s0 < TD * TH * (TW / 4); s0 += UB) {
      float xv[UB][CT];
      int ctr[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int s = s0 + u;
        const int dz = s / (TH * (TW / 4)), rem = s % (TH * (TW / 4)), h = rem / (TW / 4), w = (rem % (TW / 4)) * 4 + kq;
        const int gd = d0 + dz, gh = h0 + h, gw = w0 + w;
        const bool vok = gd < D && gh < H && gw < W;
        ctr[u] = ((dz + P) * HH + h + P) * HW + w + P;  // halo index of the centre of voxel u
        const unsigned xo = (unsigned)((((n * D + gd) * H + gh) * W + gw) * g.ald) * 4u;
#pragma unroll
        for (int c = 0; c < CT; ++c)
          xv[u][c] = cbs_load(ra, (vok && c * 16 + r < g.CA) ? xo + (unsigned)(c * 16 + r) * 4u : kOOBc);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u)
#pragma unroll
        for (int t = 0; t < RTW; ++t) {
          const float a = ys[ctr[u] + po[t]];
#pragma unroll
          for (int c = 0; c < CT; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv[u][c], acc[t][c], 0, 0, 0);
        }
    }
  }

  // D[row = 4*(lane >> 4) + j][col = lane & 15]: row = pair within the row tile, col = ca within the column tile
#pragma unroll
  for (int t = 0; t < RTW; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = (wave * RTW + t) * 16 + 4 * kq + j;
      if (q < Q) {
        const int tap = q / CB, cb = q - tap * CB;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const int ca = c * 16 + r;
          if (ca < g.CA) partial[(((long)blockIdx.x * TAPS + tap) * g.CA + ca) * CB + cb] = acc[t][c][j];
        }
      }
    }
}


// ---------------------------------------------------------------------------------------------------------
// The same weight gradient with fp16 two-piece operands (msk_wbf.h, option "conv_split" 2): x*sx = h + l, dy*sy = h + l
// in fp16, v_mfma_f32_16x16x32_f16 (16 (tap, cb) pairs x 16 input channels x 32 VOXELS), three MFMAs per fp32 product:
// 5.3x less matrix time than the eight v_mfma_f32_16x16x4_f32 the kernel above spends on the same 32 voxels.
//   * K = 32 consecutive voxels of a row; a lane's 8 K values are 8 consecutive voxels.  The A operand of pair (tap, cb)
//     is the dy row SHIFTED by the tap -- an arbitrary (odd) offset in fp16 units, which a 16-byte LDS read cannot
//     address.  LDS therefore holds, per piece and per cb plane, ONE DWORD PER VOXEL = {f16(v), f16(v + 1)}: the 8 values
//     starting at any voxel are the dwords v, v + 2, v + 4, v + 6, no shuffles, 8 B of LDS per value.  (The kernel
//     permutes K so that a lane owns the pairs v + 2 kq + {0, 8, 16, 24}: bank spread, see `ctr`.)
//   * one workgroup per CU, 512 registers per wavefront: every wavefront owns ALL row tiles (24 x 2 accumulators for
//     cb = 3) and its own plane of the voxel tile, so x is read and split into fp16 pieces exactly once; the next K step's
//     16 dwords of x are in flight while the current one's 144 MFMAs run.
//   * ds_read_b32 banks are (dword address) mod 32 within 32-lane groups: row pitch 37 and a cb-dependent plane pitch keep
//     the 16 pairs x 2 lane groups of a group on distinct banks (PMC: 63 % conflict cycles with the naive pitches).
//   * the four wavefronts' partial sums meet in LDS at the end (fixed order: bitwise reproducible), scaled back by
//     1/(sx*sy) (powers of two), one partial slab per workgroup.
typedef _Float16 cbs_f16x8 __attribute__((ext_vector_type(8)));

// RT row tiles of 16 pairs.  ALIGNED (cb = 3, RT = 25): tile = one (kd, kh), rows = kw * 3 + cb, row 15 idle -- the pairs of a
// tile then differ in kw and cb only and the bank picture above holds for every tile; otherwise pairs in natural order
// q = tap * CB + cb, RT = ceil(125 * CB / 16) (some 2-way conflicts where a tile crosses a kh row).
// NG wavefront groups of four share the row tiles (RT each): NG = 2 puts two wavefronts on every SIMD (256 registers each),
// whose MFMA streams fill each other's conversion / addressing / fill phases.
template <int RT, int NG, bool ALIGNED>
__global__ void __launch_bounds__(256 * NG, 1)
wgrad_cbs_h2_k(WGrad g, int ntiles, int tiles_d, int tiles_h, int tiles_w, float* __restrict__ partial, unsigned a_bytes,
               unsigned b_bytes, const float* __restrict__ x_amax, const float* __restrict__ y_amax, int NVP) {
  constexpr int KS = 5, TD = 4, TH = 8, TW = 32, P = 2;
  constexpr int HD = TD + 2 * P, HH = TH + 2 * P, HW = TW + 2 * P + 1;  // row pitch 37: a step in kh continues the kw sequence of banks
  constexpr int NV = HD * HH * HW;
  constexpr int TAPS = KS * KS * KS;
  extern __shared__ unsigned ysd[];  // [piece][CB][NVP] dwords {f16(v), f16(v + 1)}
  unsigned short* ys16 = reinterpret_cast<unsigned short*>(ysd);

  const int tid = threadIdx.x;
  const int wave = (tid >> 6) & 3, grp = tid >> 8, lane = tid & 63, r = lane & 15, kq = lane >> 4;
  const int D = g.BD, H = g.BH, W = g.BW, CB = g.CB;
  const int Q = TAPS * CB, PL = CB * NVP;
  constexpr int NT = 256 * NG;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, b_bytes, 0x00020000);
  const float sx = wbf_scale_of(x_amax), sy = wbf_scale_of(y_amax);

  int po[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    int tap, cb;
    const int tg = grp * RT + t;  // global row tile
    if (ALIGNED) {
      const int rr = r < 15 ? r : 0;  // idle row: the address of row 0 (broadcast)
      tap = tg < KS * KS ? tg * KS + rr / 3 : 0;
      cb = rr % 3;
    } else {
      const int q = tg * 16 + r;
      tap = q < Q ? q / CB : 0;  // padding pair: reads valid data, its rows are never stored
      cb = q < Q ? q - tap * CB : 0;
    }
    const int kd = tap / (KS * KS), kh = (tap / KS) % KS, kw = tap % KS;
    po[t] = cb * NVP - (((kd - P) * HH + (kh - P)) * HW + (kw - P));
  }
  f32x4 acc[RT][2];
#pragma unroll
  for (int t = 0; t < RT; ++t) acc[t][0] = acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bool c0 = r < g.CA, c1 = 16 + r < g.CA;
  // K step s of this wavefront = row (s & 7) of plane `wave` of the workgroup's (s >> 3)-th tile; its x operand is
  // requested TWO steps ahead, across tile boundaries (3 x 16 KiB per CU in flight)
  auto load_x = [&](int s, float (&xv)[2][8]) {
    const int tile = blockIdx.x + (s >> 3) * gridDim.x;
    int t_ = tile;
    const int twi = t_ % tiles_w;
    t_ /= tiles_w;
    const int thi = t_ % tiles_h;
    t_ /= tiles_h;
    const int tdi = t_ % tiles_d;
    const int n = t_ / tiles_d;
    const int gd = tdi * TD + wave, gh = thi * TH + (s & 7), w0 = twi * TW;
    const bool rok = tile < ntiles && gd < D && gh < H;
    if (w0 + TW <= W) {
      // full row: one per-lane base offset, the eight voxels through the scalar offset operand (no per-load address math)
      const unsigned xo = (unsigned)((((n * D + gd) * H + gh) * W + w0 + 2 * kq) * g.ald + r) * 4u;
      const unsigned b0 = (rok && c0) ? xo : kOOBc, b1 = (rok && c1) ? xo + 64u : kOOBc;
      const int vs = g.ald * 4;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int so = (8 * (j >> 1) + (j & 1)) * vs;
        xv[0][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, (int)b0, so, 0));
        xv[1][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, (int)b1, so, 0));
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int gw = w0 + 8 * (j >> 1) + 2 * kq + (j & 1);  // K permutation: see `ctr`
        const bool ok = rok && gw < W;
        const unsigned xo = (unsigned)((((n * D + gd) * H + gh) * W + gw) * g.ald + r) * 4u;
        xv[0][j] = cbs_load(ra, (ok && c0) ? xo : kOOBc);
        xv[1][j] = cbs_load(ra, (ok && c1) ? xo + 64u : kOOBc);
      }
    }
  };
  float x0[2][8], x1[2][8], x2[2][8];
  load_x(0, x0);
  load_x(1, x1);
  int sflat = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int t_ = tile;
    const int twi = t_ % tiles_w;
    t_ /= tiles_w;
    const int thi = t_ % tiles_h;
    t_ /= tiles_h;
    const int tdi = t_ % tiles_d;
    const int n = t_ / tiles_d;
    const int d0 = tdi * TD, h0 = thi * TH, w0 = twi * TW;
    const int gd = d0 + wave;
    __syncthreads();  // the previous tile's readers are done
    // dy halo -> LDS: all loads of a thread (14 voxels x CB) are in flight together, then split and stored
    constexpr int FB = (NV + NT - 1) / NT;
    {
      float fv[FB][4];
#pragma unroll
      for (int i = 0; i < FB; ++i) {
        const int hv = tid + NT * i;
        const int hd = hv / (HH * HW), rem = hv % (HH * HW), hh = rem / HW, hw = rem % HW;
        const int gdd = d0 - P + hd, gh = h0 - P + hh, gw = w0 - P + hw;
        const bool in = hv < NV && hw < HW - 1 && (unsigned)gdd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
        const unsigned off = (unsigned)((((n * D + gdd) * H + gh) * W + gw) * g.bld) * 4u;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) fv[i][cb] = cbs_load(rb, (in && cb < CB) ? off + 4u * cb : kOOBc);
      }
#pragma unroll
      for (int i = 0; i < FB; ++i) {
        const int hv = tid + NT * i;
        if (hv < NV) {
          const bool first = hv % HW == 0;
#pragma unroll
          for (int cb = 0; cb < 4; ++cb)
            if (cb < CB) {
              unsigned hi, lo;
              wbf_split2h_pair(fv[i][cb] * sy, 0.f, hi, lo);
              const int s = cb * NVP + hv;
              ys16[2 * s] = (unsigned short)hi;
              ys16[2 * (PL + s)] = (unsigned short)lo;
              if (!first) {
                ys16[2 * s - 1] = (unsigned short)hi;
                ys16[2 * (PL + s) - 1] = (unsigned short)lo;
              }
            }
        }
      }
    }
    __syncthreads();
    auto step = [&](int hs, float (&xv)[2][8]) {
      uint4 bh[2], bl[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        wbf_split2h_pair(xv[c][0] * sx, xv[c][1] * sx, bh[c].x, bl[c].x);
        wbf_split2h_pair(xv[c][2] * sx, xv[c][3] * sx, bh[c].y, bl[c].y);
        wbf_split2h_pair(xv[c][4] * sx, xv[c][5] * sx, bh[c].z, bl[c].z);
        wbf_split2h_pair(xv[c][6] * sx, xv[c][7] * sx, bh[c].w, bl[c].w);
      }
      // the lane group kq owns the voxel PAIRS kq, kq + 4, kq + 8, kq + 12 of the 32-voxel row (dwords +0, +8, +16, +24):
      // consecutive lane groups are two banks apart, and with the plane pitch (11 / 8 / 16 mod 32 for cb = 3 / 4 / 2) the 32
      // lanes of a ds_read_b32 group (16 pairs = a run of taps x cb, 2 lane groups) hit distinct banks or the same address
      const unsigned* ctr = ysd + ((wave + P) * HH + hs + P) * HW + 2 * kq + P;
      // row tiles go through the matrix pipe in PAIRS (four independent accumulators between two MFMAs of the same
      // accumulator: one wavefront per SIMD, a dependent MFMA would wait for its predecessor's result); the A fragments
      // of the next pair are requested before the MFMAs of this one are issued
      uint4 fh[4], fl[4];
      auto fetch = [&](int t, int slot) {
        const unsigned* ap = ctr + po[t];
        fh[slot] = uint4{ap[0], ap[8], ap[16], ap[24]};
        fl[slot] = uint4{ap[PL], ap[PL + 8], ap[PL + 16], ap[PL + 24]};
      };
      fetch(0, 0);
      fetch(1, 1);
#pragma unroll
      for (int t = 0; t < RT; t += 2) {
        const int nu = t + 1 < RT ? 2 : 1;  // RT odd: the last tile goes alone
        if (t + 2 < RT) fetch(t + 2, (t + 2) & 3);
        if (t + 3 < RT) fetch(t + 3, (t + 3) & 3);
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (u < nu)
#pragma unroll
            for (int c = 0; c < 2; ++c)
              acc[t + u][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cbs_f16x8, fl[(t + u) & 3]), __builtin_bit_cast(cbs_f16x8, bh[c]), acc[t + u][c], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (u < nu)
#pragma unroll
            for (int c = 0; c < 2; ++c)
              acc[t + u][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cbs_f16x8, fh[(t + u) & 3]), __builtin_bit_cast(cbs_f16x8, bl[c]), acc[t + u][c], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (u < nu)
#pragma unroll
            for (int c = 0; c < 2; ++c)
              acc[t + u][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cbs_f16x8, fh[(t + u) & 3]), __builtin_bit_cast(cbs_f16x8, bh[c]), acc[t + u][c], 0, 0, 0);
        // one wavefront per SIMD issues in order: the 8 LDS requests and their address arithmetic go INTO the gaps of
        // the MFMA stream (one per MFMA) instead of in front of it
        if (t + 3 < RT) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // VALU: the fragment addresses
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
            }
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
#pragma unroll 1
    for (int hs = 0; hs < TH; ++hs, ++sflat) {
      load_x(sflat + 2, x2);
      if (gd < D) step(hs, x0);
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          x0[c][j] = x1[c][j];
          x1[c][j] = x2[c][j];
        }
    }
  }

  // the four wavefronts' sums meet in LDS in a fixed order; D[row = 4*kq + j][col = r]
  const float ix = 1.f / sx, iy = 1.f / sy;
  float* red = reinterpret_cast<float*>(ysd);
  for (int wv = 0; wv < 4; ++wv) {
    __syncthreads();
    if (wave == wv) {
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        // the eight reads of a row tile are issued together (a read-add-write chain per value would run at LDS latency)
        float prev[2][4];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int j = 0; j < 4; ++j) prev[c][j] = wv == 0 ? 0.f : red[((grp * RT + t) * 16 + 4 * kq + j) * 32 + c * 16 + r];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            red[((grp * RT + t) * 16 + 4 * kq + j) * 32 + c * 16 + r] = prev[c][j] + acc[t][c][j] * ix * iy;
      }
    }
  }
  __syncthreads();
  const int total = TAPS * g.CA * CB;
  for (int o = tid; o < total; o += NT) {
    const int cb = o % CB, rest = o / CB, ca = rest % g.CA, tap = rest / g.CA;
    const int row = ALIGNED ? (tap / KS) * 16 + (tap % KS) * 3 + cb : tap * CB + cb;
    partial[(long)blockIdx.x * total + o] = red[row * 32 + ca];
  }
}

}  // namespace

// returns 1 when handled, 0 when not eligible, < 0 on error
int msk_wgrad_cbs(msk_ctx* ctx, const WGrad& g) {
  if (!(g.CB >= 1 && g.CB <= 4 && g.CA >= 8 && g.CA <= 32)) return 0;
  if (!(g.kd == 5 && g.kh == 5 && g.kw == 5 && g.sd == 1 && g.sh == 1 && g.sw == 1 && g.pd == 2 && g.ph == 2 && g.pw == 2))
    return 0;
  if (!(g.AD == g.BD && g.AH == g.BH && g.AW == g.BW)) return 0;
  const long M = (long)g.N * g.BD * g.BH * g.BW;
  const size_t abytes = (size_t)M * g.ald * sizeof(float), bbytes = (size_t)M * g.bld * sizeof(float);
  if (M >= (1L << 30) || abytes >= 0xFFFFFFF0ull || bbytes >= 0xFFFFFFF0ull) return 0;
  const int tiles_d = (g.BD + 3) / 4, tiles_h = (g.BH + 7) / 8, tiles_w = (g.BW + 31) / 32;
  const long ntiles = (long)g.N * tiles_d * tiles_h * tiles_w;
  if (ntiles > 0x7fffffff) return 0;
  const int taps = 125;
  // 26 = A/B: the fp32-MFMA kernel instead of the fp16 two-piece one
  // (cb = 4 would need 32 row tiles = 256 accumulator registers: it stays on the fp32 kernel)
  if (ctx->conv_split == 2 && ctx->conv_impl != 26 && g.CA >= 8 && g.CB <= 3 && !g.yfuse) {
    // max |x|: left in the xform header by the forward pass of the same layer (msk_conv3d_fwd_ex, conv_foldn_h2_k) when the
    // caller kept one; max |dy|: from the caller when it has it (msk_conv3d_bwd_bnact computes it once for both gradients)
    const float* x_amax = g.xform ? (const float*)g.xform : msk_absmax(ctx, g.A, g.ald, g.CA, M);
    const float* y_amax = g.b_amax ? g.b_amax : msk_absmax(ctx, g.B, g.bld, g.CB, M);
    if (!x_amax || !y_amax) return -1;
    const int rt = g.CB == 3 ? 25 : (taps * g.CB + 15) / 16;  // 8 / 16 / 25 (aligned) row tiles
    static const int kPitchMod[5] = {0, 0, 16, 11, 8};
    int nvp = 8 * 12 * 37;  // = 0 mod 32
    nvp += kPitchMod[g.CB];
    const size_t lds = (size_t)2 * g.CB * nvp * sizeof(unsigned);
    long splits = ctx->num_cu;  // one workgroup per CU (512 registers per wavefront)
    if (splits > ntiles) splits = ntiles;
    const size_t per = (size_t)taps * g.CA * g.CB * sizeof(float);
    float* partial = (float*)msk_workspace(ctx, (size_t)splits * per);
    if (!partial) return -1;
    {
      const char* tag = "wgrad_cbs_h2";
      if (ctx->prof && ctx->prof_shapes) {
        char buf[160];
        snprintf(buf, sizeof(buf), "wgrad_cbs_h2[ca=%d,cb=%d,M=%ld,splits=%ld]", g.CA, g.CB, M, splits);
        tag = msk_intern_tag(ctx, buf);
      }
      msk_launch_scope ls(ctx, tag);
      const dim3 grid((unsigned)splits), block(256);
#define MSK_CBSH(RT_, AL_)                                                                                                   \
  do {                                                                                                                  \
    static bool attr_set = false;                                                                                       \
    if (!attr_set) {                                                                                                    \
      (void)hipFuncSetAttribute((const void*)wgrad_cbs_h2_k<RT_, 1, AL_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr_set = true;                                                                                                  \
    }                                                                                                                   \
    hipLaunchKernelGGL((wgrad_cbs_h2_k<RT_, 1, AL_>), grid, block, lds, ctx->stream, g, (int)ntiles, tiles_d, tiles_h, tiles_w, \
                       partial, (unsigned)abytes, (unsigned)bbytes, x_amax, y_amax, nvp);                                    \
  } while (0)
      switch (rt) {
        case 8: MSK_CBSH(8, false); break;
        case 16: MSK_CBSH(16, false); break;
        default: MSK_CBSH(25, true); break;
      }
#undef MSK_CBSH
      MSK_LAUNCH_CHECK(ctx);
    }
    const int rc = msk_wgrad_reduce(ctx, partial, (int)splits, taps, g.CA, g.CB, g.dw, g.accumulate);
    return rc == 0 ? 1 : rc;
  }
  const int rtw = ((taps * g.CB + 15) / 16 + 3) / 4;  // row tiles per wavefront: 2 / 4 / 6 / 8 for CB = 1 .. 4
  const int ct = (g.CA + 15) / 16;
  const size_t lds = (size_t)g.CB * (8 * 12 * 36 + 1) * sizeof(float);  // CB planes of the 8 x 12 x 36 halo (+ the zero slot)
  long splits = 2L * ctx->num_cu;
  if (splits > ntiles) splits = ntiles;
  const size_t per = (size_t)taps * g.CA * g.CB * sizeof(float);
  float* partial = (float*)msk_workspace(ctx, (size_t)splits * per);
  if (!partial) return -1;
  {
    const char* tag = "wgrad_cbs_mfma";
    if (ctx->prof && ctx->prof_shapes) {
      char buf[160];
      snprintf(buf, sizeof(buf), "wgrad_cbs_mfma[ca=%d,cb=%d,M=%ld,splits=%ld]", g.CA, g.CB, M, splits);
      tag = msk_intern_tag(ctx, buf);
    }
    msk_launch_scope ls(ctx, tag);
    const dim3 grid((unsigned)splits), block(256);
#define MSK_CBS(RTW_, CT_)                                                                                              \
  hipLaunchKernelGGL((wgrad_cbs_mfma_k<5, RTW_, CT_>), grid, block, lds, ctx->stream, g, (int)ntiles, tiles_d, tiles_h, \
                     tiles_w, partial, (unsigned)abytes, (unsigned)bbytes)
    if (ct == 1) {
      switch (rtw) {
        case 2: MSK_CBS(2, 1); break;
        case 4: MSK_CBS(4, 1); break;
        case 6: MSK_CBS(6, 1); break;
        default: MSK_CBS(8, 1); break;
      }
    } else {
      switch (rtw) {
        case 2: MSK_CBS(2, 2); break;
        case 4: MSK_CBS(4, 2); break;
        case 6: MSK_CBS(6, 2); break;
        default: MSK_CBS(8, 2); break;
      }
    }
#undef MSK_CBS
    MSK_LAUNCH_CHECK(ctx);
  }
  const int rc = msk_wgrad_reduce(ctx, partial, (int)splits, taps, g.CA, g.CB, g.dw, g.accumulate);
  return rc == 0 ? 1 : rc;
}
