// Weight gradient of the 'same' K^3 convolutions, K = 5 (every LUConv layer, vnet.py:36; autograd of core/train.py:139)
// or K = 3 (UNet3D), on the 16-bit matrix pipe -- NP = 3: bf16 with fp32-exact operands, NP = 1: fp16 operands -- the adjoint of the three-stage pipeline of msk_conv_wbf.hip:
//
//   dU_xi[kd,kh][ci][co] = sum_{n,d,h,t} V_xi[n, d+kd-2, h+kh-2, t][ci] * Y_xi[n, d, h, t][co]
//   V = B^T x (wbf_tin_k<0>),  Y = A dy (wbf_tin_k<1>),  both split exactly into three bf16 pieces in HBM;
//   dw[co][ci][kd,kh,kw] (+)= sum_xi G[xi][kw] dU_xi[kd,kh][ci][co]          (wbf_wgrad_reduce_k, split-K slabs in order)
//
// The reduction runs over POSITIONS, so both MFMA operands need 8 consecutive positions of one channel per lane while
// the transformed tensors are position-major ([slot][8 channels]: the layout the forward pipeline stages with plain
// LDS-DMA).  gfx950's transposing LDS read does the conversion for free: ds_read_b64_tr_b16 lets each 16-lane group
// fetch a [4 positions][16 channels] block (lane i supplies the address of position i/4, channels 4(i%4)..+3) and
// hands lane i channel i of the 4 positions (verified on hardware: tools/probes/tr16_probe.hip).  Two reads give the
// 8-position fragment of v_mfma_f32_16x16x32_bf16 (M = 16 input channels, N = 16 output channels, K = 32 positions).
//
// Workgroup = 4 wavefronts sharing the 25 taps (7 + 6 + 6 + 6); one 16-channel chunk of ci x 32 output channels; the V
// halo tile (8+4) x (TH+4) and the Y tile 8 x TH of one (n, t) plane are staged once and serve all 25 taps.  A wavefront
// keeps (6..7 taps) x 2 (co halves) accumulators and walks the tile in K-steps of 32 positions; per K-step 48..54
// transposing reads feed 72..84 MFMAs.  The workgroup loops over its share of the position tiles (split-K over tiles, slabs reduced in
// a fixed order) and writes its partial dU once.  Six bf16 products per fp32 product, as in the forward pipeline.
#include "msk_wbf.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

struct WgArgs {
  const char* V;  // x side, chunk count KCA
  const char* Y;  // dy side, chunk count KCB
  float* P;       // partial dU: [xi][ks][kci][cob][tap][16][32]
  int N, T, KCA, KCB, DP, HP;
  int tiles_d, tiles_h, ntiles, ksplit, tiles_per;
  int ncob;
  long v_xi, y_xi, plane;  // bytes
  // NP = 2: per-channel renormalisation of the dy side (msk_wbf.h: wbf_chan_shift): y_cmax [CB] = max |dy| per channel,
  // y_amax = the amax array the Y transform was scaled by.  Either null: no shifts.
  const float *y_cmax, *y_amax;
};

__device__ __forceinline__ s16x4 tr_read(const char* lds_base, unsigned byte_off) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_base + byte_off));
}

#define WGW_MFMA(acc, av, bv) \
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc, 0, 0, 0)
#define WGW_MFMA_H(acc, av, bv) \
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bv), acc, 0, 0, 0)

constexpr int wg_nxi(int K) { return K == 5 ? 8 : 6; }

#ifndef WGW_LB
#define WGW_LB 3
#endif
#ifndef WGW_PIPE
#define WGW_PIPE 0
#endif
template <int TH, int K, int NP>
#if WGW_LB > 0
__global__ void __launch_bounds__(256, WGW_LB)   // three wavefronts per SIMD (<= 168 registers): two lose 8 % (round 2)
#else
__global__ void __launch_bounds__(256)
#endif
wbf_wgrad_k(WgArgs a) {
  constexpr int NXI = wg_nxi(K), T2 = K * K, PADK = (K - 1) / 2, NPL = 2 * NP;
  constexpr int NT0 = K == 5 ? 7 : 3, NT1 = K == 5 ? 6 : 2;            // taps of wavefront 0 / of the others
  constexpr int TD = 8, HPt = TH + K - 1, NSV = (TD + K - 1) * HPt, NSY = TD * TH;
  constexpr int RV1 = (NSV + 63) / 64, RY1 = (NSY + 63) / 64;           // LDS-DMA rounds per plane
  constexpr int PSV = RV1 * 1024 + (TH == 16 ? 64 : 128);               // plane strides: the pad keeps the two khalf planes
  constexpr int PSY = RY1 * 1024 + 64;                                  //   of a transposing read on disjoint banks
  constexpr int YBASE = NPL * PSV;
  constexpr int NRV = NPL * RV1, NRY = 2 * NPL * RY1, NR = NRV + NRY, RPW = (NR + 3) / 4;  // rounds per wavefront
  constexpr int KSTEPS = TD * TH / 32, ROWS_PER_STEP = 32 / TH;
  __shared__ __attribute__((aligned(16))) char lds[NPL * PSV + 2 * NPL * PSY];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int g = lane >> 4, i = lane & 15;

  int b = blockIdx.x;
  const int xi = b % NXI;
  b /= NXI;
  const int cob = b % a.ncob;
  b /= a.ncob;
  const int kci = b % a.KCA;
  const int ks = b / a.KCA;

  const __amdgpu_buffer_rsrc_t vres =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.V + (long)xi * a.v_xi), 0, 0xFFFFFFF0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t yres =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.Y + (long)xi * a.y_xi), 0, 0xFFFFFFF0u, 0x00020000);

  // LDS-DMA rounds of this wavefront: R = wave + 4 j.  V rounds first (plane pk, 64 slots each), then Y rounds
  // (chunk c, plane pk).  voff = byte offset inside the (n, t) block of the tensor, relative to the tile origin.
  unsigned voff[RPW];
  int ldst[RPW];
#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    const int R = wave + 4 * j;
    if (R < NRV) {
      const int pk = R / RV1, sub = R - pk * RV1;
      int slot = sub * 64 + lane;
      if (slot >= NSV) slot = 0;  // overshoot of the last round lands in the plane's pad
      const int row = slot / HPt, col = slot - row * HPt;
      voff[j] = (unsigned)(((long)kci * NPL + pk) * a.plane + ((long)(row + 2 - PADK) * a.HP + col + 2 - PADK) * 16);
      ldst[j] = pk * PSV + sub * 1024;
    } else {
      const int Ry = R - NRV;
      const int cp = Ry / RY1, sub = Ry - cp * RY1;  // cp = c*NPL + pk
      int slot = sub * 64 + lane;
      if (slot >= NSY) slot = 0;
      const int row = slot / TH, col = slot - row * TH;
      voff[j] = (unsigned)(((long)cob * 2 * NPL + cp) * a.plane + ((long)(row + 2) * a.HP + col + 2) * 16);
      ldst[j] = YBASE + cp * PSY + sub * 1024;
    }
  }

  // per-lane LDS byte offsets of the transposing reads (position q = 8 g + 4 r + i/4 of the K-step, channel quad i%4)
  const int q0 = 8 * g + (i >> 2);
  const int khalf = (i & 3) >> 1, sub8 = (i & 1) * 8;
  const unsigned va = (unsigned)(khalf * PSV + ((q0 / TH) * HPt + (q0 % TH)) * 16 + sub8);
  // taps of this wavefront: [tap0, tap0 + ntap) of the 25 (kd, kh) taps -- 7, 6, 6, 6.  (Five wavefronts, one kd row each,
  // balanced the taps but not the SIMDs: a 5-wave workgroup takes 2 slots on one SIMD, the third resident workgroup often
  // found no SIMD with room, and PMC showed half the expected wavefronts in flight and the matrix pipe 52 % busy.)
  const int tap0 = wave == 0 ? 0 : NT0 - NT1 + NT1 * wave, ntap = wave == 0 ? NT0 : NT1;
  unsigned tapoff[NT0];
#pragma unroll
  for (int j = 0; j < NT0; ++j) {
    const int tp = min(tap0 + j, T2 - 1);
    tapoff[j] = (unsigned)(((tp / K) * HPt + (tp % K)) * 16);   // wave-uniform
  }
  const unsigned ya = (unsigned)(YBASE + khalf * PSY + ((q0 / TH) * TH + (q0 % TH)) * 16 + sub8);

  f32x4 acc[NT0][2];
#pragma unroll
  for (int j = 0; j < NT0; ++j)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[j][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // NP = 2, round 4: both in-register factors 2^-11 of the cross terms sit on the B (dy) side -- (lx' * (hy 2^-11)) and
  // (hx * (ly' 2^-11)) -- and the B fragments of output channel co = cob*32 + c*16 + i (this lane's) are first multiplied by
  // 2^shift(co) = floor(max |dy| / max |dy[co]|) (msk_wbf.h: wbf_chan_shift; exact: the result stays below the tensor's own
  // ceiling), undone when the sums are stored.  The down-scaled pieces then underflow only for ELEMENTS 2^12 below their own
  // channel's maximum, whose absolute error (2^-25 of the channel's ceiling per product) is far below the fp32 rounding of the
  // sum; the A (x) side is used as stored, so a quiet input channel costs nothing either.  Round 3 scaled per tensor only and
  // lost one bit per factor of two for channels more than 2^13 below the loudest.
  int shb[2] = {0, 0};
  if (NP == 2 && a.y_cmax && a.y_amax) {
    const float ya_max = wbf_amax_of(a.y_amax);
    shb[0] = wbf_chan_shift(ya_max, a.y_cmax[cob * 32 + i]);
    shb[1] = wbf_chan_shift(ya_max, a.y_cmax[cob * 32 + 16 + i]);
  }
  const _Float16 mb[2] = {(_Float16)ldexpf(1.f, shb[0]), (_Float16)ldexpf(1.f, shb[1])};
  const _Float16 mb_dn[2] = {(_Float16)ldexpf(1.f, shb[0] - 11), (_Float16)ldexpf(1.f, shb[1] - 11)};
  auto fmul = [](s16x8 v, _Float16 m) { return __builtin_bit_cast(s16x8, __builtin_bit_cast(f16x8, v) * m); };

  const int t0 = ks * a.tiles_per;
  const int t1 = min(a.ntiles, t0 + a.tiles_per);
  const int tiles_nt = a.tiles_d * a.tiles_h;
  for (int tile = t0; tile < t1; ++tile) {
    const int nt = tile / tiles_nt, rem = tile - nt * tiles_nt;  // nt = n*T + t
    const int tdi = rem / a.tiles_h, thi = rem - tdi * a.tiles_h;
    const unsigned origin = (unsigned)(((long)(tdi * TD) * a.HP + thi * TH) * 16);
    const unsigned vso = (unsigned)((long)nt * a.KCA * NPL * a.plane) + origin;
    const unsigned yso = (unsigned)((long)nt * a.KCB * NPL * a.plane) + origin;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int R = wave + 4 * j;
      if (R < NRV)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(vres, (__attribute__((address_space(3))) void*)(lds + ldst[j]), 16,
                                                 (int)voff[j], (int)vso, 0, 0);
      else if (R < NR)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(yres, (__attribute__((address_space(3))) void*)(lds + ldst[j]), 16,
                                                 (int)voff[j], (int)yso, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // B (dy) fragments of one K-step: high piece times 2^shift, and both pieces times 2^(shift - 11)
    auto load_b = [&](int kst, s16x8 (&bq0)[2], s16x8 (&bq1)[2], s16x8 (&bq2)[2], s16x8 (&bdh)[2], s16x8 (&bdl)[2]) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        s16x8 t[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const unsigned o = ya + (unsigned)((c * NPL + p * 2) * PSY + kst * ROWS_PER_STEP * TH * 16);
          const s16x4 lo4 = tr_read(lds, o), hi4 = tr_read(lds, o + 64);
          t[p] = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
        }
        bq0[c] = t[0]; bq1[c] = t[NP / 2]; bq2[c] = t[NP - 1];
        bdh[c] = t[0]; bdl[c] = t[0];
        if (NP == 2) {
          bdh[c] = fmul(t[0], mb_dn[c]);
          bdl[c] = fmul(t[NP - 1], mb_dn[c]);
          bq0[c] = fmul(t[0], mb[c]);
        }
      }
    };
#if WGW_PIPE
    s16x8 nb0[2], nb1[2], nb2[2], nbh[2], nbl[2];
    load_b(0, nb0, nb1, nb2, nbh, nbl);
#endif
#pragma unroll
    for (int kst = 0; kst < KSTEPS; ++kst) {
      s16x8 bq[2][3], bdh[2], bdl[2];   // bq[c][0 / 1 / 2] = high / middle (NP = 3) / last piece
#if WGW_PIPE
#pragma unroll
      for (int c = 0; c < 2; ++c) { bq[c][0] = nb0[c]; bq[c][1] = nb1[c]; bq[c][2] = nb2[c]; bdh[c] = nbh[c]; bdl[c] = nbl[c]; }
      if (kst + 1 < KSTEPS) load_b(kst + 1, nb0, nb1, nb2, nbh, nbl);
#else
      {
        s16x8 t0[2], t1[2], t2[2];
        load_b(kst, t0, t1, t2, bdh, bdl);
#pragma unroll
        for (int c = 0; c < 2; ++c) { bq[c][0] = t0[c]; bq[c][1] = t1[c]; bq[c][2] = t2[c]; }
      }
#endif
#pragma unroll
      for (int j = 0; j < NT0; ++j) {
        if (j < ntap) {   // wave-uniform
          s16x8 aq[NP];
          const unsigned vt = va + tapoff[j];
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            const unsigned o = vt + (unsigned)(p * 2 * PSV + kst * ROWS_PER_STEP * HPt * 16);
            const s16x4 lo4 = tr_read(lds, o), hi4 = tr_read(lds, o + 64);
            aq[p] = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
          }
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            if (NP == 3) {
              WGW_MFMA(acc[j][c], aq[NP - 1], bq[c][0]);  // small terms first
              WGW_MFMA(acc[j][c], aq[0], bq[c][2]);
              WGW_MFMA(acc[j][c], aq[NP / 2], bq[c][1]);
              WGW_MFMA(acc[j][c], aq[NP / 2], bq[c][0]);
              WGW_MFMA(acc[j][c], aq[0], bq[c][1]);
              WGW_MFMA(acc[j][c], aq[0], bq[c][0]);
            } else if (NP == 2) {
              WGW_MFMA_H(acc[j][c], aq[NP - 1], bdh[c]);  // small terms first
              WGW_MFMA_H(acc[j][c], aq[0], bdl[c]);
              WGW_MFMA_H(acc[j][c], aq[0], bq[c][0]);
            } else {
              WGW_MFMA_H(acc[j][c], aq[0], bq[c][0]);
            }
          }
        }
      }
    }
  }

  // partial dU[xi][ks][kci][cob][tap = tap0 + j][ci = 4 g + reg][co = c*16 + i]
  float* pb = a.P + (((((long)xi * a.ksplit + ks) * a.KCA + kci) * a.ncob + cob) * T2 + tap0) * 512;
  const float undo[2] = {ldexpf(1.f, -shb[0]), ldexpf(1.f, -shb[1])};   // element (ci = 4 g + r, co = c*16 + i): this lane's co
#pragma unroll
  for (int j = 0; j < NT0; ++j)
    if (j < ntap) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) pb[j * 512 + (4 * g + r) * 32 + c * 16 + i] = acc[j][c][r] * undo[c];
    }
}

// dw[cb][ca][canonical tap(kd, kh, kw)] (+)= sum_xi G[xi][kw] * sum_ks P[xi][ks][ca/16][cb/32][kd*5+kh][ca%16][cb%32]
// Block = 32 consecutive outputs (one (row, ca), 32 cb: 128-byte lines of every slab) x 8 slices of the split-K slabs; a
// thread adds its slices z = slice, slice + 8, ... in double, the 8 slices are combined through LDS in a fixed order
// (deterministic).  (One thread per output walking all 8 * ksplit slabs alone took 0.87 ms per step: 25 K threads with
// 384 dependent loads each for the 32-channel layers.)
template <int K>
__global__ void __launch_bounds__(256)
wbf_wgrad_reduce_k(const float* __restrict__ P, int ksplit, int NS, int KCA, int ncob, int CA, int CB, int tsd, int tsh,
                   int tsw, float* __restrict__ dw, int accumulate, const float* __restrict__ y_amax,
                   const float* __restrict__ v_amax, int scaled) {
  constexpr int NXI = wg_nxi(K), T2 = K * K, T3 = K * K * K;
  const double G5[8][5] = {{-1, 0, 0, 0, 0},
                           {-2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9},
                           {-2.0 / 9, 2.0 / 9, -2.0 / 9, 2.0 / 9, -2.0 / 9},
                           {1.0 / 90, 1.0 / 45, 2.0 / 45, 4.0 / 45, 8.0 / 45},
                           {1.0 / 90, -1.0 / 45, 2.0 / 45, -4.0 / 45, 8.0 / 45},
                           {32.0 / 45, 16.0 / 45, 8.0 / 45, 4.0 / 45, 2.0 / 45},
                           {32.0 / 45, -16.0 / 45, 8.0 / 45, -4.0 / 45, 2.0 / 45},
                           {0, 0, 0, 0, 1}};
  const double G3[6][3] = {{0.25, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                           {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
  __shared__ double sh[8][8][32];  // [sub * NS + slice][xi][cb lane]
  const double oscale = scaled ? 1.0 / ((double)wbf_scale_of(y_amax) * (double)wbf_scale_of(v_amax)) : 1.0;  // NP = 2: the operands' power-of-two scales
  // NS = slices of the split-K range per output group (1, 2, 4 or 8: <= ksplit), SUB = 8 / NS groups per block
  const int lane = threadIdx.x & 31, w8 = threadIdx.x >> 5, slice = w8 % NS, sub = w8 / NS, SUB = 8 / NS;
  const long groups = (long)T2 * CA * ncob;         // (row, ca, cob)
  const long slab = (long)KCA * ncob * T2 * 512;    // floats per (xi, ks)
  const long rounds = (groups + SUB - 1) / SUB;
  for (long ri = blockIdx.x; ri < rounds; ri += gridDim.x) {
    const long gi = ri * SUB + sub;
    const bool live = gi < groups;
    const int cob = (int)(gi % ncob);
    long r_ = gi / ncob;
    const int ca = (int)(r_ % CA);
    const int row = (int)(r_ / CA);  // kd*K + kh
    const long off = ((((long)(ca >> 4) * ncob + cob) * T2 + row) * 16 + (ca & 15)) * 32 + lane;
    {
      // the NXI loads of a slab are issued together (one xi after the other left every load waiting for the one before: the
      // additions of a point are a dependent chain); the order of the additions per point is unchanged
      double acc[NXI];
#pragma unroll
      for (int xi = 0; xi < NXI; ++xi) acc[xi] = 0.0;
      if (live) {
        const float* p = P + off;
        const long xstep = (long)ksplit * slab;
        for (int z = slice; z < ksplit; z += NS) {
          float v[NXI];
#pragma unroll
          for (int xi = 0; xi < NXI; ++xi) v[xi] = p[(long)xi * xstep + (long)z * slab];
#pragma unroll
          for (int xi = 0; xi < NXI; ++xi) acc[xi] += (double)v[xi];
        }
      }
#pragma unroll
      for (int xi = 0; xi < NXI; ++xi) sh[w8][xi][lane] = acc[xi];
    }
    __syncthreads();
    if (slice == 0 && live) {
      double s[NXI];
#pragma unroll
      for (int xi = 0; xi < NXI; ++xi) {
        double t = sh[w8][xi][lane];
        for (int k = 1; k < NS; ++k) t += sh[w8 + k][xi][lane];  // fixed order
        s[xi] = t;
      }
      const int cb = cob * 32 + lane;
      float* o = dw + ((long)cb * CA + ca) * T3 + (row / K) * tsd + (row % K) * tsh;
#pragma unroll
      for (int kw = 0; kw < K; ++kw) {
        double v = 0.0;
#pragma unroll
        for (int xi = 0; xi < NXI; ++xi) v += (K == 5 ? G5[xi][kw] : G3[xi][kw]) * s[xi];
        v *= oscale;
        float* q = o + kw * tsw;
        *q = accumulate ? *q + (float)v : (float)v;
      }
    }
    __syncthreads();
  }
}

// Round 5 -- the same reduction for the deep layers (>= 128 input channels, <= 4 slabs), one workgroup per (ca, 32-cb group):
// the form above hands a wavefront's 32 lanes 32 DIFFERENT cb rows of dw[cb][ca][125] -- 20-byte pieces at a stride of
// CA * 500 bytes (1.1 TB/s on the 33 MB tensors of the 256-channel layers: 77-80 us per launch, four launches per step).  Here
// the 8 half-wavefronts share the 25 (kd, kh) rows, the 5 kw values of every row go to an LDS tile [32 cb][125 taps] and the
// workgroup writes 32 runs of 500 contiguous bytes.  Slabs are added in order, in double (as above).
template <int K>
__global__ void __launch_bounds__(256)
wbf_wgrad_reduce_rows_k(const float* __restrict__ P, int ksplit, int KCA, int ncob, int CA, int CB, int tsd, int tsh, int tsw,
                        float* __restrict__ dw, int accumulate, const float* __restrict__ y_amax, const float* __restrict__ v_amax,
                        int scaled) {
  constexpr int NXI = wg_nxi(K), T2 = K * K, T3 = K * K * K, PITCH = T3 + 4;   // 129 floats: odd in dwords -> conflict-free column writes
  const double G5[8][5] = {{-1, 0, 0, 0, 0},
                           {-2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9},
                           {-2.0 / 9, 2.0 / 9, -2.0 / 9, 2.0 / 9, -2.0 / 9},
                           {1.0 / 90, 1.0 / 45, 2.0 / 45, 4.0 / 45, 8.0 / 45},
                           {1.0 / 90, -1.0 / 45, 2.0 / 45, -4.0 / 45, 8.0 / 45},
                           {32.0 / 45, 16.0 / 45, 8.0 / 45, 4.0 / 45, 2.0 / 45},
                           {32.0 / 45, -16.0 / 45, 8.0 / 45, -4.0 / 45, 2.0 / 45},
                           {0, 0, 0, 0, 1}};
  const double G3[6][3] = {{0.25, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                           {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
  __shared__ float tile[32 * PITCH];
  const double oscale = scaled ? 1.0 / ((double)wbf_scale_of(y_amax) * (double)wbf_scale_of(v_amax)) : 1.0;
  const int lane = threadIdx.x & 31, w8 = threadIdx.x >> 5;
  const int cob = blockIdx.x % ncob, ca = blockIdx.x / ncob;
  const long slab = (long)KCA * ncob * T2 * 512;    // floats per (xi, ks)
  const long xstep = (long)ksplit * slab;
  for (int row = w8; row < T2; row += 8) {
    const float* p = P + ((((long)(ca >> 4) * ncob + cob) * T2 + row) * 16 + (ca & 15)) * 32 + lane;
    double acc[NXI];
#pragma unroll
    for (int xi = 0; xi < NXI; ++xi) acc[xi] = 0.0;
    for (int z = 0; z < ksplit; ++z) {
      float v[NXI];
#pragma unroll
      for (int xi = 0; xi < NXI; ++xi) v[xi] = p[(long)xi * xstep + (long)z * slab];
#pragma unroll
      for (int xi = 0; xi < NXI; ++xi) acc[xi] += (double)v[xi];
    }
    float* o = tile + lane * PITCH + (row / K) * tsd + (row % K) * tsh;
#pragma unroll
    for (int kw = 0; kw < K; ++kw) {
      double v = 0.0;
#pragma unroll
      for (int xi = 0; xi < NXI; ++xi) v += (K == 5 ? G5[xi][kw] : G3[xi][kw]) * acc[xi];
      o[kw * tsw] = (float)(v * oscale);
    }
  }
  __syncthreads();
  float* base = dw + ((long)cob * 32 * CA + ca) * T3;
  for (int i = threadIdx.x; i < 32 * T3; i += 256) {
    const int cb = i / T3, tap = i - cb * T3;
    float* q = base + (long)cb * CA * T3 + tap;
    const float v = tile[cb * PITCH + tap];
    *q = accumulate ? *q + v : v;
  }
}

}  // namespace

namespace {
template <int K, int NP>
int run_wgrad_pipeline(msk_ctx* ctx, const WGrad& g, const WbfGeom& geo, bool shared_geom, int TH) {
  constexpr int NXI = wg_nxi(K), T2 = K * K, NPL = 2 * NP;
  const int* pm = geo.perm;
  const int LD = geo.LD, LH = geo.LH, LW = geo.LW;
  const int vstr[3] = {g.BH * g.BW, g.BW, 1};
  const int tstr[3] = {K * K, K, 1};
  const int T = geo.T, KCA = g.CA / 16, KCB = g.CB / 16, ncob = g.CB / 32;
  const int tiles_d = (LD + 7) / 8, tiles_h = (LH + TH - 1) / TH;
  const int DP = geo.DP, HP = geo.HP;
  const long ntiles = (long)g.N * T * tiles_d * tiles_h;
  if (ntiles > 0x7fffffffL) return 0;

  // split K (position tiles): about 3 workgroups per CU in total
  const long base_blocks = (long)NXI * KCA * ncob;
  // split-K target in workgroups per CU (tuning: option "wgrad_wino_rounds").  3 fill the kernel's own occupancy.  2 (round 5) leave a
  // third of the registers to the compute stream's kernels beside it and a third fewer slabs to the reduce: step -0.08 ms in five of
  // five interleaved pairs (profiles/r05_ab_runs.txt) -- but the kernel then runs longer NEXT TO the data-gradient GEMMs, whose
  // in-step time grows by 5 % (0.252 -> 0.266 ms per launch): the dominant kernel's measured rate would pay for 0.4 % of the step.
  // Default stays 3; 1: +1.0 ms
  const long per_cu = ctx->wgrad_wino_rounds > 0 ? ctx->wgrad_wino_rounds : 3;
  long ksplit = (per_cu * ctx->num_cu + base_blocks / 2) / base_blocks;
  if (ksplit < 1) ksplit = 1;
  if (ksplit > ntiles) ksplit = ntiles;
  const int tiles_per = (int)((ntiles + ksplit - 1) / ksplit);
  ksplit = (ntiles + tiles_per - 1) / tiles_per;

  const size_t plane = (size_t)DP * HP * 16;
  const size_t v_xi = (size_t)g.N * T * KCA * NPL * plane, y_xi = (size_t)g.N * T * KCB * NPL * plane;
  if (v_xi >= 0xFFFFFFF0ull || y_xi >= 0xFFFFFFF0ull) return 0;
  const size_t p_floats = (size_t)NXI * ksplit * KCA * ncob * T2 * 512;
  const size_t vb = (NXI * v_xi + 255) & ~(size_t)255, yb = (NXI * y_xi + 255) & ~(size_t)255;
  const bool have_v = g.xform != nullptr && shared_geom &&
                      msk_wbf_xform_bytes(g.N, g.AD, g.AH, g.AW, g.CA, g.CB, K, NP) == NXI * v_xi + kWbfXformHeader;
  const bool have_y = g.yform != nullptr;  // A dy already written by the dual transform (msk_conv3d_bwd_bnact checked the geometry)
  if ((have_y || g.yfuse) && !(have_v && shared_geom)) return 0;
  char* wsp = (char*)msk_workspace(ctx, (have_v ? 0 : vb) + (have_y ? 0 : yb) + p_floats * sizeof(float) + 256);
  if (!wsp) return -1;
  char* V = have_v ? (char*)const_cast<void*>(g.xform) + kWbfXformHeader : wsp;
  const float* v_amax = nullptr;  // NP = 2: the scale of the A operand (from the xform header, or computed here)
  if (NP != 3 && have_v) v_amax = (const float*)g.xform;
  char* Y = have_y ? (char*)const_cast<void*>(g.yform) : (have_v ? wsp : wsp + vb);
  float* P = (float*)(have_y ? (have_v ? wsp : wsp + vb) : Y + yb);

  WbfTinArgs ta{};
  ta.src = g.A; ta.sld = g.ald;
  ta.svn = (long)g.BD * g.BH * g.BW; ta.svd = vstr[pm[0]]; ta.svh = vstr[pm[1]]; ta.svw = vstr[pm[2]];
  ta.N = g.N; ta.LD = LD; ta.LH = LH; ta.LW = LW; ta.T = T; ta.CK = g.CA; ta.KC = KCA;
  ta.DP = DP; ta.HP = HP; ta.V = V; ta.v_xi = (long)v_xi;
  // NP = 2: per-channel maxima of the dy side (msk_wbf.h: wbf_chan_shift), folded in by whichever kernel writes Y
  const float* y_cmax = nullptr;
  const int cmax_slots_b = (g.CB + kWbfAmaxWays - 1) / kWbfAmaxWays;
  if (!have_v) {  // else: V written by msk_conv3d_fwd_ex for this tensor
    if (NP != 3) {
      v_amax = g.a_amax ? g.a_amax : msk_absmax(ctx, g.A, g.ald, g.CA, (long)g.N * g.AD * g.AH * g.AW);   // the caller may hold it
      if (!v_amax) return -1;
      ta.amax = v_amax;
    }
    if (msk_wbf_transform(ctx, 0, K, NP, ta) != 0) return -1;
  }
  ta.src = g.B; ta.sld = g.bld; ta.CK = g.CB; ta.KC = KCB; ta.V = Y; ta.v_xi = (long)y_xi;
  const float* y_amax = nullptr;  // NP = 2: the gradient operand is scaled by a power of two (fp16 range), undone in the reduce
  if (g.yfuse) {
    WbfBnBwd bn = *g.yfuse;
    bn.Y = Y;
    bn.y_xi = (long)y_xi;
    y_amax = bn.amax;
    if (NP == 2) {
      bn.y_cmax = msk_scalar_slots(ctx, cmax_slots_b);
      if (!bn.y_cmax) return -1;
      y_cmax = bn.y_cmax;
    }
    if (msk_wbf_transform_dual(ctx, K, NP, ta, bn, false) != 0) return -1;
  } else if (have_y) {
    y_amax = g.y_amax;
    y_cmax = NP == 2 ? g.y_cmax : nullptr;
  } else {
    if (NP != 3) {
      y_amax = g.b_amax ? g.b_amax : msk_absmax(ctx, g.B, g.bld, g.cb_real > 0 ? g.cb_real : g.CB, (long)g.N * g.BD * g.BH * g.BW);   // the caller may hold it (amax array)
      if (!y_amax) return -1;
    }
    ta.amax = y_amax;
    ta.c_real = g.cb_real;
    if (NP == 2) {
      ta.cmax = msk_scalar_slots(ctx, cmax_slots_b);
      if (!ta.cmax) return -1;
      y_cmax = ta.cmax;
    }
    if (msk_wbf_transform(ctx, 1, K, NP, ta) != 0) return -1;
  }

  WgArgs wa{};
  wa.V = V; wa.Y = Y; wa.P = P;
  wa.N = g.N; wa.T = T; wa.KCA = KCA; wa.KCB = KCB; wa.DP = DP; wa.HP = HP;
  wa.tiles_d = tiles_d; wa.tiles_h = tiles_h; wa.ntiles = (int)ntiles; wa.ksplit = (int)ksplit; wa.tiles_per = tiles_per;
  wa.ncob = ncob; wa.v_xi = (long)v_xi; wa.y_xi = (long)y_xi; wa.plane = (long)plane;
  if (NP == 2 && ctx->wgrad_renorm) { wa.y_cmax = y_cmax; wa.y_amax = y_amax; }
  const long nblk = base_blocks * ksplit;
  {
    const char* tag = NP == 3 ? "wbf_wgrad_k" : (NP == 2 ? "wbf_wgrad_h2_k" : "wbf_wgrad_f16_k");
    if (ctx->prof && ctx->prof_shapes) {
      char buf[200];
      snprintf(buf, sizeof(buf), "%s[ca=%d,cb=%d,n=%d,dhw=%dx%dx%d,k=%d,ks=%ld]", tag, g.CA, g.CB, g.N, g.BD, g.BH, g.BW, K, ksplit);
      tag = msk_intern_tag(ctx, buf);
    }
    msk_launch_scope ls(ctx, tag);
    const unsigned pad = (unsigned)ctx->wgrad_lds_pad;
    if (TH == 16) hipLaunchKernelGGL((wbf_wgrad_k<16, K, NP>), dim3((unsigned)nblk), dim3(256), pad, ctx->stream, wa);
    else hipLaunchKernelGGL((wbf_wgrad_k<8, K, NP>), dim3((unsigned)nblk), dim3(256), pad, ctx->stream, wa);
    MSK_LAUNCH_CHECK(ctx);
  }
  {
    msk_launch_scope ls(ctx, "wbf_wgrad_reduce");
    if (ctx->wgrad_reduce_rows && ksplit <= 4 && (long)g.CA * ncob >= 2L * ctx->num_cu && g.CB % 32 == 0) {
      // deep layers: one workgroup per (ca, cb group), 500-byte runs of dw (wbf_wgrad_reduce_rows_k)
      hipLaunchKernelGGL((wbf_wgrad_reduce_rows_k<K>), dim3((unsigned)(g.CA * ncob)), dim3(256), 0, ctx->stream, (const float*)P, (int)ksplit,
                         KCA, ncob, g.CA, g.CB, tstr[pm[0]], tstr[pm[1]], tstr[pm[2]], g.dw, g.accumulate, y_amax, v_amax, NP != 3 ? 1 : 0);
      MSK_LAUNCH_CHECK(ctx);
      return 1;
    }
    const int NS = ksplit >= 8 ? 8 : (ksplit >= 4 ? 4 : (ksplit >= 2 ? 2 : 1));
    long blocks = ((long)T2 * g.CA * ncob + (8 / NS) - 1) / (8 / NS);
    if (blocks > 32L * ctx->num_cu) blocks = 32L * ctx->num_cu;
    hipLaunchKernelGGL((wbf_wgrad_reduce_k<K>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const float*)P, (int)ksplit, NS,
                       KCA, ncob, g.CA, g.CB, tstr[pm[0]], tstr[pm[1]], tstr[pm[2]], g.dw, g.accumulate, y_amax, v_amax, NP != 3 ? 1 : 0);
    MSK_LAUNCH_CHECK(ctx);
  }
  return 1;
}
}  // namespace

// the shape / alignment tests of msk_wgrad_wbf (nothing launched): msk_conv.hip's channel-padding wrapper asks before it pads
bool msk_wgrad_wbf_accepts(msk_ctx* ctx, const WGrad& g) {
  (void)ctx;
  const bool k5 = g.kd == 5 && g.kh == 5 && g.kw == 5 && g.pd == 2 && g.ph == 2 && g.pw == 2;
  const bool k3 = g.kd == 3 && g.kh == 3 && g.kw == 3 && g.pd == 1 && g.ph == 1 && g.pw == 1;
  if (!k5 && !k3) return false;
  if (!(g.sd == 1 && g.sh == 1 && g.sw == 1)) return false;
  if (!(g.AD == g.BD && g.AH == g.BH && g.AW == g.BW)) return false;
  if (g.CA < 32 || g.CA % 32 || g.CB < 32 || g.CB % 32) return false;
  if (g.ald % 4 || g.bld % 4 || (((uintptr_t)g.A) & 15) || (((uintptr_t)g.B) & 15)) return false;
  WbfGeom geo;
  int mtd, mth;
  wbf_min_tile(g.CB, &mtd, &mth);
  if (!wbf_pick_geom(g.BD, g.BH, g.BW, mtd, mth, &geo) && !wbf_pick_geom(g.BD, g.BH, g.BW, 8, 8, &geo)) return false;
  return wbf_tile_ok(geo, 8, 16) || wbf_tile_ok(geo, 8, 8);
}

// Returns 1 if handled, 0 if not eligible, < 0 on error.
int msk_wgrad_wbf(msk_ctx* ctx, const WGrad& g) {
  const bool k5 = g.kd == 5 && g.kh == 5 && g.kw == 5 && g.pd == 2 && g.ph == 2 && g.pw == 2;
  const bool k3 = g.kd == 3 && g.kh == 3 && g.kw == 3 && g.pd == 1 && g.ph == 1 && g.pw == 1;
  if (!k5 && !k3) return 0;
  if (!(g.sd == 1 && g.sh == 1 && g.sw == 1)) return 0;
  if (!(g.AD == g.BD && g.AH == g.BH && g.AW == g.BW)) return 0;
  if (g.CA < 32 || g.CA % 32 || g.CB < 32 || g.CB % 32) return 0;
  if (g.ald % 4 || g.bld % 4 || (((uintptr_t)g.A) & 15) || (((uintptr_t)g.B) & 15)) return 0;

  // logical axes and plane dims shared with msk_gconv_wino_bf3; position tiles 8 x TH over (d, h)
  WbfGeom geo;
  int mtd, mth;
  wbf_min_tile(g.CB, &mtd, &mth);
  bool shared_geom = wbf_pick_geom(g.BD, g.BH, g.BW, mtd, mth, &geo);   // the geometry the layer's forward pass used
  if (!shared_geom && !wbf_pick_geom(g.BD, g.BH, g.BW, 8, 8, &geo)) return 0;  // (its forward ran on other kernels)
  int TH = 0;
  if (wbf_tile_ok(geo, 8, 16)) TH = 16;
  else if (wbf_tile_ok(geo, 8, 8)) TH = 8;
  if (!TH) return 0;
  const int np = wbf_pieces(ctx, k5 ? 5 : 3);
  if (k5) return np == 2 ? run_wgrad_pipeline<5, 2>(ctx, g, geo, shared_geom, TH) : run_wgrad_pipeline<5, 3>(ctx, g, geo, shared_geom, TH);
  if (np == 3) return run_wgrad_pipeline<3, 3>(ctx, g, geo, shared_geom, TH);
  if (np == 2) return run_wgrad_pipeline<3, 2>(ctx, g, geo, shared_geom, TH);
  return run_wgrad_pipeline<3, 1>(ctx, g, geo, shared_geom, TH);
}

// Would msk_wgrad_wbf run this problem from two PRE-WRITTEN transforms (V = g.xform in the geometry the forward pass
// used, Y = the dual transform's second output)?  *y_bytes = size of that Y.  Launches nothing.
bool msk_wgrad_wbf_fusable(msk_ctx* ctx, const WGrad& g, size_t* y_bytes) {
  const bool k5 = g.kd == 5 && g.kh == 5 && g.kw == 5 && g.pd == 2 && g.ph == 2 && g.pw == 2;
  const bool k3 = g.kd == 3 && g.kh == 3 && g.kw == 3 && g.pd == 1 && g.ph == 1 && g.pw == 1;
  if (!k5 && !k3) return false;
  if (!(g.sd == 1 && g.sh == 1 && g.sw == 1)) return false;
  if (!(g.AD == g.BD && g.AH == g.BH && g.AW == g.BW)) return false;
  if (g.CA < 32 || g.CA % 32 || g.CB < 32 || g.CB % 32) return false;
  if (g.xform == nullptr) return false;
  WbfGeom geo;
  int mtd, mth;
  wbf_min_tile(g.CB, &mtd, &mth);
  if (!wbf_pick_geom(g.BD, g.BH, g.BW, mtd, mth, &geo)) return false;
  if (!wbf_tile_ok(geo, 8, 16) && !wbf_tile_ok(geo, 8, 8)) return false;
  const int K = k5 ? 5 : 3, NP = wbf_pieces(ctx, K), NPL = 2 * NP;
  const int NXI = K == 5 ? 8 : 6;
  const long ntiles = (long)g.N * geo.T * ((geo.LD + 7) / 8) * ((geo.LH + 7) / 8);
  if (ntiles > 0x7fffffffL) return false;
  const size_t plane = (size_t)geo.DP * geo.HP * 16;
  const size_t v_xi = (size_t)g.N * geo.T * (g.CA / 16) * NPL * plane, y_xi = (size_t)g.N * geo.T * (g.CB / 16) * NPL * plane;
  if (v_xi >= 0xFFFFFFF0ull || y_xi >= 0xFFFFFFF0ull) return false;
  if (msk_wbf_xform_bytes(g.N, g.AD, g.AH, g.AW, g.CA, g.CB, K, NP) != NXI * v_xi + kWbfXformHeader) return false;
  *y_bytes = NXI * y_xi;  // (without the header of msk_wbf_xform_bytes)
  return true;
}

