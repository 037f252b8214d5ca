// 'Same' 5x5x5 convolution (forward and data gradient of every LUConv layer, vnet.py:36) as a THREE-STAGE Winograd
// F(4,5) pipeline whose multiplication stage runs on the bf16 matrix pipe with fp32-exact operands:
//
//   1. wbf_tin_k   V = B^T x along the logical W axis (8 transformed values per 4 inputs), every fp32 V split EXACTLY
//                  into three bf16 pieces  V = hi + mid + lo  (round-to-nearest at each step: 8 + 8 + 8 significand bits
//                  plus the signs cover fp32's 24), written once to HBM in the operand order of the matrix instruction;
//   2. wbf_gemm_k  for each of the 8 Winograd points xi an independent 2-D (kd, kh) convolution
//                      M_xi[n,d,h,t][co] = sum_{kd,kh,ci} V_xi[n,d+kd-2,h+kh-2,t][ci] * U_xi[kd,kh][ci][co]
//                  as an implicit GEMM on v_mfma_f32_32x32x16_bf16 with SIX products per fp32 product
//                      hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi      (fp32 accumulate)
//                  -- the dropped terms (mid*lo, lo*mid, lo*lo) are <= 2^-23 of |V||U|, the size of one fp32 rounding;
//                  the transformed weights U = G w are computed in double and split the same way.  The bf16 pipe runs
//                  16x the fp32 matrix rate, so the six-product emulation is 2.67x the fp32 MFMA peak at fp32-class
//                  error (tools/winograd_numerics.py: below the error of the fp32 F(4,5) kernels it replaces, because
//                  their error is dominated by fp32 accumulation of the large transformed products);
//   3. wbf_tout_k  y = A^T M (4 outputs per 8 points) + bias [+ dst] [PReLU], sums split-K slabs in a fixed order.
//
// Nothing is transformed inside the MFMA loop (round-1 verdict: the per-(kd,kh) register transform of
// conv_halo_wino4_k held the matrix pipe at 0.51-0.74): the GEMM stage is LDS/L2 -> MFMA only.  HBM pays for it: V is
// 3x the input bytes and M 2x the output bytes -- the step used < 10 % of the HBM roof before.
//
// Layouts (16-byte slots of 8 bf16 = one MFMA operand fragment per lane):
//   V [xi][n][t][kc][piece][khalf][DP][HP]   slot (dp, hp) = position (d = dp-2, h = hp-2), zero outside the volume
//                                            (DP/HP = tile-padded dims + 4): the GEMM stages its halo tile with plain
//                                            address arithmetic, no bounds checks; channel = kc*16 + khalf*8 + j
//   U [xi][tap][kc][piece][khalf][CN]        B fragment of lane (khalf, co) is one contiguous slot
//   M [ks][xi][n][t][d][h][CN] fp32
#include "msk_wbf.h"

namespace {

__device__ __forceinline__ void split3_one(float x, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
  hi = __builtin_bit_cast(unsigned short, (__bf16)x);
  const float r = x - __uint_as_float((unsigned)hi << 16);
  mid = __builtin_bit_cast(unsigned short, (__bf16)r);
  const float r2 = r - __uint_as_float((unsigned)mid << 16);
  lo = __builtin_bit_cast(unsigned short, (__bf16)r2);
}

// ---------------------------------------------------------------------------------------------------------
// weights: U_xi[tap = kd*5+kh][k][n] = sum_kw G[xi][kw] w(kd, kh, kw; k, n), in double, split into 3 bf16
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
wbf_pack_weights_k(const float* __restrict__ w, int A, int B, int swap, int flip, int CK, int CN, int KC, int tsd, int tsh,
                   int tsw, unsigned short* __restrict__ out, long xi_stride /*elements*/) {
  const double G[8][5] = {{-1, 0, 0, 0, 0},
                          {-2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9},
                          {-2.0 / 9, 2.0 / 9, -2.0 / 9, 2.0 / 9, -2.0 / 9},
                          {1.0 / 90, 1.0 / 45, 2.0 / 45, 4.0 / 45, 8.0 / 45},
                          {1.0 / 90, -1.0 / 45, 2.0 / 45, -4.0 / 45, 8.0 / 45},
                          {32.0 / 45, 16.0 / 45, 8.0 / 45, 4.0 / 45, 2.0 / 45},
                          {32.0 / 45, -16.0 / 45, 8.0 / 45, -4.0 / 45, 2.0 / 45},
                          {0, 0, 0, 0, 1}};
  // one thread per (tap row, k, n): reads its 5 kw taps once, writes 8 xi x 3 pieces
  const long total = 25L * KC * 16 * CN;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int j = (int)(idx & 7);  // channel within the slot (fastest: 16-byte slots are written by 8 neighbours)
    long r_ = idx >> 3;
    const int n = (int)(r_ % CN);
    r_ /= CN;
    const int khalf = (int)(r_ & 1);
    r_ >>= 1;
    const int kc = (int)(r_ % KC);
    const int row = (int)(r_ / KC);
    const int k = kc * 16 + khalf * 8 + j;
    double t[5] = {0, 0, 0, 0, 0};
    if (k < CK) {
      const int ia = swap ? n : k, ib = swap ? k : n;
      const float* wp = w + ((long)ia * B + ib) * 125;
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) {
        const int tap = (row / 5) * tsd + (row % 5) * tsh + kw * tsw;  // logical (kd, kh, kw) -> canonical tap
        t[kw] = (double)wp[flip ? 124 - tap : tap];
      }
    }
    // element index = (((((xi*25 + row)*KC + kc)*3 + piece)*2 + khalf)*CN + n)*8 + j
    const long base = ((((long)row * KC + kc) * 3 * 2 + khalf) * CN + n) * 8 + j;
    const long pstep = 2L * CN * 8;
#pragma unroll
    for (int xi = 0; xi < 8; ++xi) {
      double s_ = 0.0;
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) s_ += G[xi][kw] * t[kw];
      unsigned short hi, mid, lo;
      split3_one((float)s_, hi, mid, lo);
      unsigned short* o = out + (long)xi * xi_stride + base;
      o[0] = hi;
      o[pstep] = mid;
      o[2 * pstep] = lo;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// stage 1: input transform + split
// ---------------------------------------------------------------------------------------------------------

__device__ __forceinline__ void bt8(const float d0, const float d1, const float d2, const float d3, const float d4,
                                    const float d5, const float d6, const float d7, float (&v)[8]) {
  v[0] = (d6 - d0) + 5.25f * (d2 - d4);
  v[7] = (d7 - d1) + 5.25f * (d3 - d5);
  const float t1 = (d2 + d6) - 4.25f * d4, t2 = (d1 + d5) - 4.25f * d3;
  v[1] = t1 + t2;
  v[2] = t1 - t2;
  const float m3 = 2.5f * d3;
  const float t3 = (d6 + 0.25f * d2) - 1.25f * d4, t4 = (0.5f * d1 - m3) + 2.f * d5;
  v[3] = t3 + t4;
  v[4] = t3 - t4;
  const float t5 = (d6 + 4.f * d2) - 5.f * d4, t6 = (2.f * d1 - m3) + 0.5f * d5;
  v[5] = t5 + t6;
  v[6] = t5 - t6;
}

// adjoint of the output transform: Y_xi = sum_j AT[j][xi] dy_j  (AT as in wbf_tout_k)
__device__ __forceinline__ void at8(const float e0, const float e1, const float e2, const float e3, float (&v)[8]) {
  v[0] = e0;
  v[7] = e3;
  const float s02 = e0 + e2, s13 = e1 + e3;
  v[1] = s02 + s13;
  v[2] = s02 - s13;
  const float p = e0 + 4.f * e2, q = 2.f * e1 + 8.f * e3;
  v[3] = p + q;
  v[4] = p - q;
  const float p2 = e0 + 0.25f * e2, q2 = 0.5f * e1 + 0.125f * e3;
  v[5] = p2 + q2;
  v[6] = p2 - q2;
}

// thread = (n, padded position (dp, hp), 8-channel group); walks t = 0 .. T-1 (MODE 0: with a sliding 8-wide W window, x
// is read once).  Lanes: 4 channel groups fastest (one 128-byte line of x per 4 lanes), then 64 consecutive positions
// (each (xi, piece) store of a wavefront covers 4 runs of 16 consecutive slots).
template <int MODE>
__global__ void __launch_bounds__(256)
wbf_tin_k(WbfTinArgs a) {
  // lane mapping (A/B, option "wbf_tin_map"): 0 = 4 channel groups fastest (reads: full 128-byte lines per 4 lanes;
  // stores: 4 runs of 256 B per wavefront), 1 = one channel group per wavefront (stores: one 1 KiB run; reads: 32 of
  // every 128 bytes per lane, the rest of the line goes to the block's other wavefronts through L1/L2)
  const int cgl = a.lane_map ? (threadIdx.x >> 6) : (threadIdx.x & 3), pl = a.lane_map ? (threadIdx.x & 63) : (threadIdx.x >> 2);
  const int ncgb = a.CK >> 5;
  const int cgb = blockIdx.x % ncgb, pb = blockIdx.x / ncgb;
  const int pos = pb * 64 + pl;
  const int n = blockIdx.y;
  if (pos >= a.DP * a.HP) return;
  const int dp = pos / a.HP, hp = pos - dp * a.HP;
  const int cg = cgb * 4 + cgl, kc = cg >> 1, khalf = cg & 1;
  const int d = dp - 2, h = hp - 2;
  const bool live = d >= 0 && d < a.LD && h >= 0 && h < a.LH;
  const long plane = (long)a.DP * a.HP * 16;
  char* vb = a.V + (((long)n * a.T * a.KC + kc) * 6 + khalf) * plane + (long)pos * 16;
  const long tstep = (long)a.KC * 6 * plane;
  const float* xb = a.src + ((long)n * a.svn + (long)d * a.svd + (long)h * a.svh) * a.sld + cg * 8;
  const long wstep = (long)a.svw * a.sld;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

  float4 win[8][2];
  if (MODE == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int w = j - 2;
      if (live && w >= 0 && w < a.LW) {
        const float4* p = reinterpret_cast<const float4*>(xb + w * wstep);
        win[j][0] = p[0];
        win[j][1] = p[1];
      } else {
        win[j][0] = win[j][1] = z4;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (live && j < a.LW) {
        const float4* p = reinterpret_cast<const float4*>(xb + j * wstep);
        win[j][0] = p[0];
        win[j][1] = p[1];
      } else {
        win[j][0] = win[j][1] = z4;
      }
    }
  }
  for (int t = 0; t < a.T; ++t) {
    float4 nxt[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int w = MODE == 0 ? 4 * t + 6 + j : 4 * t + 4 + j;  // the part of tile t + 1 not yet in registers
      if (live && t + 1 < a.T && w < a.LW) {
        const float4* p = reinterpret_cast<const float4*>(xb + w * wstep);
        nxt[j][0] = p[0];
        nxt[j][1] = p[1];
      } else {
        nxt[j][0] = nxt[j][1] = z4;
      }
    }
    float v[8][8];  // [channel][xi]
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (MODE == 0) {
        bt8(win[0][q].x, win[1][q].x, win[2][q].x, win[3][q].x, win[4][q].x, win[5][q].x, win[6][q].x, win[7][q].x, v[q * 4 + 0]);
        bt8(win[0][q].y, win[1][q].y, win[2][q].y, win[3][q].y, win[4][q].y, win[5][q].y, win[6][q].y, win[7][q].y, v[q * 4 + 1]);
        bt8(win[0][q].z, win[1][q].z, win[2][q].z, win[3][q].z, win[4][q].z, win[5][q].z, win[6][q].z, win[7][q].z, v[q * 4 + 2]);
        bt8(win[0][q].w, win[1][q].w, win[2][q].w, win[3][q].w, win[4][q].w, win[5][q].w, win[6][q].w, win[7][q].w, v[q * 4 + 3]);
      } else {
        at8(win[0][q].x, win[1][q].x, win[2][q].x, win[3][q].x, v[q * 4 + 0]);
        at8(win[0][q].y, win[1][q].y, win[2][q].y, win[3][q].y, v[q * 4 + 1]);
        at8(win[0][q].z, win[1][q].z, win[2][q].z, win[3][q].z, v[q * 4 + 2]);
        at8(win[0][q].w, win[1][q].w, win[2][q].w, win[3][q].w, v[q * 4 + 3]);
      }
    }
    char* vt = vb + t * tstep;
#pragma unroll
    for (int xi = 0; xi < 8; ++xi) {
      uint4 hi, mid, lo;
      wbf_split3_pair(v[0][xi], v[1][xi], hi.x, mid.x, lo.x);
      wbf_split3_pair(v[2][xi], v[3][xi], hi.y, mid.y, lo.y);
      wbf_split3_pair(v[4][xi], v[5][xi], hi.z, mid.z, lo.z);
      wbf_split3_pair(v[6][xi], v[7][xi], hi.w, mid.w, lo.w);
      char* o = vt + (long)xi * a.v_xi;
      *reinterpret_cast<uint4*>(o) = hi;
      *reinterpret_cast<uint4*>(o + 2 * plane) = mid;
      *reinterpret_cast<uint4*>(o + 4 * plane) = lo;
    }
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        win[j][0] = win[j + 4][0];
        win[j][1] = win[j + 4][1];
        win[j + 4][0] = nxt[j][0];
        win[j + 4][1] = nxt[j][1];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        win[j][0] = nxt[j][0];
        win[j][1] = nxt[j][1];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// stage 2: per-xi 2-D convolution as an implicit GEMM on the bf16 matrix pipe
// ---------------------------------------------------------------------------------------------------------
struct GemmArgs {
  const char* V;
  const char* U;
  float* M;
  int N, T, KC, CN, LD, LH, DP, HP;
  int tiles_d, tiles_h, ngrp, ksplit, kc_per;
  long v_xi, v_plane;  // bytes
  long u_xi;           // bytes
  long m_xi;           // floats between xi planes of M (= N*T*LD*LH*CN); split slabs are 8*m_xi apart
};

__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned uniform_bytes) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_bytes, (int)uniform_bytes, 0));
}

#define WBF_MFMA(acc, av, bv) \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc, 0, 0, 0)

// Workgroup = 4 wavefronts as WM (rows) x WN (column groups of 32); a wavefront owns MR row fragments of 32 positions
// and ONE 32-channel column fragment: its B fragments come straight from L2/L1 (3 slots per tap and 16-channel chunk,
// used by MR*6 MFMAs), A fragments from the LDS halo tile (TD+4) x (TH+4) that the whole workgroup shares and every
// one of the 25 taps re-reads.  The tile is filled by LDS-DMA (buffer_load ... lds, 16 B per lane), no registers.
template <int MR, int WM, int WN, int TD, int TH>
__global__ void __launch_bounds__(256)
wbf_gemm_k(GemmArgs a) {
  static_assert(WM * WN == 4 && WM * MR * 32 == TD * TH, "tile shape");
  constexpr int HDt = TD + 4, HPt = TH + 4, NSLOT = HDt * HPt, NIT = 6 * NSLOT, ROUNDS = (NIT + 255) / 256;
  __shared__ uint4 lds[ROUNDS * 256];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  int b = blockIdx.x;
  const int xi = b & 7;  // block b runs on XCD b % 8: one Winograd point per XCD, its weights stay in that L2
  b >>= 3;
  const int grp = b % a.ngrp;
  b /= a.ngrp;
  const int thi = b % a.tiles_h;
  b /= a.tiles_h;
  const int tdi = b % a.tiles_d;
  b /= a.tiles_d;
  const int t = b % a.T;
  b /= a.T;
  const int n = b % a.N;
  const int ks = b / a.N;

  // LDS-DMA: item it = r*256 + tid -> (plane pk = piece*2 + khalf, slot); slot = row*HPt + col of the halo tile
  unsigned voff[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    int it = r * 256 + tid;
    if (it >= NIT) it = 0;  // the last round overshoots: re-read item 0 into the unused tail of the image
    const int pk = it / NSLOT, slot = it - pk * NSLOT;
    const int row = slot / HPt, col = slot - row * HPt;
    voff[r] = (unsigned)(pk * a.v_plane + ((long)row * a.HP + col) * 16);
  }
  const char* vtile = a.V + (long)xi * a.v_xi + ((long)(n * a.T + t) * a.KC) * 6 * a.v_plane +
                      ((long)(tdi * TD) * a.HP + thi * TH) * 16;
  const __amdgpu_buffer_rsrc_t vres = __builtin_amdgcn_make_buffer_rsrc((void*)vtile, 0, 0xFFFFFFF0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t ures =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.U + (long)xi * a.u_xi), 0, 0xFFFFFFF0u, 0x00020000);
  const unsigned ulane = (unsigned)(lh * a.CN + (grp * WN + wn) * 32 + li) * 16u;
  const unsigned ustep = (unsigned)a.CN * 32u;          // bytes between pieces (2 khalf planes of CN slots)
  const unsigned uchunk = 3u * ustep;                   // bytes between 16-channel chunks
  const unsigned utap = (unsigned)a.KC * uchunk;        // bytes between taps

  // A rows of this lane: fragment f = wm*MR + mr covers tile rows [32 f, 32 f + 32), row -> (dd, hh) = (r / TH, r % TH)
  int arow[MR];
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
    const int r = (wm * MR + mr) * 32 + li;
    arow[mr] = lh * NSLOT + (r / TH) * HPt + (r % TH);
  }

  f32x16 acc[MR];
#pragma unroll
  for (int mr = 0; mr < MR; ++mr)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[mr][j] = 0.f;

  const int kc0 = ks * a.kc_per;
  const int kc1 = min(a.KC, kc0 + a.kc_per);
  for (int kc = kc0; kc < kc1; ++kc) {
    __syncthreads();  // every wavefront is done reading the previous chunk's tile
    const unsigned vsoff = (unsigned)(kc * 6 * a.v_plane);
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(vres, (__attribute__((address_space(3))) void*)(lds + r * 256 + wave * 64), 16,
                                               (int)voff[r], (int)vsoff, 0, 0);
    const unsigned ukc = (unsigned)kc * uchunk;
    uint4 bq[2][3];
    bq[0][0] = buf_load16(ures, ulane, ukc);
    bq[0][1] = buf_load16(ures, ulane, ukc + ustep);
    bq[0][2] = buf_load16(ures, ulane, ukc + 2 * ustep);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // 25 taps, fully unrolled, software pipelined by hand: the A fragments (LDS) and B fragments (L2) of tap + 1 are
    // requested before the MFMAs of tap, and the scheduler may not move them (sched_barrier) -- left alone it sinks every
    // load to its first use to save registers, and each tap then waits out an LDS and an L2 round trip
    // (PMC: matrix pipe 66 % busy).  Consecutive MFMAs alternate between the accumulators.
    uint4 aq[2][MR][3];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
      const uint4* ap = lds + arow[mr];
      aq[0][mr][0] = ap[0];
      aq[0][mr][1] = ap[2 * NSLOT];
      aq[0][mr][2] = ap[4 * NSLOT];
    }
#pragma unroll
    for (int tap = 0; tap < 25; ++tap) {
      const int cur = tap & 1, nx = cur ^ 1;
      if (tap + 1 < 25) {
        const unsigned ub = (unsigned)(tap + 1) * utap + ukc;
        bq[nx][0] = buf_load16(ures, ulane, ub);
        bq[nx][1] = buf_load16(ures, ulane, ub + ustep);
        bq[nx][2] = buf_load16(ures, ulane, ub + 2 * ustep);
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
          const uint4* ap = lds + arow[mr] + ((tap + 1) / 5) * HPt + ((tap + 1) % 5);
          aq[nx][mr][0] = ap[0];
          aq[nx][mr][1] = ap[2 * NSLOT];
          aq[nx][mr][2] = ap[4 * NSLOT];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // small terms first: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
#pragma unroll
      for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][2], bq[cur][0]);
#pragma unroll
      for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][0], bq[cur][2]);
#pragma unroll
      for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][1], bq[cur][1]);
#pragma unroll
      for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][1], bq[cur][0]);
#pragma unroll
      for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][0], bq[cur][1]);
#pragma unroll
      for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][0], bq[cur][0]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // store M[ks][xi][n][t][d][h][co]
  const int co = (grp * WN + wn) * 32 + li;
  float* mbase = a.M + ((long)ks * 8 + xi) * a.m_xi + ((long)(n * a.T + t) * a.LD) * a.LH * a.CN + co;
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int r = (wm * MR + mr) * 32 + (j & 3) + 8 * (j >> 2) + 4 * lh;
      const int d = tdi * TD + r / TH, h = thi * TH + r % TH;
      if (d < a.LD && h < a.LH) mbase[((long)d * a.LH + h) * a.CN] = acc[mr][j];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// stage 3: output transform
// ---------------------------------------------------------------------------------------------------------
struct ToutArgs {
  const float* M;
  long m_xi;
  int ksplit;
  int N, T, LD, LH, LW, CN;
  float* dst;
  int dld;
  long dvn;
  int dvd, dvh, dvw;
  const float* bias;
  const float* prelu;
  int accumulate;
  float* stat_partial;  // STATS: per-block BatchNorm records [gridDim.x][CN][3] = (n, mean, M2) of the stored values
};

struct WfRec {
  float n, mean, m2;
};
__device__ __forceinline__ WfRec wfrec_merge(WfRec a, WfRec b) {
  if (b.n == 0.f) return a;
  if (a.n == 0.f) return b;
  WfRec r;
  r.n = a.n + b.n;
  const float d = b.mean - a.mean, f = b.n / r.n;
  r.mean = a.mean + d * f;
  r.m2 = a.m2 + b.m2 + d * d * a.n * f;
  return r;
}

// STATS: the BatchNorm statistics of the convolution output (vnet.py:38,41 -- conv followed by BatchNorm) are taken
// here, from the values on their way to HBM: shifted sums per thread (a thread keeps its channel quad over the grid-stride
// loop because gridDim.x * 256 is a multiple of CN / 4), Chan merge inside the block, one record per block and channel;
// msk_bn_stats_merge finishes in double.  Saves the separate read of y by bn_stats_partial.
template <bool STATS>
__global__ void __launch_bounds__(256)
wbf_tout_k(ToutArgs a) {
  const int c4n = a.CN >> 2;
  float sk[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  float cnt = 0.f;
  const long total = (long)a.N * a.T * a.LD * a.LH * c4n;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % c4n);
    long r_ = idx / c4n;
    const int h = (int)(r_ % a.LH);
    r_ /= a.LH;
    const int d = (int)(r_ % a.LD);
    r_ /= a.LD;
    const int t = (int)(r_ % a.T);
    const int n = (int)(r_ / a.T);
    float4 m[8];
#pragma unroll
    for (int xi = 0; xi < 8; ++xi) {
      const float4* p = reinterpret_cast<const float4*>(a.M + (long)xi * a.m_xi) + idx;
      float4 s = p[0];
      for (int z = 1; z < a.ksplit; ++z) {  // fixed order
        const float4 q = p[(long)z * 2 * a.m_xi];  // 8 * m_xi floats = 2 * m_xi float4
        s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
      }
      m[xi] = s;
    }
    float4 y[4];
#define WBF_AT(c)                                                                 \
    {                                                                             \
      const float s12 = m[1].c + m[2].c, d12 = m[1].c - m[2].c;                   \
      const float s34 = m[3].c + m[4].c, d34 = m[3].c - m[4].c;                   \
      const float s56 = m[5].c + m[6].c, d56 = m[5].c - m[6].c;                   \
      y[0].c = ((m[0].c + s12) + s34) + s56;                                      \
      y[1].c = (d12 + 2.f * d34) + 0.5f * d56;                                    \
      y[2].c = (s12 + 4.f * s34) + 0.25f * s56;                                   \
      y[3].c = ((d12 + 8.f * d34) + 0.125f * d56) + m[7].c;                       \
    }
    WBF_AT(x) WBF_AT(y) WBF_AT(z) WBF_AT(w)
#undef WBF_AT
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), sl = make_float4(1.f, 1.f, 1.f, 1.f);
    if (a.bias) bv = reinterpret_cast<const float4*>(a.bias)[c4];
    if (a.prelu) sl = reinterpret_cast<const float4*>(a.prelu)[c4];
    float* o = a.dst + ((long)n * a.dvn + (long)d * a.dvd + (long)h * a.dvh + (long)(4 * t) * a.dvw) * a.dld + c4 * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (4 * t + i < a.LW) {
        float4* op = reinterpret_cast<float4*>(o + (long)i * a.dvw * a.dld);
        float4 r = make_float4(y[i].x + bv.x, y[i].y + bv.y, y[i].z + bv.z, y[i].w + bv.w);
        if (a.accumulate) {
          const float4 e = *op;
          r.x += e.x; r.y += e.y; r.z += e.z; r.w += e.w;
        }
        r.x = r.x > 0.f ? r.x : sl.x * r.x;
        r.y = r.y > 0.f ? r.y : sl.y * r.y;
        r.z = r.z > 0.f ? r.z : sl.z * r.z;
        r.w = r.w > 0.f ? r.w : sl.w * r.w;
        *op = r;
        if (STATS) {
          const float rv[4] = {r.x, r.y, r.z, r.w};
          if (cnt == 0.f) {
#pragma unroll
            for (int j = 0; j < 4; ++j) sk[j] = rv[j];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float dlt = rv[j] - sk[j];
            s1[j] += dlt;
            s2[j] = fmaf(dlt, dlt, s2[j]);
          }
          cnt += 1.f;
        }
      }
    }
  }
  if (STATS) {
    __shared__ WfRec sh[4][256];
    const int t = threadIdx.x, vl = t / c4n, VL = 256 / c4n;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      WfRec w = {0.f, 0.f, 0.f};
      if (cnt > 0.f) {
        w.n = cnt;
        w.mean = sk[j] + s1[j] / cnt;
        w.m2 = fmaxf(s2[j] - s1[j] * s1[j] / cnt, 0.f);
      }
      sh[j][t] = w;
    }
    __syncthreads();
    for (int s_ = VL >> 1; s_ > 0; s_ >>= 1) {
      if (vl < s_) {
#pragma unroll
        for (int j = 0; j < 4; ++j) sh[j][t] = wfrec_merge(sh[j][t], sh[j][t + s_ * c4n]);
      }
      __syncthreads();
    }
    if (vl == 0) {
      const int c = (t % c4n) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float* p_ = a.stat_partial + ((long)blockIdx.x * a.CN + c + j) * 3;
        p_[0] = sh[j][t].n;
        p_[1] = sh[j][t].mean;
        p_[2] = sh[j][t].m2;
      }
    }
  }
}

template <int MR, int WM, int WN, int TD, int TH>
void launch_gemm(msk_ctx* ctx, const GemmArgs& a, long nblk) {
  hipLaunchKernelGGL((wbf_gemm_k<MR, WM, WN, TD, TH>), dim3((unsigned)nblk), dim3(256), 0, ctx->stream, a);
}

}  // namespace

int msk_wbf_transform(msk_ctx* ctx, int mode, const WbfTinArgs& ta_in) {
  WbfTinArgs ta = ta_in;
  ta.lane_map = ctx->wbf_tin_map;
  const int pblocks = (ta.DP * ta.HP + 63) / 64;
  msk_launch_scope ls(ctx, mode == 0 ? "wbf_tin_k" : "wbf_ty_k");
  if (mode == 0)
    hipLaunchKernelGGL(wbf_tin_k<0>, dim3((unsigned)(pblocks * (ta.CK / 32)), ta.N), dim3(256), 0, ctx->stream, ta);
  else
    hipLaunchKernelGGL(wbf_tin_k<1>, dim3((unsigned)(pblocks * (ta.CK / 32)), ta.N), dim3(256), 0, ctx->stream, ta);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

namespace {
// tile variants {id, MR, WM, WN, TD, TH} by output channels (first = preferred)
struct Var { int id, MR, WM, WN, TD, TH; };
// Measured (tools/bench_conv.py, 2 x 128^3 .. 2 x 16^3): the MR = 2 variants win everywhere (32ch@128^3 3.06 vs 3.13 ms,
// 128ch@32^3 0.60 vs 0.85 ms, 256ch@16^3 0.32 vs 0.45 ms): smaller LDS tiles -> 3-4 workgroups per CU hide the
// staging barriers; the MR = 4 variants halve the B-fragment traffic and stay selectable ("wbf_variant").
const Var kVars[6] = {{4, 2, 4, 1, 16, 16}, {0, 4, 4, 1, 16, 32},    // CN == 32
                      {5, 2, 2, 2, 8, 16},  {1, 4, 2, 2, 16, 16},    // CN == 64
                      {3, 2, 1, 4, 8, 8},   {2, 4, 1, 4, 8, 16}};    // CN % 128 == 0
const Var* pick_variant(const msk_ctx* ctx, const WbfGeom& geo, int CN) {
  int v0;
  if (CN == 32) v0 = 0;
  else if (CN == 64) v0 = 2;
  else if (CN >= 128 && CN % 128 == 0) v0 = 4;
  else return nullptr;
  for (int c = v0; c < v0 + 2; ++c) {
    const Var& v = kVars[c];
    if (ctx->wbf_variant >= 0 && ctx->wbf_variant != v.id) continue;  // tuning knob "wbf_variant"
    if (wbf_tile_ok(geo, v.TD, v.TH)) return &v;
  }
  return nullptr;
}
}  // namespace

// Bytes of the transformed input V = split(B^T x) of a 'same' 5^3 convolution over a [n, d, h, w, c] tensor in the shared
// geometry, or 0 when the tensor is not eligible.
size_t msk_wbf_xform_bytes(int n, int d, int h, int w, int c, int cout) {
  WbfGeom geo;
  int mtd, mth;
  wbf_min_tile(cout, &mtd, &mth);
  if (c < 32 || c % 32 || !wbf_pick_geom(d, h, w, mtd, mth, &geo)) return 0;
  const size_t v_xi = (size_t)n * geo.T * (c / 16) * 6 * geo.DP * geo.HP * 16;
  if (v_xi >= 0xFFFFFFF0ull) return 0;
  return 8 * v_xi;
}
// the same, 0 unless msk_gconv_wino_bf3 will run the forward convolution c -> cout of that tensor
size_t msk_wbf_fwd_xform_bytes(const msk_ctx* ctx, int n, int d, int h, int w, int c, int cout) {
  WbfGeom geo;
  int mtd, mth;
  wbf_min_tile(cout, &mtd, &mth);
  if (!wbf_pick_geom(d, h, w, mtd, mth, &geo) || !pick_variant(ctx, geo, cout)) return 0;
  return msk_wbf_xform_bytes(n, d, h, w, c, cout);
}

// Returns 1 if handled, 0 if the problem is not eligible, < 0 on error.
int msk_gconv_wino_bf3(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap) {
  if (!(g.kd == 5 && g.kh == 5 && g.kw == 5 && g.pd == 2 && g.ph == 2 && g.pw == 2)) return 0;
  if (!(g.sd == 1 && g.sh == 1 && g.sw == 1)) return 0;
  if (!(g.SD == g.DD && g.SH == g.DH && g.SW == g.DW)) return 0;
  if (g.CK < 32 || g.CK % 32 || g.CN < 32 || g.CN % 32) return 0;
  if (g.sld % 4 || g.dld % 4 || (((uintptr_t)g.src) & 15) || (((uintptr_t)g.dst) & 15)) return 0;
  if (g.bias && (((uintptr_t)g.bias) & 15)) return 0;
  if (g.prelu && (((uintptr_t)g.prelu) & 15)) return 0;

  // logical axes and plane dims (shared with the weight gradient), then the first tile variant of the class that fits
  WbfGeom geo;
  int mtd, mth;
  wbf_min_tile(g.CN, &mtd, &mth);
  if (!wbf_pick_geom(g.DD, g.DH, g.DW, mtd, mth, &geo)) return 0;
  const Var* bv = pick_variant(ctx, geo, g.CN);
  if (!bv) return 0;
  const int TD = bv->TD, TH = bv->TH, variant = bv->id;
  const int* pm = geo.perm;
  const int LD = geo.LD, LH = geo.LH, LW = geo.LW;
  const int vstr[3] = {g.DH * g.DW, g.DW, 1};
  const int tstr[3] = {25, 5, 1};
  const int T = geo.T, KC = g.CK / 16;
  const int tiles_d = (LD + TD - 1) / TD, tiles_h = (LH + TH - 1) / TH;
  const int DP = geo.DP, HP = geo.HP;
  const int WN = bv->WN;
  const int ngrp = g.CN / (WN * 32);

  // split K (16-channel chunks) when the tiling alone cannot fill the chip
  const long base_blocks = 8L * ngrp * tiles_h * tiles_d * T * g.N;
  int ksplit = 1, kc_per = KC;
  if (base_blocks < 3L * ctx->num_cu) {
    long want = (4L * ctx->num_cu + base_blocks - 1) / base_blocks;
    if (want > KC) want = KC;
    kc_per = (int)((KC + want - 1) / want);
    ksplit = (KC + kc_per - 1) / kc_per;
  }
  const long nblk = base_blocks * ksplit;
  if (nblk > 0x7fffffffL) return 0;

  const size_t v_plane = (size_t)DP * HP * 16;
  const size_t v_xi = (size_t)g.N * T * KC * 6 * v_plane;
  const size_t m_xi = (size_t)g.N * T * LD * LH * g.CN;  // floats
  if (v_xi >= 0xFFFFFFF0ull) return 0;                     // 32-bit offsets inside one xi plane
  const size_t u_xi = (size_t)25 * KC * 3 * 2 * g.CN * 16;
  if (u_xi >= 0xFFFFFFF0ull) return 0;
  const size_t v_bytes = (8 * v_xi + 255) & ~(size_t)255, m_bytes = ((size_t)ksplit * 8 * m_xi * sizeof(float) + 255) & ~(size_t)255;
  long tout_blocks = ((long)m_xi / 4 + 255) / 256;
  if (tout_blocks > 16L * ctx->num_cu) tout_blocks = 16L * ctx->num_cu;
  const int c4n = g.CN / 4;
  const bool fuse_stats = g.stats != nullptr && !g.accumulate && !g.prelu && c4n <= 256 && (c4n & (c4n - 1)) == 0;
  const size_t s_bytes = fuse_stats ? (size_t)tout_blocks * g.CN * 3 * sizeof(float) : 0;
  char* wsp = (char*)msk_workspace(ctx, (g.xform ? 0 : v_bytes) + m_bytes + s_bytes + 256);
  if (!wsp) return -1;
  char* V = g.xform ? (char*)g.xform : wsp;
  float* M = (float*)(g.xform ? wsp : wsp + v_bytes);
  float* SP = (float*)((char*)M + m_bytes);
  char* U = (char*)msk_workspace2(ctx, 8 * u_xi);
  if (!U) return -1;

  {
    msk_launch_scope ls(ctx, "wbf_pack_weights");
    long blocks = (25L * KC * 16 * g.CN + 255) / 256;
    if (blocks > 16L * ctx->num_cu) blocks = 16L * ctx->num_cu;
    hipLaunchKernelGGL(wbf_pack_weights_k, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, w_canon, A, B, swap,
                       g.transposed ? 1 : 0, g.CK, g.CN, KC, tstr[pm[0]], tstr[pm[1]], tstr[pm[2]], (unsigned short*)U,
                       (long)(u_xi / 2));
    MSK_LAUNCH_CHECK(ctx);
  }
  {
    WbfTinArgs ta{};
    ta.src = g.src; ta.sld = g.sld;
    ta.svn = (long)g.DD * g.DH * g.DW; ta.svd = vstr[pm[0]]; ta.svh = vstr[pm[1]]; ta.svw = vstr[pm[2]];
    ta.N = g.N; ta.LD = LD; ta.LH = LH; ta.LW = LW; ta.T = T; ta.CK = g.CK; ta.KC = KC;
    ta.DP = DP; ta.HP = HP; ta.V = V; ta.v_xi = (long)v_xi;
    if (msk_wbf_transform(ctx, 0, ta) != 0) return -1;
  }
  {
    GemmArgs ga{};
    ga.V = V; ga.U = U; ga.M = M;
    ga.N = g.N; ga.T = T; ga.KC = KC; ga.CN = g.CN; ga.LD = LD; ga.LH = LH; ga.DP = DP; ga.HP = HP;
    ga.tiles_d = tiles_d; ga.tiles_h = tiles_h; ga.ngrp = ngrp; ga.ksplit = ksplit; ga.kc_per = kc_per;
    ga.v_xi = (long)v_xi; ga.v_plane = (long)v_plane; ga.u_xi = (long)u_xi; ga.m_xi = (long)m_xi;
    const char* tag = "wbf_gemm_k";
    if (ctx->prof && ctx->prof_shapes) {
      char buf[200];
      snprintf(buf, sizeof(buf), "wbf_gemm_k[ck=%d,cn=%d,n=%d,dhw=%dx%dx%d,ks=%d]", g.CK, g.CN, g.N, g.DD, g.DH, g.DW, ksplit);
      tag = msk_intern_tag(ctx, buf);
    }
    msk_launch_scope ls(ctx, tag);
    switch (variant) {
      case 0: launch_gemm<4, 4, 1, 16, 32>(ctx, ga, nblk); break;
      case 1: launch_gemm<4, 2, 2, 16, 16>(ctx, ga, nblk); break;
      case 2: launch_gemm<4, 1, 4, 8, 16>(ctx, ga, nblk); break;
      case 3: launch_gemm<2, 1, 4, 8, 8>(ctx, ga, nblk); break;
      case 4: launch_gemm<2, 4, 1, 16, 16>(ctx, ga, nblk); break;
      default: launch_gemm<2, 2, 2, 8, 16>(ctx, ga, nblk); break;
    }
    MSK_LAUNCH_CHECK(ctx);
  }
  {
    ToutArgs oa{};
    oa.M = M; oa.m_xi = (long)m_xi; oa.ksplit = ksplit;
    oa.N = g.N; oa.T = T; oa.LD = LD; oa.LH = LH; oa.LW = LW; oa.CN = g.CN;
    oa.dst = g.dst; oa.dld = g.dld;
    oa.dvn = (long)g.DD * g.DH * g.DW; oa.dvd = vstr[pm[0]]; oa.dvh = vstr[pm[1]]; oa.dvw = vstr[pm[2]];
    oa.bias = g.bias; oa.prelu = g.prelu; oa.accumulate = g.accumulate;
    oa.stat_partial = SP;
    {
      msk_launch_scope ls(ctx, "wbf_tout_k");
      if (fuse_stats) hipLaunchKernelGGL(wbf_tout_k<true>, dim3((unsigned)tout_blocks), dim3(256), 0, ctx->stream, oa);
      else hipLaunchKernelGGL(wbf_tout_k<false>, dim3((unsigned)tout_blocks), dim3(256), 0, ctx->stream, oa);
      MSK_LAUNCH_CHECK(ctx);
    }
    if (fuse_stats) {
      if (msk_bn_stats_merge(ctx, SP, (int)tout_blocks, g.CN, g.stats) != 0) return -1;
      ctx->stats_fused = true;
    }
  }
  if (g.xform) ctx->xform_written = true;
  return 1;
}
