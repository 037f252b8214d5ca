// 'Same' K x K x K convolution, K = 5 (every LUConv layer, vnet.py:36: forward and data gradient) or K = 3 (UNet3D's
// DoubleConvs, the deep-supervision heads), as a THREE-STAGE 1-D Winograd pipeline -- F(4,5): 8 points, F(4,3): 6 points --
// whose multiplication stage runs on the 16-bit matrix pipe:
//
//   1. wbf_tin_k   V = B^T x along the logical W axis (NXI transformed values per 4 inputs), written once to HBM in the
//                  operand order of the matrix instruction, as
//                    NP = 3: three bf16 pieces  V = hi + mid + lo, an EXACT split of the fp32 value (round-to-nearest at
//                            each step: 8 + 8 + 8 significand bits plus the signs cover fp32's 24), or
//                    NP = 1: one fp16 value (option "conv_fp16": the fp16 path of BASELINE configs[3], UNet3D);
//   2. wbf_gemm_k  for each Winograd point xi an independent 2-D (kd, kh) convolution
//                      M_xi[n,d,h,t][co] = sum_{kd,kh,ci} V_xi[n,d+kd-p,h+kh-p,t][ci] * U_xi[kd,kh][ci][co]
//                  as an implicit GEMM on v_mfma_f32_32x32x16_{bf16,f16}; NP = 3: SIX products per fp32 product
//                      hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi      (fp32 accumulate)
//                  -- the dropped terms (mid*lo, lo*mid, lo*lo) are <= 2^-23 of |V||U|, the size of one fp32 rounding;
//                  the transformed weights U = G w are computed in double and split the same way.  The 16-bit pipe runs
//                  16x the fp32 matrix rate, so the six-product emulation is 2.67x the fp32 MFMA peak at fp32-class
//                  error; NP = 1: one product, fp16 operands, fp32 accumulate;
//   3. wbf_tout_k  y = A^T M (4 outputs per NXI points) + bias [+ dst] [PReLU], sums split-K slabs in a fixed order,
//                  optionally takes the BatchNorm statistics of y on the way (msk_conv3d_fwd_ex).
//
// Nothing is transformed inside the MFMA loop (round-1 verdict: the per-(kd,kh) register transform of
// conv_halo_wino4_k held the matrix pipe at 0.51-0.74): the GEMM stage is LDS/L2 -> MFMA only.  HBM pays for it: V is
// 3x the input bytes (NP = 3) and M 2x the output bytes -- the step used < 10 % of the HBM roof before.
//
// Layouts (16-byte slots of 8 16-bit values = one MFMA operand fragment per lane):
//   V [xi][n][t][kc][piece][khalf][DP][HP]   slot (dp, hp) = position (d = dp-2, h = hp-2), zero outside the volume
//                                            (DP/HP = tile-padded dims + 4): the GEMM stages its halo tile with plain
//                                            address arithmetic, no bounds checks; channel = kc*16 + khalf*8 + j
//   U [xi][tap][kc][piece][khalf][CN]        B fragment of lane (khalf, co) is one contiguous slot
//   M [ks][xi][n][t][d][h][CN] fp32
#include <type_traits>
#include "msk_wbf.h"
#ifndef WBF_TIN_SYNC
#define WBF_TIN_SYNC 1   // transform kernels: one barrier per W tile keeps a block's four wavefronts on the same 128-byte lines (0 = A/B)
#endif

#ifndef WBF_PROBE
#define WBF_PROBE 0   // knock-out probes of wbf_gemm_fused_k (tools/probe_fused.sh): 1 no tile refill, 2 no B loads, 4 no A reads, 8 no stores, 16 no MFMAs -- WRONG RESULTS, timing only
#endif
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int nxi_of(int K) { return K == 5 ? 8 : 6; }

__device__ __forceinline__ void split3_one(float x, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
  hi = __builtin_bit_cast(unsigned short, (__bf16)x);
  const float r = x - __uint_as_float((unsigned)hi << 16);
  mid = __builtin_bit_cast(unsigned short, (__bf16)r);
  const float r2 = r - __uint_as_float((unsigned)mid << 16);
  lo = __builtin_bit_cast(unsigned short, (__bf16)r2);
}
__device__ __forceinline__ unsigned pack_f16_pair(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}

// G (weight transform) of F(4,5) -- points 0, +-1, +-2, +-1/2, inf -- and F(4,3) -- points 0, +-1, +-2, inf
__device__ __forceinline__ double g_coef(int K, int xi, int kw) {
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
  return K == 5 ? G5[xi][kw] : G3[xi][kw];
}

// ---------------------------------------------------------------------------------------------------------
// weights: U_xi[tap = kd*K+kh][k][n] = sum_kw G[xi][kw] w(kd, kh, kw; k, n), in double, then 3 bf16 pieces or one fp16
// ---------------------------------------------------------------------------------------------------------
// One packed-weight problem (device-visible descriptor; the table lives in the context's pack cache, see below)
struct WbfPackDesc {
  const float* w;
  int A, B, swap, flip, CK, CN, KC, tsd, tsh, tsw;
  unsigned short* out;
  long xi_stride;  // elements
  float* amax;     // amax array of the weights (kWbfAmaxWays floats): NP = 2 scale, also read by the output stage
  long count;      // floats of the canonical weight tensor
};
// up to 250 table rows handled by one launch (blockIdx.y -> row), passed by value
struct WbfPackList {
  int n;
  unsigned char row[252];
};
__global__ void wbf_pack_desc_store_k(WbfPackDesc* slot, WbfPackDesc d) { *slot = d; }

template <int K, int NP>
__global__ void __launch_bounds__(256)
wbf_pack_weights_k(const WbfPackDesc* __restrict__ table, WbfPackList list, WbfPackDesc single) {
  const WbfPackDesc d = list.n < 0 ? single : table[list.row[blockIdx.y]];  // n < 0: one problem, descriptor by value
  const float* __restrict__ w = d.w;
  const int A = d.A, B = d.B, swap = d.swap, flip = d.flip, CK = d.CK, CN = d.CN, KC = d.KC, tsd = d.tsd, tsh = d.tsh, tsw = d.tsw;
  (void)A;
  unsigned short* __restrict__ out = d.out;
  const long xi_stride = d.xi_stride;
  const double wsc = NP != 3 ? (double)wbf_scale_of(d.amax) : 1.0;  // fp16 operands (NP = 2, 1): power-of-two scale into fp16 range
  constexpr int NXI = nxi_of(K), T2 = K * K, T3 = K * K * K;
  // one thread per (tap row, k, n): reads its K kw taps once, writes NXI x NP values
  const long total = (long)T2 * KC * 16 * CN;
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
    double t[K];
#pragma unroll
    for (int kw = 0; kw < K; ++kw) t[kw] = 0.0;
    if (k < CK) {
      const int ia = swap ? n : k, ib = swap ? k : n;
      const float* wp = w + ((long)ia * B + ib) * T3;
#pragma unroll
      for (int kw = 0; kw < K; ++kw) {
        const int tap = (row / K) * tsd + (row % K) * tsh + kw * tsw;  // logical (kd, kh, kw) -> canonical tap
        t[kw] = (double)wp[flip ? T3 - 1 - tap : tap];
      }
    }
    // element index = (((((xi*T2 + row)*KC + kc)*NP + piece)*2 + khalf)*CN + n)*8 + j
    const long base = ((((long)row * KC + kc) * NP * 2 + khalf) * CN + n) * 8 + j;
    const long pstep = 2L * CN * 8;
#pragma unroll
    for (int xi = 0; xi < NXI; ++xi) {
      double s_ = 0.0;
#pragma unroll
      for (int kw = 0; kw < K; ++kw) s_ = fma(g_coef(K, xi, kw), t[kw], s_);   // (explicit: both pack kernels must round alike)
      unsigned short* o = out + (long)xi * xi_stride + base;
      if (NP == 3) {
        unsigned short hi, mid, lo;
        float xf = (float)s_;
        asm volatile("" : "+v"(xf));   // (one rounding to fp32 first, in both pack kernels: the compiler may otherwise fold double -> float -> bf16)
        split3_one(xf, hi, mid, lo);
        o[0] = hi;
        o[pstep] = mid;
        o[2 * pstep] = lo;
      } else if (NP == 2) {
        // through fp32: a double -> fp16 conversion has no instruction (the compiler expands it to ~40 integer operations: this
        // kernel was VALU-bound on 16 of them per element, 0.43 ms per step); the low piece takes whatever the high piece's
        // rounding left, so the pair represents v exactly as well
        // (the empty asm keeps the compiler from folding float -> half of double -> float back into the one conversion)
        const double v = s_ * wsc;
        float vf = (float)v;
        asm volatile("" : "+v"(vf));
        const _Float16 h = (_Float16)vf;
        float rf = (float)(v - (double)(float)h);
        asm volatile("" : "+v"(rf));
        o[0] = __builtin_bit_cast(unsigned short, h);
        o[pstep] = __builtin_bit_cast(unsigned short, (_Float16)rf);
      } else {
        float vf = (float)(s_ * wsc);
        asm volatile("" : "+v"(vf));
        o[0] = __builtin_bit_cast(unsigned short, (_Float16)vf);
      }
    }
  }
}

// Round 5 -- the same packing through an LDS tile.  Above, a thread owns ONE element (row, k, n): its five kw taps are 20 bytes
// at a stride of 500 (or 500 B) bytes from its neighbour's, every 128-byte line of the weight tensor is requested by ~25
// different wavefronts (the 25 (kd, kh) rows), and the results leave as 2-byte stores.  Here a workgroup owns 16 input
// channels x 8 output channels: their 128 canonical rows of 125 taps are 8 (or 16) contiguous stretches of 8000 (4000) bytes --
// float4 loads of consecutive lanes -- and a thread then produces the 8 channels of one 16-byte operand slot for all NXI
// points and NP pieces: 16-byte stores, 1 KiB per store instruction.  Same arithmetic per element, bitwise the same image.
template <int K, int NP>
__global__ void __launch_bounds__(256)
wbf_pack_weights_lds_k(const WbfPackDesc* __restrict__ table, WbfPackList list, WbfPackDesc single) {
  constexpr int NXI = nxi_of(K), T2 = K * K, T3 = K * K * K, NN = 8;   // NN output channels per workgroup
  extern __shared__ __attribute__((aligned(16))) float wt[];           // [128 pairs][T3]
  const WbfPackDesc d = list.n < 0 ? single : table[list.row[blockIdx.y]];
  const int B = d.B, swap = d.swap, flip = d.flip, CN = d.CN, KC = d.KC, tsd = d.tsd, tsh = d.tsh, tsw = d.tsw;
  const int ntile = CN / NN;
  const double wsc = NP != 3 ? (double)wbf_scale_of(d.amax) : 1.0;
  for (int blk = blockIdx.x; blk < KC * ntile; blk += gridDim.x) {
    const int kc = blk / ntile, n0 = (blk - kc * ntile) * NN, k0 = kc * 16;
    __syncthreads();   // the previous tile's readers are done
    // canonical w[ia][ib][T3], (ia, ib) = swap ? (n, k) : (k, n): per outer index one contiguous run over the inner indices
    const int NO = swap ? NN : 16, NI = swap ? 16 : NN;      // outer / inner counts
    const int o0 = swap ? n0 : k0, i0 = swap ? k0 : n0;
    const int run4 = NI * T3 / 4;                            // float4 per run (NI * 125 * 4 bytes: a multiple of 16)
    for (int f = threadIdx.x; f < NO * run4; f += 256) {
      const int o = f / run4, r4 = f - o * run4;
      // (CK % 16 == 0 and CN % 8 == 0 are launch conditions: every row of the tile exists)
      const float4 v = *reinterpret_cast<const float4*>(d.w + ((long)(o0 + o) * B + i0) * T3 + (long)r4 * 4);
      *reinterpret_cast<float4*>(wt + (long)o * NI * T3 + r4 * 4) = v;   // pair (o, i) at (o * NI + i) * T3
    }
    __syncthreads();
    // items: (row, khalf, n_l) -> the slot of channels k0 + khalf * 8 + 0..7 at output channel n0 + n_l
    for (int it = threadIdx.x; it < T2 * 2 * NN; it += 256) {
      const int n_l = it % NN, khalf = (it / NN) & 1, row = it / (2 * NN);
      const int tap0 = (row / K) * tsd + (row % K) * tsh;
      double t[8][K];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k_l = khalf * 8 + j;
        const float* pr = wt + (long)(swap ? n_l * 16 + k_l : k_l * NN + n_l) * T3;
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
          const int tap = tap0 + kw * tsw;
          t[j][kw] = (double)pr[flip ? T3 - 1 - tap : tap];
        }
      }
      const long base = ((((long)row * KC + kc) * NP * 2 + khalf) * CN + (n0 + n_l)) * 8;
      const long pstep = 2L * CN * 8;
#pragma unroll
      for (int xi = 0; xi < NXI; ++xi) {
        unsigned short q[NP][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          double s_ = 0.0;
#pragma unroll
          for (int kw = 0; kw < K; ++kw) s_ = fma(g_coef(K, xi, kw), t[j][kw], s_);
          if (NP == 3) {
            float xf = (float)s_;
            asm volatile("" : "+v"(xf));
            split3_one(xf, q[0][j], q[NP / 2][j], q[NP - 1][j]);
          } else if (NP == 2) {
            const double v = s_ * wsc;
            float vf = (float)v;
            asm volatile("" : "+v"(vf));
            const _Float16 h = (_Float16)vf;
            float rf = (float)(v - (double)(float)h);
            asm volatile("" : "+v"(rf));
            q[0][j] = __builtin_bit_cast(unsigned short, h);
            q[NP - 1][j] = __builtin_bit_cast(unsigned short, (_Float16)rf);
          } else {
            float vf = (float)(s_ * wsc);
            asm volatile("" : "+v"(vf));
            q[0][j] = __builtin_bit_cast(unsigned short, (_Float16)vf);
          }
        }
        unsigned short* o = d.out + (long)xi * d.xi_stride + base;
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) {
          uint4 u;
          u.x = (unsigned)q[pc][0] | ((unsigned)q[pc][1] << 16);
          u.y = (unsigned)q[pc][2] | ((unsigned)q[pc][3] << 16);
          u.z = (unsigned)q[pc][4] | ((unsigned)q[pc][5] << 16);
          u.w = (unsigned)q[pc][6] | ((unsigned)q[pc][7] << 16);
          *reinterpret_cast<uint4*>(o + pc * pstep) = u;
        }
      }
    }
  }
}

// max |w| of the listed rows' weight tensors into their (zeroed) amax arrays: one atomic per block on the way blockIdx.x
__global__ void __launch_bounds__(256)
wbf_pack_absmax_k(const WbfPackDesc* __restrict__ table, WbfPackList list, WbfPackDesc single) {
  const WbfPackDesc d = list.n < 0 ? single : table[list.row[blockIdx.y]];
  float m = 0.f;
  const long n4 = d.count >> 2, stride = (long)gridDim.x * blockDim.x;
  const float4* p = reinterpret_cast<const float4*>(d.w);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 q = p[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(q.x), fabsf(q.y))), fmaxf(fabsf(q.z), fabsf(q.w)));
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < d.count; i += stride) m = fmaxf(m, fabsf(d.w[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __shared__ float sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    if (m > 0.f) (void)atomicMax(reinterpret_cast<unsigned*>(d.amax) + blockIdx.x % kWbfAmaxWays, __float_as_uint(m));
  }
}

// ---------------------------------------------------------------------------------------------------------
// stage 1: input transform (+ split)
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
// F(4,3) input transform (the six points of F(2,5): same B^T)
__device__ __forceinline__ void bt6(const float d0, const float d1, const float d2, const float d3, const float d4,
                                    const float d5, float (&v)[8]) {
  v[0] = 4.f * d0 + (d4 - 5.f * d2);
  const float pa = d4 - 4.f * d2, qa = d3 - 4.f * d1;
  v[1] = pa + qa;
  v[2] = pa - qa;
  const float pb = d4 - d2, qb = 2.f * (d3 - d1);
  v[3] = pb + qb;
  v[4] = pb - qb;
  v[5] = 4.f * d1 + (d5 - 5.f * d3);
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
__device__ __forceinline__ void at6(const float e0, const float e1, const float e2, const float e3, float (&v)[8]) {
  v[0] = e0;
  v[5] = e3;
  const float s02 = e0 + e2, s13 = e1 + e3;
  v[1] = s02 + s13;
  v[2] = s02 - s13;
  const float p = e0 + 4.f * e2, q = 2.f * e1 + 8.f * e3;
  v[3] = p + q;
  v[4] = p - q;
}

template <int MODE, int K>
__device__ __forceinline__ void tin_transform(const float (&w)[8], float (&v)[8]) {
  if (MODE == 0) {
    if (K == 5) bt8(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], v);
    else bt6(w[0], w[1], w[2], w[3], w[4], w[5], v);
  } else {
    if (K == 5) at8(w[0], w[1], w[2], w[3], v);
    else at6(w[0], w[1], w[2], w[3], v);
  }
}

// thread = (n, padded position (dp, hp), 8-channel group); walks t = 0 .. T-1 (MODE 0: with a sliding (K+3)-wide W
// window, x is read once).
template <int MODE, int K, int NP>
__global__ void __launch_bounds__(256)
wbf_tin_k(WbfTinArgs a) {
  constexpr int NXI = nxi_of(K), WIN = MODE == 0 ? K + 3 : 4, PADW = (K - 1) / 2, KEEP = WIN - 4;
  // lane mapping (A/B, option "wbf_tin_map"): 0 = 4 channel groups fastest (reads: full 128-byte lines per 4 lanes;
  // stores: 4 runs of 256 B per wavefront), 1 = one channel group per wavefront (stores: one 1 KiB run; reads: 32 of
  // every 128 bytes per lane, the rest of the line goes to the block's other wavefronts through L1/L2)
  if (a.amax_copy && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < kWbfAmaxWays) a.amax_copy[threadIdx.x] = a.amax[threadIdx.x];
  const int t0 = blockIdx.z * a.t_per, t1 = min(a.T, t0 + a.t_per);   // this workgroup's W tiles
  const int cgl = a.lane_map ? (threadIdx.x >> 6) : (threadIdx.x & 3), pl = a.lane_map ? (threadIdx.x & 63) : (threadIdx.x >> 2);
  const int ncgb = a.CK >> 5;
  const int cgb = blockIdx.x % ncgb, pb = blockIdx.x / ncgb;
  const int pos = pb * 64 + pl;
  const int n = blockIdx.y;
  const int cg = cgb * 4 + cgl, kc = cg >> 1, khalf = cg & 1;
  float cmx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // NP = 2: max |source value| per channel (a.cmax)
  const bool want_cmax = MODE == 1 && NP == 2 && a.cmax != nullptr;   // the dy side only (msk_wgrad_wbf.hip)
  auto cmax_of = [&](const float4& e0, const float4& e1) {
    cmx[0] = fmaxf(cmx[0], fabsf(e0.x)); cmx[1] = fmaxf(cmx[1], fabsf(e0.y)); cmx[2] = fmaxf(cmx[2], fabsf(e0.z)); cmx[3] = fmaxf(cmx[3], fabsf(e0.w));
    cmx[4] = fmaxf(cmx[4], fabsf(e1.x)); cmx[5] = fmaxf(cmx[5], fabsf(e1.y)); cmx[6] = fmaxf(cmx[6], fabsf(e1.z)); cmx[7] = fmaxf(cmx[7], fabsf(e1.w));
  };
  if (pos < a.DP * a.HP) {
  const int dp = pos / a.HP, hp = pos - dp * a.HP;
  const int d = dp - 2, h = hp - 2;
  const bool live = d >= 0 && d < a.LD && h >= 0 && h < a.LH;
  const long plane = (long)a.DP * a.HP * 16;
  char* vb = a.V + (((long)n * a.T * a.KC + kc) * 2 * NP + khalf) * plane + (long)pos * 16;
  const long tstep = (long)a.KC * 2 * NP * plane;
  const float* xb = a.src + ((long)n * a.svn + (long)d * a.svd + (long)h * a.svh) * a.sld + cg * 8;
  const long wstep = (long)a.svw * a.sld;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int creal = a.c_real > 0 ? a.c_real : a.CK;          // channels that exist in the source (the others are zeros)
  const bool q0ok = cg * 8 < creal, q1ok = cg * 8 + 4 < creal;
  const float sc2 = NP != 3 ? wbf_scale_of(a.amax) : 1.f;  // fp16 operands: the tensor's power-of-two scale (round 3: also the single-fp16 form -- gradients of ~1e-7 sit inside fp16's subnormal range)
  (void)sc2;

  float4 win[WIN][2];
#pragma unroll
  for (int j = 0; j < WIN; ++j) {
    const int w = 4 * t0 + (MODE == 0 ? j - PADW : j);
    if (live && w >= 0 && w < a.LW) {
      const float4* p = reinterpret_cast<const float4*>(xb + w * wstep);
      win[j][0] = q0ok ? p[0] : z4;
      win[j][1] = q1ok ? p[1] : z4;
      if (want_cmax) cmax_of(win[j][0], win[j][1]);
    } else {
      win[j][0] = win[j][1] = z4;
    }
  }
  for (int t = t0; t < t1; ++t) {
#if WBF_TIN_SYNC
    if (a.lane_map) __builtin_amdgcn_s_barrier();   // see wbf_tin_dual_k: the four quarters of a line are requested together
#endif
    float4 nxt[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int w = 4 * (t + 1) + (MODE == 0 ? KEEP - PADW : 0) + j;  // the part of tile t + 1 not yet in registers
      if (live && t + 1 < t1 && w < a.LW) {
        const float4* p = reinterpret_cast<const float4*>(xb + w * wstep);
        nxt[j][0] = q0ok ? p[0] : z4;
        nxt[j][1] = q1ok ? p[1] : z4;
        if (want_cmax) cmax_of(nxt[j][0], nxt[j][1]);
      } else {
        nxt[j][0] = nxt[j][1] = z4;
      }
    }
    float v[8][8];  // [channel][xi]
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float wx[8], wy[8], wz[8], ww[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 e = j < WIN ? win[j < WIN ? j : 0][q] : z4;
        wx[j] = e.x; wy[j] = e.y; wz[j] = e.z; ww[j] = e.w;
      }
      tin_transform<MODE, K>(wx, v[q * 4 + 0]);
      tin_transform<MODE, K>(wy, v[q * 4 + 1]);
      tin_transform<MODE, K>(wz, v[q * 4 + 2]);
      tin_transform<MODE, K>(ww, v[q * 4 + 3]);
    }
    char* vt = vb + t * tstep;
#pragma unroll
    for (int xi = 0; xi < NXI; ++xi) {
      char* o = vt + (long)xi * a.v_xi;
      if (NP == 3) {
        uint4 hi, mid, lo;
        wbf_split3_pair(v[0][xi], v[1][xi], hi.x, mid.x, lo.x);
        wbf_split3_pair(v[2][xi], v[3][xi], hi.y, mid.y, lo.y);
        wbf_split3_pair(v[4][xi], v[5][xi], hi.z, mid.z, lo.z);
        wbf_split3_pair(v[6][xi], v[7][xi], hi.w, mid.w, lo.w);
        *reinterpret_cast<uint4*>(o) = hi;
        *reinterpret_cast<uint4*>(o + 2 * plane) = mid;
        *reinterpret_cast<uint4*>(o + 4 * plane) = lo;
      } else if (NP == 2) {
        uint4 hi, lo;
        wbf_split2hs_pair(v[0][xi] * sc2, v[1][xi] * sc2, hi.x, lo.x);
        wbf_split2hs_pair(v[2][xi] * sc2, v[3][xi] * sc2, hi.y, lo.y);
        wbf_split2hs_pair(v[4][xi] * sc2, v[5][xi] * sc2, hi.z, lo.z);
        wbf_split2hs_pair(v[6][xi] * sc2, v[7][xi] * sc2, hi.w, lo.w);
        *reinterpret_cast<uint4*>(o) = hi;
        *reinterpret_cast<uint4*>(o + 2 * plane) = lo;
      } else {
        uint4 hv;
        hv.x = pack_f16_pair(v[0][xi] * sc2, v[1][xi] * sc2);
        hv.y = pack_f16_pair(v[2][xi] * sc2, v[3][xi] * sc2);
        hv.z = pack_f16_pair(v[4][xi] * sc2, v[5][xi] * sc2);
        hv.w = pack_f16_pair(v[6][xi] * sc2, v[7][xi] * sc2);
        *reinterpret_cast<uint4*>(o) = hv;
      }
    }
#pragma unroll
    for (int j = 0; j < KEEP; ++j) {
      win[j][0] = win[j + 4][0];
      win[j][1] = win[j + 4][1];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      win[KEEP + j][0] = nxt[j][0];
      win[KEEP + j][1] = nxt[j][1];
    }
  }
  }  // pos < DP * HP
  if (want_cmax) wbf_cmax_commit(a.cmax, cg, cmx, a.lane_map != 0);
}

// Both transforms of dy = BatchNorm/PReLU-backward(y, dout) (WbfBnBwd, msk_wbf.h) in one pass: the thread mapping and
// the sliding window of wbf_tin_k<0> (one 8-channel group per wavefront, so the per-channel coefficients are scalar
// registers), dy evaluated as the window is filled, tile t's own four positions (window slots PADW .. PADW+3) feed the
// A dy transform.  Reads 8 B and writes 2 x 12 B per element instead of 12 + 16 + 16 B of the three-kernel form.
struct DualArgs {
  WbfTinArgs t;
  const float* y;
  int yld;
  const float* dout;
  int dld;
  const float *scale, *shift, *alpha, *mean, *invstd, *sums;
  float invM;
  int C;
  char* Y;
  long y_xi;
  const float* amax;
  const float* maxes;  // see WbfBnBwd
  float* y_cmax;       // see WbfBnBwd
  int coef_stride, sums_stride;
  int bound_shift;     // debug option "dy_bound_shift": the bound of max |dy| times 2^shift (tools/diag_fullsize_flip.py)
};

template <int K, int NP>
__device__ __forceinline__ void store_xi(char* o, long plane, const float (&v)[8][8], int xi, float sc2) {
  if (NP == 3) {
    uint4 hi, mid, lo;
    wbf_split3_pair(v[0][xi], v[1][xi], hi.x, mid.x, lo.x);
    wbf_split3_pair(v[2][xi], v[3][xi], hi.y, mid.y, lo.y);
    wbf_split3_pair(v[4][xi], v[5][xi], hi.z, mid.z, lo.z);
    wbf_split3_pair(v[6][xi], v[7][xi], hi.w, mid.w, lo.w);
    *reinterpret_cast<uint4*>(o) = hi;
    *reinterpret_cast<uint4*>(o + 2 * plane) = mid;
    *reinterpret_cast<uint4*>(o + 4 * plane) = lo;
  } else if (NP == 2) {
    uint4 hi, lo;
    wbf_split2hs_pair(v[0][xi] * sc2, v[1][xi] * sc2, hi.x, lo.x);
    wbf_split2hs_pair(v[2][xi] * sc2, v[3][xi] * sc2, hi.y, lo.y);
    wbf_split2hs_pair(v[4][xi] * sc2, v[5][xi] * sc2, hi.z, lo.z);
    wbf_split2hs_pair(v[6][xi] * sc2, v[7][xi] * sc2, hi.w, lo.w);
    *reinterpret_cast<uint4*>(o) = hi;
    *reinterpret_cast<uint4*>(o + 2 * plane) = lo;
  } else {
    uint4 hv;
    hv.x = pack_f16_pair(v[0][xi] * sc2, v[1][xi] * sc2);
    hv.y = pack_f16_pair(v[2][xi] * sc2, v[3][xi] * sc2);
    hv.z = pack_f16_pair(v[4][xi] * sc2, v[5][xi] * sc2);
    hv.w = pack_f16_pair(v[6][xi] * sc2, v[7][xi] * sc2);
    *reinterpret_cast<uint4*>(o) = hv;
  }
}

#ifndef WBF_DUAL_LB
#define WBF_DUAL_LB 0   // 3 = force three wavefronts per SIMD: the one-kernel form then spills 8 registers and is 6 % SLOWER (A/B, round 4)
#endif
template <int K, int NP, bool WV, bool WY>  // which of the two transforms are written (each stream may take its own)
#if WBF_DUAL_LB > 0
__global__ void __launch_bounds__(256, WBF_DUAL_LB)
#else
__global__ void __launch_bounds__(256)
#endif
wbf_tin_dual_k(DualArgs b) {
  constexpr int NXI = nxi_of(K), WIN = K + 3, PADW = (K - 1) / 2, KEEP = WIN - 4;
  const WbfTinArgs& a = b.t;
  const int cgl = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), pl = threadIdx.x & 63;
  const int ncgb = a.CK >> 5;
  const int cgb = blockIdx.x % ncgb, pb = blockIdx.x / ncgb;
  const int pos = pb * 64 + pl;
  const int n = blockIdx.y;
  const int cg = cgb * 4 + cgl, kc = cg >> 1, khalf = cg & 1;
  // per-channel coefficients of this wavefront's 8 channels: wave-uniform -> scalar registers
  float sc[8], sf[8], al[8], mu[8], is[8], s1[8], s2[8];
  const long co = (long)n * b.coef_stride, so = (long)n * b.sums_stride;   // per-sample statistics (InstanceNorm): this block's sample
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg * 8 + j;
    sc[j] = b.scale[co + c];
    sf[j] = b.shift[co + c];
    al[j] = b.alpha ? b.alpha[c] : 1.f;
    mu[j] = b.mean[co + c];
    is[j] = b.invstd[co + c];
    s1[j] = b.sums[so + c] * b.invM;
    s2[j] = b.sums[so + b.C + c] * b.invM;
  }
  // NP = 2: power-of-two scale of dy from (a bound of) its maximum: given (b.amax), or evaluated here (b.maxes)
  float sc2 = 1.f;
  if (NP != 3) {
    if (b.maxes) {
      __shared__ float shb[3][4];
      float ma = 0.f, mb = 0.f, mc = 0.f;
      const int sets = (b.coef_stride || b.sums_stride) ? a.N : 1;   // every block bounds the WHOLE tensor: all samples' coefficients
      for (int i = threadIdx.x; i < sets * b.C; i += 256) {
        const int sn = i / b.C, c = i - sn * b.C;
        ma = fmaxf(ma, fabsf(b.scale[(long)sn * b.coef_stride + c]));
        mb = fmaxf(mb, fabsf(b.sums[(long)sn * b.sums_stride + c] * b.invM));
        mc = fmaxf(mc, fabsf(b.sums[(long)sn * b.sums_stride + b.C + c] * b.invM));
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        ma = fmaxf(ma, __shfl_xor(ma, o, 64));
        mb = fmaxf(mb, __shfl_xor(mb, o, 64));
        mc = fmaxf(mc, __shfl_xor(mc, o, 64));
      }
      if (pl == 0) { shb[0][cgl] = ma; shb[1][cgl] = mb; shb[2][cgl] = mc; }
      __syncthreads();
      ma = fmaxf(fmaxf(shb[0][0], shb[0][1]), fmaxf(shb[0][2], shb[0][3]));
      mb = fmaxf(fmaxf(shb[1][0], shb[1][1]), fmaxf(shb[1][2], shb[1][3]));
      mc = fmaxf(fmaxf(shb[2][0], shb[2][1]), fmaxf(shb[2][2], shb[2][3]));
      const float bound = ldexpf(ma * (wbf_amax_of(b.maxes) + mb + wbf_amax_of(b.maxes + kWbfAmaxWays) * mc), b.bound_shift);
      if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) const_cast<float*>(b.amax)[0] = bound;  // zeroed ring array
      sc2 = wbf_scale_from(bound);
    } else {
      sc2 = wbf_scale_of(b.amax);
    }
  }
  (void)sc2;
  float cmx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // NP = 2, WY: max |dy| per channel (b.y_cmax)
  const bool want_cmax = NP == 2 && WY && b.y_cmax != nullptr;
  if (pos < a.DP * a.HP) {
  const int dp = pos / a.HP, hp = pos - dp * a.HP;
  const int d = dp - 2, h = hp - 2;
  const bool live = d >= 0 && d < a.LD && h >= 0 && h < a.LH;
  const long plane = (long)a.DP * a.HP * 16;
  const long voff = (((long)n * a.T * a.KC + kc) * 2 * NP + khalf) * plane + (long)pos * 16;
  char* vb = a.V + voff;
  char* yb2 = b.Y + voff;
  const long tstep = (long)a.KC * 2 * NP * plane;
  const long vox0 = (long)n * a.svn + (long)d * a.svd + (long)h * a.svh;
  const float* xb = b.y + vox0 * b.yld + cg * 8;
  const float* gb = b.dout + vox0 * b.dld + cg * 8;
  const long xstep = (long)a.svw * b.yld, gstep = (long)a.svw * b.dld;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

  // dy of 8 channels at logical position w (zero outside the volume)
  auto dy_at = [&](int w, float4& o0, float4& o1) {
    if (!(live && w >= 0 && w < a.LW)) {
      o0 = o1 = z4;
      return;
    }
    const float4* px = reinterpret_cast<const float4*>(xb + w * xstep);
    const float4* pg = reinterpret_cast<const float4*>(gb + w * gstep);
    const float4 x0 = px[0], x1 = px[1], g0 = pg[0], g1 = pg[1];
    const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float dd = gv[j];
      const float u = fmaf(xv[j], sc[j], sf[j]);
      if (!(u > 0.f)) dd *= al[j];
      const float xh = (xv[j] - mu[j]) * is[j];
      r[j] = sc[j] * (dd - s1[j] - xh * s2[j]);
      if (want_cmax) cmx[j] = fmaxf(cmx[j], fabsf(r[j]));
    }
    o0 = make_float4(r[0], r[1], r[2], r[3]);
    o1 = make_float4(r[4], r[5], r[6], r[7]);
  };

  float4 win[WIN][2];
  const int t0 = blockIdx.z * a.t_per, t1 = min(a.T, t0 + a.t_per);   // this workgroup's W tiles
#pragma unroll
  for (int j = 0; j < WIN; ++j) dy_at(4 * t0 + j - PADW, win[j][0], win[j][1]);
  for (int t = t0; t < t1; ++t) {
#if WBF_TIN_SYNC
    // the block's four wavefronts read the four 32-byte quarters of the same 128-byte lines: kept in step, the quarters reach
    // L2 together and the line is fetched once (PMC, round 3: this kernel fetched 1.67x its inputs; A/B round 4: transforms
    // bucket 3.28 -> 3.18 ms, step -0.13 ms).  Wavefronts beyond the plane have left the kernel: the barrier does not wait for them.
    __builtin_amdgcn_s_barrier();
#endif
    float4 nxt[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int w = 4 * (t + 1) + KEEP - PADW + j;  // the part of tile t + 1 not yet in registers
      if (t + 1 < t1) dy_at(w, nxt[j][0], nxt[j][1]);
      else nxt[j][0] = nxt[j][1] = z4;
    }
    float v[8][8];  // [channel][xi]
    if (WV) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float wx[8], wy[8], wz[8], ww[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float4 e = j < WIN ? win[j < WIN ? j : 0][q] : z4;
          wx[j] = e.x; wy[j] = e.y; wz[j] = e.z; ww[j] = e.w;
        }
        tin_transform<0, K>(wx, v[q * 4 + 0]);
        tin_transform<0, K>(wy, v[q * 4 + 1]);
        tin_transform<0, K>(wz, v[q * 4 + 2]);
        tin_transform<0, K>(ww, v[q * 4 + 3]);
      }
#pragma unroll
      for (int xi = 0; xi < NXI; ++xi) store_xi<K, NP>(vb + t * tstep + (long)xi * a.v_xi, plane, v, xi, sc2);
    }
    if (WY) {
      // A dy of the tile's own positions 4t .. 4t+3 = window slots PADW .. PADW+3
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float wx[8], wy[8], wz[8], ww[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 e = win[PADW + j][q];
          wx[j] = e.x; wy[j] = e.y; wz[j] = e.z; ww[j] = e.w;
        }
#pragma unroll
        for (int j = 4; j < 8; ++j) wx[j] = wy[j] = wz[j] = ww[j] = 0.f;
        tin_transform<1, K>(wx, v[q * 4 + 0]);
        tin_transform<1, K>(wy, v[q * 4 + 1]);
        tin_transform<1, K>(wz, v[q * 4 + 2]);
        tin_transform<1, K>(ww, v[q * 4 + 3]);
      }
#pragma unroll
      for (int xi = 0; xi < NXI; ++xi) store_xi<K, NP>(yb2 + t * tstep + (long)xi * b.y_xi, plane, v, xi, sc2);
    }
#pragma unroll
    for (int j = 0; j < KEEP; ++j) {
      win[j][0] = win[j + 4][0];
      win[j][1] = win[j + 4][1];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      win[KEEP + j][0] = nxt[j][0];
      win[KEEP + j][1] = nxt[j][1];
    }
  }
  }  // pos < DP * HP
  if (want_cmax) wbf_cmax_commit(b.y_cmax, cg, cmx, true);
}

// ---------------------------------------------------------------------------------------------------------
// stage 2: per-xi 2-D convolution as an implicit GEMM on the 16-bit matrix pipe
// ---------------------------------------------------------------------------------------------------------
struct GemmArgs {
  const char* V;
  const char* U;
  float* M;
  int N, T, KC, CN, LD, LH, DP, HP;
  int tiles_d, tiles_h, ngrp, ksplit, kc_per;
  int tpb, nblk;       // tiles per workgroup, total tiles
  long v_xi, v_plane;  // bytes
  long u_xi;           // bytes
  long m_xi;           // floats between xi planes of M (= N*T*LD*LH*CN); split slabs are NXI*m_xi apart
};

__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned uniform_bytes) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_bytes, (int)uniform_bytes, 0));
}

typedef int i32x4 __attribute__((ext_vector_type(4)));
// One LDS-DMA request (64 lanes x 16 B, global -> LDS at the wave-uniform byte address lds_addr + lane * 16) issued BEHIND THE
// COMPILER'S BACK: see wbf_gemm_fused_k, pipelined form.  m0 = LDS address of the request (the instruction's implicit operand).
__device__ __forceinline__ void wbf_dma16(i32x4 rs, unsigned lds_addr, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

#define WBF_MFMA(acc, av, bv) \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc, 0, 0, 0)
#define WBF_MFMA_H(acc, av, bv) \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bv), acc, 0, 0, 0)

// Row r (0..31) of an A fragment -> tile position (a, b) inside the fragment's 32/TH x TH block of (d, h) positions.
// ds_read_b128 is serviced in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32 for the upper half), 16 lanes x
// 16 B = one 256-byte bank row per LDS cycle (MI355X_MICROARCH.md, LDS): the 16 slots of a group must be distinct
// modulo 16.  With the halo-tile row pitch TH + K - 1 (20 or 12 slots) the natural order r -> (r / TH, r % TH) puts two
// lanes of every group on the same banks (PMC: half of the LDS cycles were bank-conflict cycles); these orders do not.
// The output rows of the MFMA follow the same map (epilogue).
template <int TH>
__device__ __forceinline__ void frag_pos(int r, int& a, int& b) {
  if (TH == 16) {
    // group 1 -> row 0 cols 0-7 + row 1 cols 4-11 (slot values 0-7, 8-15 mod 16 at pitch 20); group 2 -> the rest
    if (r < 4) { a = 0; b = r; }
    else if (r < 12) { a = 0; b = r + 4; }
    else if (r < 16) { a = 0; b = r - 8; }
    else if (r < 20) { a = 1; b = r - 4; }
    else if (r < 28) { a = 1; b = r - 16; }
    else { a = 1; b = r - 28; }
  } else if (TH == 8) {
    // pitch 12: rows 0 and 2 hold slot values 0-7 and 8-15 (group 1), rows 1 and 3 hold 12-15,0-3 and 4-11 (group 2)
    if (r < 4) { a = 0; b = r; }
    else if (r < 12) { a = 1; b = r - 4; }
    else if (r < 16) { a = 0; b = r - 8; }
    else if (r < 20) { a = 3; b = r - 16; }
    else if (r < 28) { a = 2; b = r - 20; }
    else { a = 3; b = r - 24; }
  } else {
    a = r / TH;  // TH = 32: one row of 32 contiguous slots is conflict-free as it is
    b = r % TH;
  }
}

// Workgroup = 4 wavefronts as WM (rows) x WN (column groups of 32); a wavefront owns MR row fragments of 32 positions
// and ONE 32-channel column fragment: its B fragments come straight from L2/L1 (NP slots per tap and 16-channel chunk,
// used by MR*6 (or MR) MFMAs), A fragments from the LDS halo tile (TD+K-1) x (TH+K-1) that the whole workgroup shares
// and every one of the K*K taps re-reads.  The tile is filled by LDS-DMA (buffer_load ... lds, 16 B per lane), no registers.
// BPF (round 5): how many taps ahead the B (weight) fragments are requested.  One tap ahead leaves ~200 cycles of matrix
// instructions between a request and its use -- an L2 round trip is several times that -- which the big launches hide behind
// three or four resident workgroups per CU; the deep levels (<= 2048 workgroups of one or two chunks each) run one or two
// per CU and waited on every tap.  BPF = 4: a ring of five fragment sets (+24 registers).
template <int MR, int WM, int WN, int TD, int TH, int K, int NP, int BPF = 1>
__global__ void __launch_bounds__(WM * WN * 64)
wbf_gemm_k(GemmArgs a) {
  static_assert((WM * WN == 4 || WM * WN == 2) && WM * MR * 32 == TD * TH, "tile shape");
  constexpr int NT = WM * WN * 64;  // workgroup size: 4 wavefronts, or 2 with twice the row fragments each (half the B loads per MFMA)
  constexpr int NXI = nxi_of(K), T2 = K * K, PADK = (K - 1) / 2;
  constexpr int HDt = TD + K - 1, HPt = TH + K - 1, NSLOT = HDt * HPt, NPL = 2 * NP, NIT = NPL * NSLOT, ROUNDS = (NIT + NT - 1) / NT;
  __shared__ uint4 lds[ROUNDS * NT];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  // a workgroup walks `tpb` tiles, b = blockIdx.x + i * gridDim.x (gridDim.x is a multiple of 8: the Winograd point and with
  // it the XCD stay the same): 32 768 five-microsecond workgroups per launch were paying for their dispatch
#pragma unroll 1
  for (int rep = 0; rep < a.tpb; ++rep) {
  int b = blockIdx.x + rep * gridDim.x;
  if (b >= a.nblk) break;
  if (rep > 0) __syncthreads();  // the previous tile's LDS reads are done (the chunk loop opens with a barrier as well)
  const int xi = b % NXI;  // block b runs on XCD b % 8: with 8 points one Winograd point per XCD, its weights stay in that L2
  b /= NXI;
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

  // LDS-DMA: item it = r*NT + tid -> (plane pk = piece*2 + khalf, slot); slot = row*HPt + col of the halo tile
  unsigned voff[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    int it = r * NT + tid;
    if (it >= NIT) it = 0;  // the last round overshoots: re-read item 0 into the unused tail of the image
    const int pk = it / NSLOT, slot = it - pk * NSLOT;
    const int row = slot / HPt, col = slot - row * HPt;
    voff[r] = (unsigned)(pk * a.v_plane + ((long)row * a.HP + col) * 16);
  }
  const char* vtile = a.V + (long)xi * a.v_xi + ((long)(n * a.T + t) * a.KC) * NPL * a.v_plane +
                      ((long)(tdi * TD + 2 - PADK) * a.HP + thi * TH + 2 - PADK) * 16;
  const __amdgpu_buffer_rsrc_t vres = __builtin_amdgcn_make_buffer_rsrc((void*)vtile, 0, 0xFFFFFFF0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t ures =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.U + (long)xi * a.u_xi), 0, 0xFFFFFFF0u, 0x00020000);
  const unsigned ulane = (unsigned)(lh * a.CN + (grp * WN + wn) * 32 + li) * 16u;
  const unsigned ustep = (unsigned)a.CN * 32u;          // bytes between pieces (2 khalf planes of CN slots)
  const unsigned uchunk = (unsigned)NP * ustep;         // bytes between 16-channel chunks
  const unsigned utap = (unsigned)a.KC * uchunk;        // bytes between taps

  // A rows of this lane: fragment f = wm*MR + mr covers tile rows [32 f, 32 f + 32), row -> (dd, hh) = (r / TH, r % TH)
  int arow[MR];
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
    int fa, fb;
    frag_pos<TH>(li, fa, fb);
    arow[mr] = lh * NSLOT + ((wm * MR + mr) * (32 / TH) + fa) * HPt + fb;
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
    const unsigned vsoff = (unsigned)(kc * NPL * a.v_plane);
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(vres, (__attribute__((address_space(3))) void*)(lds + r * NT + wave * 64), 16,
                                               (int)voff[r], (int)vsoff, 0, 0);
    const unsigned ukc = (unsigned)kc * uchunk;
    constexpr int BR = BPF + 1;   // ring of B fragment sets: tap t lives in bq[t % BR]
    uint4 bq[BR][NP];
#pragma unroll
    for (int t0 = 0; t0 < BPF && t0 < T2; ++t0)
#pragma unroll
      for (int p = 0; p < NP; ++p) bq[t0][p] = buf_load16(ures, ulane, (unsigned)t0 * utap + ukc + p * ustep);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // K*K taps, fully unrolled, software pipelined by hand: the A fragments (LDS) and B fragments (L2) of tap + 1 are
    // requested before the MFMAs of tap, and the scheduler may not move them (sched_barrier).  Consecutive MFMAs alternate
    // between the accumulators.  (Measured: no gain over the compiler's own schedule -- the kernel is power-limited.)
    uint4 aq[2][MR][NP];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
      const uint4* ap = lds + arow[mr];
#pragma unroll
      for (int p = 0; p < NP; ++p) aq[0][mr][p] = ap[p * 2 * NSLOT];
    }
#pragma unroll
    for (int tap = 0; tap < T2; ++tap) {
      const int cur = tap & 1, nx = cur ^ 1;
      const int bcur = tap % BR;
      if (tap + BPF < T2) {
        const unsigned ub = (unsigned)(tap + BPF) * utap + ukc;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          bq[(tap + BPF) % BR][p] = buf_load16(ures, ulane, ub + p * ustep);
        }
      }
      if (tap + 1 < T2) {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
          const uint4* ap = lds + arow[mr] + ((tap + 1) / K) * HPt + ((tap + 1) % K);
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            aq[nx][mr][p] = ap[p * 2 * NSLOT];
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (NP == 3) {
        // small terms first: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][NP - 1], bq[bcur][0]);
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][0], bq[bcur][NP - 1]);
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][NP / 2], bq[bcur][NP / 2]);
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][NP / 2], bq[bcur][0]);
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][0], bq[bcur][NP / 2]);
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][0], bq[bcur][0]);
      } else if (NP == 2) {
        // small terms first: lo*hi, hi*lo, hi*hi.  The activation's low piece is stored times 2^11 (msk_wbf.h): its partner
        // is the weight's high piece times 2^-11, made here
        const uint4 bdown = wbf_hi_down(bq[bcur][0]);
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) WBF_MFMA_H(acc[mr], aq[cur][mr][NP - 1], bdown);
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) WBF_MFMA_H(acc[mr], aq[cur][mr][0], bq[bcur][NP - 1]);
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) WBF_MFMA_H(acc[mr], aq[cur][mr][0], bq[bcur][0]);
      } else {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) WBF_MFMA_H(acc[mr], aq[cur][mr][0], bq[bcur][0]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // store M[ks][xi][n][t][d][h][co]
  const int co = (grp * WN + wn) * 32 + li;
  float* mbase = a.M + ((long)ks * NXI + xi) * a.m_xi + ((long)(n * a.T + t) * a.LD) * a.LH * a.CN + co;
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      int fa, fb;
      frag_pos<TH>((j & 3) + 8 * (j >> 2) + 4 * lh, fa, fb);
      const int d = tdi * TD + (wm * MR + mr) * (32 / TH) + fa, h = thi * TH + fb;
      if (d < a.LD && h < a.LH) mbase[((long)d * a.LH + h) * a.CN] = acc[mr][j];
    }
  }
  }  // tiles of this workgroup
}

// ---------------------------------------------------------------------------------------------------------
// stages 2 + 3 in one kernel (round 3): a workgroup owns a (TD x TH) position tile of one (n, W-tile) plane and walks ALL
// Winograd points; the per-point product M_xi lives in the accumulators only and is folded into the four output
// accumulators  y_j += AT[j][xi] * M_xi  (exact power-of-two coefficients) as soon as its K loop ends.  M never reaches HBM
// (it was 2x the output bytes written + read back by wbf_tout_k), the output transform, bias / accumulate / PReLU epilogue
// and the BatchNorm statistics of y run on the registers.  Eligible when one workgroup covers all output channels
// (ngrp == 1: CN <= 64 with the tile variants above), no split-K, and the packed weights of all points stay in an XCD's
// L2 (<= 3.5 MB): the 32- and 64-channel levels = 88 % of the transform bytes of a VNet step.  The block -> tile map hands
// every XCD a contiguous range of tiles (neighbouring halo tiles share their overlap in that XCD's L2).
// 4 x MR x 16 output + MR x 16 product accumulators + fragments = ~230 VGPRs: two wavefronts per SIMD.
// ---------------------------------------------------------------------------------------------------------
struct FusedArgs {
  GemmArgs g;  // M unused
  float* dst;
  int dld;
  long dvn;
  int dvd, dvh, dvw;
  const float* bias;
  const float* prelu;
  int accumulate;
  const float* in_amax;
  const float* w_amax;
  int scaled;
  float* stat_partial;  // STATS: [tiles][CN][3] = (n, mean, M2) of the stored values
  int per_xcd;          // tiles per XCD
  int LW;               // extent of the transform axis: the last W tile may reach beyond it (ragged: stores and statistics stop there)
  // round 5 (msk_conv3d_bwd_bnact_split): the accumulating data gradient of the layer behind a zero-copy concat READS the old
  // values from the interleaved buffer `dst` (whole lines) but STORES channels [0, csplit) to the dense tensor st_lo and
  // [csplit, 2 csplit) to st_hi (voxel stride csplit each; dld == 2 csplit): the consumers of the two halves -- the up-convolution's
  // and the skip's backward -- then read dense 64-byte voxels instead of one half of every 128-byte line.  Null: store to dst.
  float* st_lo;
  float* st_hi;
  int csplit;
  const float* acc_src;   // accumulate: the old values come from here (same geometry / voxel stride as dst) instead of dst; null = dst
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

// AT[j][xi] of F(4,5) (points 0, +-1, +-2, +-1/2, inf) and F(4,3) (0, +-1, +-2, inf): column xi as (c0, c1, c2, c3)
template <int K>
__device__ __forceinline__ void at_column(int xi, float& c0, float& c1, float& c2, float& c3) {
  const int last = K == 5 ? 7 : 5;
  c0 = xi == last ? 0.f : 1.f;
  if (xi == 0) { c1 = c2 = c3 = 0.f; return; }
  if (xi == last) { c1 = c2 = 0.f; c3 = 1.f; return; }
  const float sg = (xi & 1) ? 1.f : -1.f;
  const int pr = (xi - 1) >> 1;                       // 0: +-1, 1: +-2, 2: +-1/2
  const float p = pr == 0 ? 1.f : (pr == 1 ? 2.f : 0.5f);
  c1 = sg * p;
  c2 = p * p;
  c3 = sg * p * p * p;
}

#ifndef WBF_FUSED_BPF
#define WBF_FUSED_BPF 3   // B (weight) fragments of the one-kernel form are requested this many taps ahead (fp16 x 2 pieces: +24 VGPRs over 1)
#endif
// BL (round 6, 32 output channels = ONE column fragment, WN == 1): the B (weight) fragments of a stage -- K*K taps x NP pieces
// of 1 KB -- come through LDS as well, fetched ONCE per workgroup by LDS-DMA next to the halo tile, instead of by every
// wavefront from L1 inside the tap loop (4 x the bytes: all four wavefronts of a CN = 32 tile read the same fragments).
// Knock-out probes (profiles/r06_fused_probes.txt): without the in-loop B loads the kernel ran 15 % faster, a deeper
// prefetch ring (BPF 1 -> 3) recovered only 4 % -- the texture path (B loads + tile fill, ~70 % busy at the full matrix rate),
// not the latency.  The tap loop is then LDS -> MFMA only (ds_read_b128 at 256 B/clk/CU: A + B reads = half of that at the
// full matrix rate).  LDS: 25 KB tile + 50 KB weights per workgroup (exact sizes: the last fill round is wave-granular), two
// workgroups per CU = 150 of 160 KB.
template <int MR, int WM, int WN, int TD, int TH, int K, int NP, bool STATS, int BPF = (NP == 2 ? WBF_FUSED_BPF : 1), int BL = 0>
__global__ void __launch_bounds__(WM * WN * 64, 2)
wbf_gemm_fused_k(FusedArgs f) {
  static_assert(WM * WN == 4 && WM * MR * 32 == TD * TH, "tile shape");
  const GemmArgs& a = f.g;
  constexpr int NT = WM * WN * 64;
  constexpr int NXI = nxi_of(K), T2 = K * K, PADK = (K - 1) / 2;
  constexpr int HDt = TD + K - 1, HPt = TH + K - 1, NSLOT = HDt * HPt, NPL = 2 * NP, NIT = NPL * NSLOT, ROUNDS = (NIT + NT - 1) / NT;
  constexpr int NBF = T2 * NP;                      // BL: B fragments (64 slots of 16 B) per stage
  constexpr int BROUNDS = (NBF + WM * WN - 1) / (WM * WN);
  static_assert(BL == 0 || NIT % 64 == 0, "weights through LDS: wave-granular tile");
  static_assert(BL != 1 || WN == 1, "B through LDS, whole stage: one column fragment");
  // BL == 2 (pipelined): two tile buffers + two weight slots of one PHASE of a stage: PH phases of T2 / PH taps (K = 5: 6 6 6 7
  // taps with one column fragment, 5 x 5 taps = the kd rows with two), slot = the fragments of the largest (last) phase
  constexpr int PH = WN == 1 ? 4 : 5;
  constexpr int PQ_T[6] = {0, T2 * 1 / PH, T2 * 2 / PH, T2 * 3 / PH, PH > 4 ? T2 * 4 / PH : T2, T2};
  constexpr int PQ_FR = (T2 - PQ_T[PH - 1]) * NP * WN;
  constexpr int PQ_SLOT = PQ_FR * 64;
  static_assert(BL != 2 || (NP == 2 && WM * WN == 4 && WN <= 2 && PQ_T[1] * NP * WN >= 4), "pipelined form: fp16 x 2 pieces, four wavefronts");
  __shared__ uint4 lds[BL == 2 ? 2 * NIT + 2 * PQ_SLOT : (BL ? NIT + NBF * 64 : ROUNDS * NT)];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  // block -> tile: XCD x (= blockIdx % 8) walks the contiguous tile range [x * per_xcd, (x + 1) * per_xcd)
  int b = (blockIdx.x & 7) * f.per_xcd + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= f.per_xcd || b >= a.nblk) return;
  const int tile_id = b;
  const int thi = b % a.tiles_h;
  b /= a.tiles_h;
  const int tdi = b % a.tiles_d;
  b /= a.tiles_d;
  const int t = b % a.T;
  const int n = b / a.T;

  unsigned voff[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    int it = r * NT + tid;
    if (it >= NIT) it = 0;
    const int pk = it / NSLOT, slot = it - pk * NSLOT;
    const int row = slot / HPt, col = slot - row * HPt;
    voff[r] = (unsigned)(pk * a.v_plane + ((long)row * a.HP + col) * 16);
  }
  const char* vtile0 = a.V + ((long)(n * a.T + t) * a.KC) * NPL * a.v_plane +
                       ((long)(tdi * TD + 2 - PADK) * a.HP + thi * TH + 2 - PADK) * 16;
  const unsigned ulane = (unsigned)(lh * a.CN + wn * 32 + li) * 16u;
  const unsigned ustep = (unsigned)a.CN * 32u;
  const unsigned uchunk = (unsigned)NP * ustep;
  const unsigned utap = (unsigned)a.KC * uchunk;

  int arow[MR];
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
    int fa, fb;
    frag_pos<TH>(li, fa, fb);
    arow[mr] = lh * NSLOT + ((wm * MR + mr) * (32 / TH) + fa) * HPt + fb;
  }

  f32x16 yo[4][MR];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
      for (int j = 0; j < 16; ++j) yo[i][mr][j] = 0.f;

  if constexpr (BL == 2) {
    // ---- pipelined form (round 6): nothing the tap loop reads is waited for where it is requested ------------------------
    // A stage (point xi, 16-channel chunk kc) runs in FOUR phases of 6/6/6/7 taps.  LDS holds two tile buffers (stage s reads
    // [s & 1] while LDS-DMA fills the other with stage s + 1) and two quarter-stage weight slots (phase q reads slot q & 1 while
    // the other receives the fragments of the next phase).  Every fill is requested one whole phase (weights) or two (tile)
    // before its first read, so the only synchronisation is ONE counted s_waitcnt + ONE raw s_barrier per phase.  The fills are
    // issued by inline assembly: hipcc drains every LDS-DMA it knows about before the next ds_read (s_waitcnt vmcnt(0) right
    // behind the request -- profiles/r06_fused_probes.txt (h)); what it does not see it does not wait for, and the waits
    // here are counted by hand (loads retire in order: "vmcnt(7)" = everything but the 7 requests of the next tile).
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)lds;
    const int NS = NXI * a.KC;
    const unsigned voff_last = (ROUNDS - 1) * NT + wave * 64 < NIT ? voff[ROUNDS - 1] : voff[ROUNDS - 2];   // (uniform choice: the
    const int r_last = (ROUNDS - 1) * NT + wave * 64 < NIT ? ROUNDS - 1 : ROUNDS - 2;   // last round of three wavefronts repeats the one before: equal request counts)
    auto rsrc_of = [](const char* p_) {
      const unsigned long pa = (unsigned long)p_;
      i32x4 r_ = {(int)__builtin_amdgcn_readfirstlane((unsigned)pa), (int)(__builtin_amdgcn_readfirstlane((unsigned)(pa >> 32)) & 0xffffu),
                  (int)0xFFFFFFF0u, 0x00020000};
      return r_;
    };
    auto fill_tile = [&](int s_) {        // ROUNDS requests per wavefront
      const int xi_ = s_ / a.KC, kc_ = s_ - xi_ * a.KC;
      const i32x4 vr = rsrc_of(vtile0 + (long)xi_ * a.v_xi);
      const unsigned so = (unsigned)(kc_ * NPL * a.v_plane);
      const unsigned dst = lds0 + (unsigned)((s_ & 1) * NIT + wave * 64) * 16u;
#pragma unroll
      for (int r = 0; r < ROUNDS - 1; ++r) wbf_dma16(vr, dst + (unsigned)(r * NT) * 16u, voff[r], so);
      wbf_dma16(vr, dst + (unsigned)(r_last * NT) * 16u, voff_last, so);
    };
    auto fill_weights = [&](int s_, auto qc) {   // ceil(fragments / 4) requests per wavefront
      constexpr int q = decltype(qc)::value;
      constexpr int f0 = PQ_T[q] * NP * WN, nf = (PQ_T[q + 1] - PQ_T[q]) * NP * WN, rounds = (nf + 3) / 4;   // fragment = ((tap, piece), column fragment)
      const int xi_ = s_ / a.KC, kc_ = s_ - xi_ * a.KC;
      const i32x4 ur = rsrc_of(a.U + (long)xi_ * a.u_xi);
      const unsigned ukc = (unsigned)kc_ * uchunk;
#pragma unroll
      for (int r = 0; r < rounds; ++r) {
        int fr = r * 4 + wave;
        if (fr >= nf) fr -= 4;             // (uniform) repeat a fragment of the round before: equal request counts
        const int fg = (f0 + fr) / WN, fw = (f0 + fr) % WN;     // (tap, piece) and column fragment: lane (lh, li) -> slot lh * CN + fw * 32 + li
        wbf_dma16(ur, lds0 + (unsigned)(2 * NIT + ((q + (s_ & PH & 1)) & 1) * PQ_SLOT + fr * 64) * 16u, (unsigned)(lh * a.CN + fw * 32 + li) * 16u,
                  (unsigned)(fg / NP) * utap + ukc + (unsigned)(fg % NP) * ustep);
      }
    };
    f32x16 acc[MR];
    int abase[MR];
    auto run_phase = [&](int s_, auto qc) {
      constexpr int q = decltype(qc)::value;
      constexpr int T0 = PQ_T[q], T1 = PQ_T[q + 1];
      const bool more = s_ + 1 < NS;
      // (1) what this phase reads has landed -- in THIS wavefront's requests; the barrier extends that to all four
      if (q == 1 && more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ROUNDS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // (2) the slots the previous phase was reading are free: request the next phase's weights, then (phase 0) the next tile
      if constexpr (q < PH - 1) fill_weights(s_, std::integral_constant<int, q + 1>{});
      else if (more) fill_weights(s_ + 1, std::integral_constant<int, 0>{});
      if (q == 0 && more) fill_tile(s_ + 1);
      // (3) the taps of this phase: LDS -> MFMA only
      const uint4* bl = lds + 2 * NIT + ((q + (s_ & PH & 1)) & 1) * PQ_SLOT + wn * 64 + lane;   // slot = parity of the running phase count s * PH + q
      uint4 aq[2][MR][NP], bq[2][NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) bq[T0 & 1][p] = bl[p * WN * 64];
#pragma unroll
      for (int mr = 0; mr < MR; ++mr) {
        const uint4* ap = lds + abase[mr] + (T0 / K) * HPt + (T0 % K);
#pragma unroll
        for (int p = 0; p < NP; ++p) aq[T0 & 1][mr][p] = ap[p * 2 * NSLOT];
      }
#pragma unroll
      for (int tap = T0; tap < T1; ++tap) {
        const int cur = tap & 1, nx = cur ^ 1;
        if (tap + 1 < T1) {
#pragma unroll
          for (int p = 0; p < NP; ++p) bq[nx][p] = bl[((tap + 1 - T0) * NP + p) * WN * 64];
#pragma unroll
          for (int mr = 0; mr < MR; ++mr) {
            const uint4* ap = lds + abase[mr] + ((tap + 1) / K) * HPt + ((tap + 1) % K);
#pragma unroll
            for (int p = 0; p < NP; ++p) aq[nx][mr][p] = ap[p * 2 * NSLOT];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        const uint4 bdown = wbf_hi_down(bq[cur][0]);  // partner of the scaled low piece (msk_wbf.h)
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) WBF_MFMA_H(acc[mr], aq[cur][mr][NP - 1], bdown);
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) WBF_MFMA_H(acc[mr], aq[cur][mr][0], bq[cur][NP - 1]);
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) WBF_MFMA_H(acc[mr], aq[cur][mr][0], bq[cur][0]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    fill_weights(0, std::integral_constant<int, 0>{});
    fill_tile(0);
#pragma unroll 1
    for (int s_ = 0; s_ < NS; ++s_) {
      const int xi = s_ / a.KC, kc = s_ - xi * a.KC;
      if (kc == 0) {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
          for (int j = 0; j < 16; ++j) acc[mr][j] = 0.f;
      }
#pragma unroll
      for (int mr = 0; mr < MR; ++mr) abase[mr] = arow[mr] + (s_ & 1) * NIT;
      run_phase(s_, std::integral_constant<int, 0>{});
      run_phase(s_, std::integral_constant<int, 1>{});
      run_phase(s_, std::integral_constant<int, 2>{});
      run_phase(s_, std::integral_constant<int, 3>{});
      if constexpr (PH > 4) run_phase(s_, std::integral_constant<int, 4>{});
      if (kc == a.KC - 1) {
        float c0, c1, c2, c3;
        at_column<K>(xi, c0, c1, c2, c3);
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const float m = acc[mr][j];
            yo[0][mr][j] = fmaf(c0, m, yo[0][mr][j]);
            yo[1][mr][j] = fmaf(c1, m, yo[1][mr][j]);
            yo[2][mr][j] = fmaf(c2, m, yo[2][mr][j]);
            yo[3][mr][j] = fmaf(c3, m, yo[3][mr][j]);
          }
      }
    }
    __syncthreads();   // (nothing is in flight any more: the statistics epilogue reuses the LDS)
  } else {
#pragma unroll 1
  for (int xi = 0; xi < NXI; ++xi) {
    const __amdgpu_buffer_rsrc_t vres =
        __builtin_amdgcn_make_buffer_rsrc((void*)(vtile0 + (long)xi * a.v_xi), 0, 0xFFFFFFF0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ures =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.U + (long)xi * a.u_xi), 0, 0xFFFFFFF0u, 0x00020000);
    f32x16 acc[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[mr][j] = 0.f;

#pragma unroll 1
    for (int kc = 0; kc < a.KC; ++kc) {
      if constexpr (BL != 0) {
        __syncthreads();  // every wavefront is done reading the previous stage's tile and weights
        const unsigned vsoff = (unsigned)(kc * NPL * a.v_plane);
        const bool fill = !(WBF_PROBE & 1) || (xi == 0 && kc == 0);
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
          if (fill && r * NT + wave * 64 < NIT)   // (wave-uniform: the last round stops where the tile does, the weights follow it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vres, (__attribute__((address_space(3))) void*)(lds + r * NT + wave * 64), 16,
                                                     (int)voff[r], (int)vsoff, 0, 0);
        const unsigned ukc = (unsigned)kc * uchunk;
#pragma unroll
        for (int r = 0; r < BROUNDS; ++r) {
          const int q = r * (WM * WN) + wave;   // fragment q = tap * NP + piece: 1 KB, lane-linear in U and in LDS
          if (fill && q < NBF)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ures, (__attribute__((address_space(3))) void*)(lds + NIT + q * 64), 16,
                                                     (int)(lane * 16), (int)((unsigned)(q / NP) * utap + ukc + (unsigned)(q % NP) * ustep), 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        const uint4* bl = lds + NIT + lane;
        uint4 aq[2][MR][NP], bq[2][NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) bq[0][p] = bl[p * 64];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
          const uint4* ap = lds + arow[mr];
#pragma unroll
          for (int p = 0; p < NP; ++p) aq[0][mr][p] = ap[p * 2 * NSLOT];
        }
#pragma unroll
        for (int tap = 0; tap < T2; ++tap) {
          const int cur = tap & 1, nx = cur ^ 1;
          if (tap + 1 < T2) {
#pragma unroll
            for (int p = 0; p < NP; ++p) { if (WBF_PROBE & 2) bq[nx][p] = bq[cur][p]; else bq[nx][p] = bl[((tap + 1) * NP + p) * 64]; }
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) {
              const uint4* ap = lds + arow[mr] + ((tap + 1) / K) * HPt + ((tap + 1) % K);
#pragma unroll
              for (int p = 0; p < NP; ++p) { if (WBF_PROBE & 4) aq[nx][mr][p] = aq[cur][mr][p]; else aq[nx][mr][p] = ap[p * 2 * NSLOT]; }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (NP == 3) {
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][NP - 1], bq[cur][0]);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][0], bq[cur][NP - 1]);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][NP / 2], bq[cur][NP / 2]);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][NP / 2], bq[cur][0]);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][0], bq[cur][NP / 2]);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][0], bq[cur][0]);
          } else if (NP == 2 && (WBF_PROBE & 16)) {
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
              for (int p = 0; p < NP; ++p)
                asm volatile("" ::"v"(aq[cur][mr][p].x), "v"(aq[cur][mr][p].y), "v"(aq[cur][mr][p].z), "v"(aq[cur][mr][p].w),
                             "v"(bq[cur][p].x), "v"(bq[cur][p].y), "v"(bq[cur][p].z), "v"(bq[cur][p].w));
          } else if (NP == 2) {
            const uint4 bdown = wbf_hi_down(bq[cur][0]);  // partner of the scaled low piece (msk_wbf.h)
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) WBF_MFMA_H(acc[mr], aq[cur][mr][NP - 1], bdown);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) WBF_MFMA_H(acc[mr], aq[cur][mr][0], bq[cur][NP - 1]);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) WBF_MFMA_H(acc[mr], aq[cur][mr][0], bq[cur][0]);
          } else {
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) WBF_MFMA_H(acc[mr], aq[cur][mr][0], bq[cur][0]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
      __syncthreads();  // every wavefront is done reading the previous stage's tile
      const unsigned vsoff = (unsigned)(kc * NPL * a.v_plane);
      if (!(WBF_PROBE & 1) || (xi == 0 && kc == 0)) {
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(vres, (__attribute__((address_space(3))) void*)(lds + r * NT + wave * 64), 16,
                                                 (int)voff[r], (int)vsoff, 0, 0);
      }
      const unsigned ukc = (unsigned)kc * uchunk;
      constexpr int BR = BPF + 1;   // ring of B fragment sets: tap t lives in bq[t % BR]
      uint4 bq[BR][NP];
#pragma unroll
      for (int t0 = 0; t0 < BPF && t0 < T2; ++t0)
#pragma unroll
        for (int p = 0; p < NP; ++p) bq[t0][p] = buf_load16(ures, ulane, (unsigned)t0 * utap + ukc + p * ustep);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();

      uint4 aq[2][MR][NP];
#pragma unroll
      for (int mr = 0; mr < MR; ++mr) {
        const uint4* ap = lds + arow[mr];
#pragma unroll
        for (int p = 0; p < NP; ++p) aq[0][mr][p] = ap[p * 2 * NSLOT];
      }
#pragma unroll
      for (int tap = 0; tap < T2; ++tap) {
        const int cur = tap & 1, nx = cur ^ 1;
        const int bcur = tap % BR;
        if (tap + BPF < T2) {
          const unsigned ub = (unsigned)(tap + BPF) * utap + ukc;
#pragma unroll
          for (int p = 0; p < NP; ++p) { if (WBF_PROBE & 2) bq[(tap + BPF) % BR][p] = bq[bcur][p]; else bq[(tap + BPF) % BR][p] = buf_load16(ures, ulane, ub + p * ustep); }
        }
        if (tap + 1 < T2) {
#pragma unroll
          for (int mr = 0; mr < MR; ++mr) {
            const uint4* ap = lds + arow[mr] + ((tap + 1) / K) * HPt + ((tap + 1) % K);
#pragma unroll
            for (int p = 0; p < NP; ++p) { if (WBF_PROBE & 4) aq[nx][mr][p] = aq[cur][mr][p]; else aq[nx][mr][p] = ap[p * 2 * NSLOT]; }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (NP == 3) {
#pragma unroll
          for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][NP - 1], bq[bcur][0]);
#pragma unroll
          for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][0], bq[bcur][NP - 1]);
#pragma unroll
          for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][NP / 2], bq[bcur][NP / 2]);
#pragma unroll
          for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][NP / 2], bq[bcur][0]);
#pragma unroll
          for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][0], bq[bcur][NP / 2]);
#pragma unroll
          for (int mr = 0; mr < MR; ++mr) WBF_MFMA(acc[mr], aq[cur][mr][0], bq[bcur][0]);
        } else if (NP == 2 && (WBF_PROBE & 16)) {
#pragma unroll
          for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int p = 0; p < NP; ++p)
              asm volatile("" ::"v"(aq[cur][mr][p].x), "v"(aq[cur][mr][p].y), "v"(aq[cur][mr][p].z), "v"(aq[cur][mr][p].w),
                           "v"(bq[bcur][p].x), "v"(bq[bcur][p].y), "v"(bq[bcur][p].z), "v"(bq[bcur][p].w));
        } else if (NP == 2) {
          const uint4 bdown = wbf_hi_down(bq[bcur][0]);  // partner of the scaled low piece (msk_wbf.h)
#pragma unroll
          for (int mr = 0; mr < MR; ++mr) WBF_MFMA_H(acc[mr], aq[cur][mr][NP - 1], bdown);
#pragma unroll
          for (int mr = 0; mr < MR; ++mr) WBF_MFMA_H(acc[mr], aq[cur][mr][0], bq[bcur][NP - 1]);
#pragma unroll
          for (int mr = 0; mr < MR; ++mr) WBF_MFMA_H(acc[mr], aq[cur][mr][0], bq[bcur][0]);
        } else {
#pragma unroll
          for (int mr = 0; mr < MR; ++mr) WBF_MFMA_H(acc[mr], aq[cur][mr][0], bq[bcur][0]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      }  // !BL
    }
    // output transform, one column of A^T: wave-uniform coefficients (scalar registers)
    float c0, c1, c2, c3;
    at_column<K>(xi, c0, c1, c2, c3);
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float m = acc[mr][j];
        yo[0][mr][j] = fmaf(c0, m, yo[0][mr][j]);
        yo[1][mr][j] = fmaf(c1, m, yo[1][mr][j]);
        yo[2][mr][j] = fmaf(c2, m, yo[2][mr][j]);
        yo[3][mr][j] = fmaf(c3, m, yo[3][mr][j]);
      }
  }

  }  // BL != 2
  // epilogue: y = yo / (operand scales) + bias [+ dst] [PReLU]; a lane owns ONE output channel (column li of the MFMA
  // result) and MR x 16 positions x 4 W outputs of it
  const int co = wn * 32 + li;
  const float osc = f.scaled ? 1.f / (wbf_scale_of(f.in_amax) * wbf_scale_of(f.w_amax)) : 1.f;
  const float bv = f.bias ? f.bias[co] : 0.f;
  const float sl = f.prelu ? f.prelu[co] : 1.f;
  float* obase = f.dst + ((long)n * f.dvn + (long)(4 * t) * f.dvw) * f.dld + co;
  const long wst = (long)f.dvw * f.dld;
  // split store: the same voxel offsets at half the stride (dld == 2 csplit), per-lane base by channel half
  const bool split_st = f.st_lo != nullptr;
  float* sbase = split_st ? (co < f.csplit ? f.st_lo + co : f.st_hi + (co - f.csplit)) + (((long)n * f.dvn + (long)(4 * t) * f.dvw) * f.dld >> 1) : obase;
  const long swst = split_st ? (wst >> 1) : wst;
  const long aoff = f.acc_src ? (f.acc_src - f.dst) : 0;   // (element distance between the tensor the old values are read from and dst)
  const int wlim = f.LW - 4 * t;   // W outputs of this tile inside the volume (wave-uniform; >= 4 except in a ragged last tile)
  float sk = 0.f, s1 = 0.f, s2 = 0.f, cnt = 0.f;
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) {
      // four rows x four W outputs at a time; the accumulate path loads all 16 old values BEFORE the first store (a
      // load -> add -> store chain per element serialises 128 memory round trips: the stores may alias the next load)
      float* op[4];
      bool ok[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        int fa, fb;
        frag_pos<TH>(jj + 8 * jq + 4 * lh, fa, fb);
        const int d = tdi * TD + (wm * MR + mr) * (32 / TH) + fa, h = thi * TH + fb;
        ok[jj] = d < a.LD && h < a.LH;
        op[jj] = obase + ((long)d * f.dvd + (long)h * f.dvh) * f.dld;
      }
      float old[4][4];
      if (f.accumulate) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int i = 0; i < 4; ++i) old[jj][i] = (ok[jj] && i < wlim) ? __builtin_nontemporal_load(op[jj] + aoff + i * wst) : 0.f;
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        if (!ok[jj]) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i >= wlim) break;
          float r = fmaf(yo[i][mr][jq * 4 + jj], osc, bv);
          if (f.accumulate) r += old[jj][i];
          if (f.prelu) r = r > 0.f ? r : sl * r;
          if ((WBF_PROBE & 8) && r != 12345.678f) continue;   // probe: no stores (the value still has to be computed)
          if (split_st) sbase[((op[jj] - obase) >> 1) + i * swst] = r;     // (nontemporal stores: measured, no difference)
          else op[jj][i * wst] = r;
          if (STATS) {
            if (cnt == 0.f) sk = r;
            const float dlt = r - sk;
            s1 += dlt;
            s2 = fmaf(dlt, dlt, s2);
            cnt += 1.f;
          }
        }
      }
    }
  }
  if (STATS) {
    // (n, mean, M2) per lane -> the two lane halves -> the WM wavefronts that share this channel group -> one record
    WfRec w = {0.f, 0.f, 0.f};
    if (cnt > 0.f) {
      w.n = cnt;
      w.mean = sk + s1 / cnt;
      w.m2 = fmaxf(s2 - s1 * s1 / cnt, 0.f);
    }
    WfRec o;
    o.n = __shfl_xor(w.n, 32, 64);
    o.mean = __shfl_xor(w.mean, 32, 64);
    o.m2 = __shfl_xor(w.m2, 32, 64);
    w = lh == 0 ? wfrec_merge(w, o) : wfrec_merge(o, w);  // both halves hold (lower, upper) merged in the same order
    __syncthreads();  // the last stage's LDS reads are done
    float* sh = reinterpret_cast<float*>(lds);
    if (lh == 0) {
      float* p_ = sh + (wave * 32 + li) * 3;
      p_[0] = w.n; p_[1] = w.mean; p_[2] = w.m2;
    }
    __syncthreads();
    if (wm == 0 && lh == 0) {
      WfRec r = w;
#pragma unroll
      for (int q = 1; q < WM; ++q) {
        const float* p_ = sh + ((q * WN + wn) * 32 + li) * 3;
        WfRec e = {p_[0], p_[1], p_[2]};
        r = wfrec_merge(r, e);
      }
      float* g_ = f.stat_partial + ((long)tile_id * a.CN + co) * 3;
      g_[0] = r.n; g_[1] = r.mean; g_[2] = r.m2;
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
  const float* acc_src; // accumulate: read the old values from this tensor (geometry of dst) instead of dst; null = dst
  const float* in_amax; // NP = 2: the device scalars the input transform and the weight pack scaled by (wbf_scale_of)
  const float* w_amax;
  int scaled;           // NP = 2
  float* stat_partial;  // STATS: per-block BatchNorm records [gridDim.x][CN][3] = (n, mean, M2) of the stored values
};

// y_j = sum_xi AT[j][xi] m_xi for one scalar lane of the 8 (6) point values
template <int K>
__device__ __forceinline__ void at_apply(const float (&m)[8], float (&y)[4]) {
  const float s12 = m[1] + m[2], d12 = m[1] - m[2];
  const float s34 = m[3] + m[4], d34 = m[3] - m[4];
  if (K == 5) {
    const float s56 = m[5] + m[6], d56 = m[5] - m[6];
    y[0] = ((m[0] + s12) + s34) + s56;
    y[1] = (d12 + 2.f * d34) + 0.5f * d56;
    y[2] = (s12 + 4.f * s34) + 0.25f * s56;
    y[3] = ((d12 + 8.f * d34) + 0.125f * d56) + m[7];
  } else {
    y[0] = (m[0] + s12) + s34;
    y[1] = d12 + 2.f * d34;
    y[2] = s12 + 4.f * s34;
    y[3] = (d12 + 8.f * d34) + m[5];
  }
}

// STATS: the BatchNorm statistics of the convolution output (vnet.py:38,41 -- conv followed by BatchNorm) are taken
// here, from the values on their way to HBM: shifted sums per thread (a thread keeps its channel quad over the grid-stride
// loop because gridDim.x * 256 is a multiple of CN / 4), Chan merge inside the block, one record per block and channel;
// msk_bn_stats_merge finishes in double.  Saves the separate read of y by bn_stats_partial.
template <bool STATS, int K>
__global__ void __launch_bounds__(256)
wbf_tout_k(ToutArgs a) {
  constexpr int NXI = nxi_of(K);
  const int c4n = a.CN >> 2;
  float sk[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  float cnt = 0.f;
  const float osc = a.scaled ? 1.f / (wbf_scale_of(a.in_amax) * wbf_scale_of(a.w_amax)) : 1.f;  // power of two: exact
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
    float mx[8], my[8], mz[8], mw[8];
    {
      // the NXI loads of one split-K slab are issued together, slab after slab in a fixed order (a point's slabs one after the
      // other made every load wait for the previous one: the deep levels run with 2-16 slabs)
      float4 s[NXI];
#pragma unroll
      for (int xi = 0; xi < NXI; ++xi) s[xi] = *reinterpret_cast<const float4*>(a.M + (long)xi * a.m_xi + idx * 4);
      for (int z = 1; z < a.ksplit; ++z) {
        float4 q[NXI];
#pragma unroll
        for (int xi = 0; xi < NXI; ++xi) q[xi] = *reinterpret_cast<const float4*>(a.M + ((long)z * NXI + xi) * a.m_xi + idx * 4);
#pragma unroll
        for (int xi = 0; xi < NXI; ++xi) {
          s[xi].x += q[xi].x; s[xi].y += q[xi].y; s[xi].z += q[xi].z; s[xi].w += q[xi].w;
        }
      }
#pragma unroll
      for (int xi = 0; xi < 8; ++xi) {
        if (xi < NXI) {
          mx[xi] = s[xi < NXI ? xi : 0].x; my[xi] = s[xi < NXI ? xi : 0].y; mz[xi] = s[xi < NXI ? xi : 0].z; mw[xi] = s[xi < NXI ? xi : 0].w;
        } else {
          mx[xi] = my[xi] = mz[xi] = mw[xi] = 0.f;
        }
      }
    }
    float yx[4], yy[4], yz[4], yw[4];
    at_apply<K>(mx, yx);
    at_apply<K>(my, yy);
    at_apply<K>(mz, yz);
    at_apply<K>(mw, yw);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), sl = make_float4(1.f, 1.f, 1.f, 1.f);
    if (a.bias) bv = reinterpret_cast<const float4*>(a.bias)[c4];
    if (a.prelu) sl = reinterpret_cast<const float4*>(a.prelu)[c4];
    float* o = a.dst + ((long)n * a.dvn + (long)d * a.dvd + (long)h * a.dvh + (long)(4 * t) * a.dvw) * a.dld + c4 * 4;
    float4 old4[4];   // accumulate: the four old values are loaded before the first store (load -> add -> store per row serialised)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      old4[i] = (a.accumulate && 4 * t + i < a.LW) ? *reinterpret_cast<const float4*>((a.acc_src ? a.acc_src + (o - a.dst) : o) + (long)i * a.dvw * a.dld) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (4 * t + i < a.LW) {
        float4* op = reinterpret_cast<float4*>(o + (long)i * a.dvw * a.dld);
        float4 r = make_float4(fmaf(yx[i], osc, bv.x), fmaf(yy[i], osc, bv.y), fmaf(yz[i], osc, bv.z), fmaf(yw[i], osc, bv.w));
        if (a.accumulate) {
          const float4 e = old4[i];
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

template <int MR, int WM, int WN, int TD, int TH, int K, int NP>
void launch_gemm(msk_ctx* ctx, const char* tag, const GemmArgs& a, long nblk) {
  GemmArgs b = a;
  b.nblk = (int)nblk;
  // tiles per workgroup: keep >= ~6 workgroups per CU in the grid (tuning knob "wbf_tpb")
  int tpb = ctx->wbf_tpb > 0 ? ctx->wbf_tpb : 1;
  while (tpb > 1 && nblk / tpb < 6L * ctx->num_cu) tpb >>= 1;
  long grid = (nblk + tpb - 1) / tpb;
  grid = (grid + 7) & ~7L;
  b.tpb = (int)((nblk + grid - 1) / grid);
  // weight fragments four taps ahead for the launches that cannot hide an L2 round trip behind other workgroups (option "wbf_bpf":
  // 0 = by grid size, 1 / 4 = force)
  const bool deep = MR == 2 && (ctx->wbf_bpf == 4 || (ctx->wbf_bpf == 0 && nblk <= 8L * ctx->num_cu));
  if constexpr (MR == 2) {
    if (deep) {
      MSK_LAUNCH_TIMED(ctx, tag, (wbf_gemm_k<MR, WM, WN, TD, TH, K, NP, 4>), dim3((unsigned)grid), dim3(WM * WN * 64), 0, b);
      return;
    }
  }
  MSK_LAUNCH_TIMED(ctx, tag, (wbf_gemm_k<MR, WM, WN, TD, TH, K, NP>), dim3((unsigned)grid), dim3(WM * WN * 64), 0, b);
}

// tile variants {id, MR, WM, WN, TD, TH} by output channels (first = preferred)
struct Var { int id, MR, WM, WN, TD, TH; };
// Measured (tools/bench_conv.py, 2 x 128^3 .. 2 x 16^3): the MR = 2 variants win everywhere (32ch@128^3 3.06 vs 3.13 ms,
// 128ch@32^3 0.60 vs 0.85 ms, 256ch@16^3 0.32 vs 0.45 ms): smaller LDS tiles -> 3-4 workgroups per CU hide the
// staging barriers; the MR = 4 variants halve the B-fragment traffic and stay selectable ("wbf_variant", K = 5 only).
const Var kVars[7] = {{4, 2, 4, 1, 16, 16}, {0, 4, 4, 1, 16, 32},    // CN == 32
                      {5, 2, 2, 2, 8, 16},  {1, 4, 2, 2, 16, 16},    // CN == 64
                      {3, 2, 1, 4, 8, 8},   {2, 4, 1, 4, 8, 16},     // CN % 128 == 0
                      {6, 4, 2, 1, 16, 16}};                         // CN == 32, two wavefronts x 4 row fragments ("wbf_variant" 6)
const Var* pick_variant(const msk_ctx* ctx, const WbfGeom& geo, int CN, int K) {
  int v0;
  if (CN == 32) v0 = 0;
  else if (CN == 64) v0 = 2;
  else if (CN >= 128 && CN % 128 == 0) v0 = 4;
  else return nullptr;
  if (K == 5 && CN == 32 && ctx->wbf_variant == 6 && wbf_tile_ok(geo, 16, 16)) return &kVars[6];
  // Round 5 (tools/sweep_deep_gemm.sh): the levels whose packed weights do not stay in an XCD's L2 next to the tiles are bound by
  // the weight fragments they pull from L2, and a wavefront with FOUR row fragments uses every fragment twice as often --
  // 128ch@32^3 0.276 -> 0.258 ms, 256ch@16^3 0.148 -> 0.133, 64ch@32^3 0.074 -> 0.071, 128ch@16^3 0.042 -> 0.040 (round 2 had
  // measured the MR = 2 tiles ahead everywhere, with six products per operand pair and no split-K cap).  The MR = 4 variant goes
  // first where the one-kernel form (MR = 2 tiles only) would not take the layer anyway: >= 128 output channels (packed weights
  // > 3.5 MB), 64 channels below two tiles per CU.  Option "wbf_mr4" 0 = MR 2 first (A/B).
  if (K == 5 && ctx->wbf_variant < 0 && ctx->wbf_mr4 && CN >= 64) {
    const Var& m4 = kVars[v0 + 1];
    bool first = wbf_tile_ok(geo, m4.TD, m4.TH);
    if (first && CN == 64) {
      const Var& m2 = kVars[v0];
      const long tiles2 = (long)geo.T * ((geo.LD + m2.TD - 1) / m2.TD) * ((geo.LH + m2.TH - 1) / m2.TH);   // per sample
      // (two samples: the one-kernel form wants 2 tiles per CU in all; "wbf_fuse" 2 = tests force that form at any size)
      first = ctx->wbf_fuse == 0 || (ctx->wbf_fuse != 2 && tiles2 < (long)ctx->num_cu);
    }
    if (first) return &m4;
  }
  for (int c = v0; c < v0 + (K == 5 ? 2 : 1); ++c) {
    const Var& v = kVars[c];
    if (K == 5 && ctx->wbf_variant >= 0 && ctx->wbf_variant != v.id) continue;  // tuning knob "wbf_variant"
    if (wbf_tile_ok(geo, v.TD, v.TH)) return &v;
  }
  return nullptr;
}

template <int K, int NP>
void launch_gemm_variant(msk_ctx* ctx, const char* tag, int variant, const GemmArgs& ga, long nblk) {
  switch (variant) {
    case 3: launch_gemm<2, 1, 4, 8, 8, K, NP>(ctx, tag, ga, nblk); break;
    case 4: launch_gemm<2, 4, 1, 16, 16, K, NP>(ctx, tag, ga, nblk); break;
    case 5: launch_gemm<2, 2, 2, 8, 16, K, NP>(ctx, tag, ga, nblk); break;
    default:
      if constexpr (K == 5) {
        if (variant == 6) launch_gemm<4, 2, 1, 16, 16, 5, NP>(ctx, tag, ga, nblk);
        else if (variant == 0) launch_gemm<4, 4, 1, 16, 32, 5, NP>(ctx, tag, ga, nblk);
        else if (variant == 1) launch_gemm<4, 2, 2, 16, 16, 5, NP>(ctx, tag, ga, nblk);
        else launch_gemm<4, 1, 4, 8, 16, 5, NP>(ctx, tag, ga, nblk);
      }
      break;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Packed-weight cache (round 3).  The transformed + split weights U of a layer depend on the weights only, and those
// change once per optimizer step -- not per call: every (weights, direction, geometry) combination owns a table row with
// a persistent U buffer and the amax array of its weights.  A row is rebuilt (absmax + pack)
//   * for ALL rows in use at once, two launches, at the end of msk_sgd_momentum / msk_adam (msk_wbf_prepack): 28 pack
//     launches + 14 absmax passes + their memsets per step became 2 + 1;
//   * lazily at its next use otherwise.
// Rows are invalidated by every entry point that writes caller memory which may hold weights (msk_weights_changed:
// optimizer kernels, msk_h2d / msk_h2d_async / msk_memset / msk_d2d, msk_conv_fold_bn, msk_dp_broadcast, msk_free).
// ---------------------------------------------------------------------------------------------------------
struct WbfPackEntry {
  WbfPackDesc d{};
  int K = 0, NP = 0;
  size_t u_bytes = 0;
  bool live = false, valid = false;
  long last_use = 0;
};
struct WbfPackCache {
  static constexpr int kRows = 250;
  WbfPackEntry e[kRows];
  WbfPackDesc* table = nullptr;  // device copy of the descriptors
  float* amax = nullptr;         // device [kRows][kWbfAmaxWays]
  long epoch = 1;
};

WbfPackCache* pack_cache(msk_ctx* ctx) {
  if (ctx->wpack) return (WbfPackCache*)ctx->wpack;
  WbfPackCache* c = new WbfPackCache();
  if (hipMalloc((void**)&c->table, sizeof(WbfPackDesc) * WbfPackCache::kRows) != hipSuccess ||
      hipMalloc((void**)&c->amax, sizeof(float) * kWbfAmaxWays * WbfPackCache::kRows) != hipSuccess) {
    msk_fail(ctx, __FILE__, __LINE__, "pack_cache", "hipMalloc failed");
    delete c;
    return nullptr;
  }
  ctx->wpack = c;
  return c;
}

void pack_row_drop(msk_ctx* ctx, WbfPackEntry& e) {
  if (!e.live) return;
  hipStreamSynchronize(ctx->stream);
  if (ctx->side) hipStreamSynchronize(ctx->side);
  if (e.d.out) hipFree(e.d.out);
  e = WbfPackEntry();
}

// rebuild the listed rows: zero their amax arrays, one absmax launch, one pack launch per (K, NP) class.  A single row
// travels by value (its table slot may not be written yet / is the scratch row).
int pack_rows_build(msk_ctx* ctx, WbfPackCache* c, const std::vector<int>& rows) {
  if (rows.empty()) return 0;
  const bool one = rows.size() == 1;
  const WbfPackDesc single = c->e[rows[0]].d;
  bool need_amax = false;
  for (int r : rows) need_amax |= c->e[r].NP != 3;
  if (need_amax) {
    // zero the amax arrays of runs of consecutive rows with one memset each (in steady state: one run)
    size_t i = 0;
    while (i < rows.size()) {
      size_t j = i;
      while (j + 1 < rows.size() && rows[j + 1] == rows[j] + 1) ++j;
      MSK_CHECK_HIP(ctx, hipMemsetAsync(c->amax + (size_t)rows[i] * kWbfAmaxWays, 0, (j - i + 1) * kWbfAmaxWays * sizeof(float), ctx->stream));
      i = j + 1;
    }
    WbfPackList l{};
    for (int r : rows)
      if (c->e[r].NP != 3) l.row[l.n++] = (unsigned char)r;
    const unsigned ny = (unsigned)l.n;
    if (one) l.n = -1;
    msk_launch_scope ls(ctx, "wbf_pack_absmax");
    // (round 4: 256 workgroups per row instead of 64 -- the eight 33 MB tensors of the 256-channel layers set this launch's time)
    hipLaunchKernelGGL(wbf_pack_absmax_k, dim3(256, ny), dim3(256), 0, ctx->stream, c->table, l, single);
    MSK_LAUNCH_CHECK(ctx);
  }
  static const int kClasses[5][2] = {{5, 2}, {5, 3}, {3, 2}, {3, 3}, {3, 1}};
  for (const auto& kc : kClasses) {
    WbfPackList l{};
    long most = 0;
    for (int r : rows) {
      const WbfPackEntry& e = c->e[r];
      if (e.K != kc[0] || e.NP != kc[1]) continue;
      l.row[l.n++] = (unsigned char)r;
      const long blocks = ((long)e.K * e.K * e.d.KC * 16 * e.d.CN + 255) / 256;
      if (blocks > most) most = blocks;
    }
    if (l.n == 0) continue;
    // LDS-staged form (round 5, option "wbf_pack_lds"): every row of the class must tile into 16 x 8 (k, n) blocks of aligned runs
    bool lds_ok = ctx->wbf_pack_lds != 0;
    long most_lds = 0;
    for (int i = 0; i < l.n && lds_ok; ++i) {
      const WbfPackDesc& dd = c->e[l.row[i]].d;
      lds_ok = dd.CK == dd.KC * 16 && dd.CN % 8 == 0 && dd.B % 4 == 0 && (((uintptr_t)dd.w) & 15) == 0 &&
               (dd.swap ? dd.B >= dd.CK : dd.B >= dd.CN);
      const long blocks = (long)dd.KC * (dd.CN / 8);
      if (blocks > most_lds) most_lds = blocks;
    }
    if (lds_ok) {
      if (most_lds > 4L * ctx->num_cu) most_lds = 4L * ctx->num_cu;
      const dim3 grid((unsigned)most_lds, (unsigned)l.n);
      if (one) l.n = -1;
      const size_t lds = (size_t)128 * kc[0] * kc[0] * kc[0] * sizeof(float);
      msk_launch_scope ls(ctx, "wbf_pack_weights");
      if (kc[0] == 5 && kc[1] == 2) hipLaunchKernelGGL((wbf_pack_weights_lds_k<5, 2>), grid, dim3(256), lds, ctx->stream, c->table, l, single);
      else if (kc[0] == 5) hipLaunchKernelGGL((wbf_pack_weights_lds_k<5, 3>), grid, dim3(256), lds, ctx->stream, c->table, l, single);
      else if (kc[1] == 2) hipLaunchKernelGGL((wbf_pack_weights_lds_k<3, 2>), grid, dim3(256), lds, ctx->stream, c->table, l, single);
      else if (kc[1] == 3) hipLaunchKernelGGL((wbf_pack_weights_lds_k<3, 3>), grid, dim3(256), lds, ctx->stream, c->table, l, single);
      else hipLaunchKernelGGL((wbf_pack_weights_lds_k<3, 1>), grid, dim3(256), lds, ctx->stream, c->table, l, single);
      MSK_LAUNCH_CHECK(ctx);
      continue;
    }
    if (most > 8L * ctx->num_cu) most = 8L * ctx->num_cu;
    const dim3 grid((unsigned)most, (unsigned)l.n);
    if (one) l.n = -1;
    msk_launch_scope ls(ctx, "wbf_pack_weights");
    if (kc[0] == 5 && kc[1] == 2) hipLaunchKernelGGL((wbf_pack_weights_k<5, 2>), grid, dim3(256), 0, ctx->stream, c->table, l, single);
    else if (kc[0] == 5) hipLaunchKernelGGL((wbf_pack_weights_k<5, 3>), grid, dim3(256), 0, ctx->stream, c->table, l, single);
    else if (kc[1] == 2) hipLaunchKernelGGL((wbf_pack_weights_k<3, 2>), grid, dim3(256), 0, ctx->stream, c->table, l, single);
    else if (kc[1] == 3) hipLaunchKernelGGL((wbf_pack_weights_k<3, 3>), grid, dim3(256), 0, ctx->stream, c->table, l, single);
    else hipLaunchKernelGGL((wbf_pack_weights_k<3, 1>), grid, dim3(256), 0, ctx->stream, c->table, l, single);
    MSK_LAUNCH_CHECK(ctx);
  }
  for (int r : rows) c->e[r].valid = true;
  return 0;
}

// the row of (weights, direction, geometry), created on first use; < 0 on error
int pack_row_lookup(msk_ctx* ctx, WbfPackCache* c, const WbfPackDesc& key, int K, int NP, size_t u_bytes) {
  int free_row = -1, lru = -1;
  for (int r = 0; r < WbfPackCache::kRows - 1; ++r) {  // (the last row is the scratch row of uncached calls)
    WbfPackEntry& e = c->e[r];
    if (!e.live) {
      if (free_row < 0) free_row = r;
      continue;
    }
    const WbfPackDesc& d = e.d;
    if (d.w == key.w && e.K == K && e.NP == NP && d.A == key.A && d.B == key.B && d.swap == key.swap && d.flip == key.flip &&
        d.CK == key.CK && d.CN == key.CN && d.KC == key.KC && d.tsd == key.tsd && d.tsh == key.tsh && d.tsw == key.tsw) {
      e.last_use = c->epoch;
      return r;
    }
    if (lru < 0 || e.last_use < c->e[lru].last_use) lru = r;
  }
  if (free_row < 0) {
    pack_row_drop(ctx, c->e[lru]);
    free_row = lru;
  }
  WbfPackEntry& e = c->e[free_row];
  e.d = key;
  e.K = K; e.NP = NP; e.u_bytes = u_bytes;
  if (hipMalloc((void**)&e.d.out, u_bytes) != hipSuccess) {
    msk_fail(ctx, __FILE__, __LINE__, "pack_row_lookup", "hipMalloc failed");
    return -1;
  }
  e.d.amax = c->amax + (size_t)free_row * kWbfAmaxWays;
  e.live = true; e.valid = false; e.last_use = c->epoch;
  hipLaunchKernelGGL(wbf_pack_desc_store_k, dim3(1), dim3(1), 0, ctx->stream, c->table + free_row, e.d);  // no pageable-memory copy
  return free_row;
}

template <int MR, int WM, int WN, int TD, int TH, int K, int NP>
void launch_fused(msk_ctx* ctx, const char* tag, const FusedArgs& fa, bool stats, bool fork) {
  const dim3 grid((unsigned)(8 * fa.per_xcd));
  constexpr int NITc = 2 * NP * (TD + K - 1) * (TH + K - 1);
  if constexpr (WN == 2 && WM == 2 && NP == 2 && NITc % 64 == 0) {
    if (ctx->wbf_fused_bl == 2 && fa.g.CN == 64) {   // pipelined fills, two column fragments (round 6)
      if (stats) MSK_LAUNCH_TIMED_F(ctx, tag, fork, (wbf_gemm_fused_k<MR, WM, WN, TD, TH, K, NP, true, 1, 2>), grid, dim3(WM * WN * 64), 0, fa);
      else MSK_LAUNCH_TIMED_F(ctx, tag, fork, (wbf_gemm_fused_k<MR, WM, WN, TD, TH, K, NP, false, 1, 2>), grid, dim3(WM * WN * 64), 0, fa);
      return;
    }
  }
  if constexpr (WN == 1 && NP == 2 && NITc % 64 == 0) {
    // one column fragment (32 output channels): the stage's weights through LDS as well ("wbf_fused_bl": 2 = pipelined (default),
    // 1 = whole stage behind one wait, 0 = per-wavefront L1 loads)
    if (ctx->wbf_fused_bl == 2 && fa.g.CN == 32) {   // pipelined fills (round 6)
      if (stats) MSK_LAUNCH_TIMED_F(ctx, tag, fork, (wbf_gemm_fused_k<MR, WM, WN, TD, TH, K, NP, true, 1, 2>), grid, dim3(WM * WN * 64), 0, fa);
      else MSK_LAUNCH_TIMED_F(ctx, tag, fork, (wbf_gemm_fused_k<MR, WM, WN, TD, TH, K, NP, false, 1, 2>), grid, dim3(WM * WN * 64), 0, fa);
      return;
    }
    if (ctx->wbf_fused_bl && fa.g.CN == 32) {
      if (stats) MSK_LAUNCH_TIMED_F(ctx, tag, fork, (wbf_gemm_fused_k<MR, WM, WN, TD, TH, K, NP, true, 1, 1>), grid, dim3(WM * WN * 64), 0, fa);
      else MSK_LAUNCH_TIMED_F(ctx, tag, fork, (wbf_gemm_fused_k<MR, WM, WN, TD, TH, K, NP, false, 1, 1>), grid, dim3(WM * WN * 64), 0, fa);
      return;
    }
  }
  if (stats) MSK_LAUNCH_TIMED_F(ctx, tag, fork, (wbf_gemm_fused_k<MR, WM, WN, TD, TH, K, NP, true>), grid, dim3(WM * WN * 64), 0, fa);
  else MSK_LAUNCH_TIMED_F(ctx, tag, fork, (wbf_gemm_fused_k<MR, WM, WN, TD, TH, K, NP, false>), grid, dim3(WM * WN * 64), 0, fa);
}
template <int K, int NP>
void launch_fused_variant(msk_ctx* ctx, const char* tag, int variant, const FusedArgs& fa, bool stats, bool fork) {
  switch (variant) {
    case 3: launch_fused<2, 1, 4, 8, 8, K, NP>(ctx, tag, fa, stats, fork); break;
    case 4: launch_fused<2, 4, 1, 16, 16, K, NP>(ctx, tag, fa, stats, fork); break;
    default: launch_fused<2, 2, 2, 8, 16, K, NP>(ctx, tag, fa, stats, fork); break;
  }
}

template <int K, int NP>
int run_pipeline(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap, const WbfGeom& geo, const Var* bv, bool dry) {
  constexpr int NXI = nxi_of(K), NPL = 2 * NP;
  const int TD = bv->TD, TH = bv->TH, variant = bv->id;
  const int* pm = geo.perm;
  const int LD = geo.LD, LH = geo.LH, LW = geo.LW;
  const int vstr[3] = {g.DH * g.DW, g.DW, 1};
  const int tstr[3] = {K * K, K, 1};
  const int T = geo.T, KC = g.CK / 16;
  const int tiles_d = (LD + TD - 1) / TD, tiles_h = (LH + TH - 1) / TH;
  const int DP = geo.DP, HP = geo.HP;
  const int WN = bv->WN;
  const int ngrp = g.CN / (WN * 32);

  // split K (16-channel chunks) when the tiling alone cannot fill the chip
  const long base_blocks = (long)NXI * ngrp * tiles_h * tiles_d * T * g.N;
  int ksplit = 1, kc_per = KC;
  if (base_blocks < 3L * ctx->num_cu && ctx->wbf_fuse != 2) {  // ("wbf_fuse" 2: tests force the one-kernel form, which has no split-K)
    long want = ((long)ctx->wbf_ks_blocks * ctx->num_cu + base_blocks - 1) / base_blocks;
    if (want > KC) want = KC;
    kc_per = (int)((KC + want - 1) / want);
    ksplit = (KC + kc_per - 1) / kc_per;
  }
  const long nblk = base_blocks * ksplit;
  if (nblk > 0x7fffffffL) return 0;

  const size_t v_plane = (size_t)DP * HP * 16;
  const size_t v_xi = (size_t)g.N * T * KC * NPL * v_plane;
  const size_t m_xi = (size_t)g.N * T * LD * LH * g.CN;  // floats
  if (v_xi >= 0xFFFFFFF0ull) return 0;                     // 32-bit offsets inside one xi plane
  const size_t u_xi = (size_t)K * K * KC * NPL * g.CN * 16;
  if (u_xi >= 0xFFFFFFF0ull) return 0;
  if (dry) return 1;  // every eligibility test passed; nothing launched
  // stages 2 + 3 in one kernel (wbf_gemm_fused_k): one workgroup per tile covers every output channel and every point
  // (a workgroup does the work of NXI unfused ones: below two tiles per CU the chip is not filled -- 64ch@32^3, 128 tiles:
  // 0.169 ms fused against 0.072 + 0.014 ms)
  const bool fuse_out = ctx->wbf_fuse != 0 && ksplit == 1 && ngrp == 1 && (variant == 3 || variant == 4 || variant == 5) &&
                        NXI * u_xi <= ((size_t)3584 << 10) && (base_blocks / NXI >= 2L * ctx->num_cu || ctx->wbf_fuse == 2);
  const size_t v_bytes = (NXI * v_xi + 255) & ~(size_t)255;
  const size_t m_bytes = fuse_out ? 0 : (((size_t)ksplit * NXI * m_xi * sizeof(float) + 255) & ~(size_t)255);
  long tout_blocks = ((long)m_xi / 4 + 255) / 256;
  if (tout_blocks > 16L * ctx->num_cu) tout_blocks = 16L * ctx->num_cu;
  const int c4n = g.CN / 4;
  // per-sample statistics (InstanceNorm): only the one-kernel form, whose records are per TILE and the tiles sample-major
  const bool fuse_stats = g.stats != nullptr && !g.accumulate && !g.prelu && (fuse_out || (!g.stats_ps && c4n <= 256 && (c4n & (c4n - 1)) == 0));
  const long stat_rows = fuse_out ? base_blocks / NXI : tout_blocks;
  const size_t s_bytes = fuse_stats ? (size_t)stat_rows * g.CN * 3 * sizeof(float) : 0;
  char* wsp = (char*)msk_workspace(ctx, (g.xform ? 0 : v_bytes) + m_bytes + s_bytes + 256);
  if (!wsp) return -1;
  char* V = g.xform ? (char*)g.xform + kWbfXformHeader : wsp;
  float* M = (float*)(g.xform ? wsp : wsp + v_bytes);
  float* SP = (float*)((char*)M + m_bytes);
  // weights: the packed form U and the amax array the pack scaled by come from the cache row of (weights, direction,
  // geometry) when the weights are the caller's own (g.w_persistent), else they are packed into scratch here
  char* U = nullptr;
  const float* w_amax = nullptr;
  {
    WbfPackDesc key{};
    key.w = w_canon; key.A = A; key.B = B; key.swap = swap; key.flip = g.transposed ? 1 : 0; key.CK = g.CK; key.CN = g.CN; key.KC = KC;
    key.tsd = tstr[pm[0]]; key.tsh = tstr[pm[1]]; key.tsw = tstr[pm[2]];
    key.xi_stride = (long)(u_xi / 2);
    key.count = (long)K * K * K * A * B;
    WbfPackCache* c = pack_cache(ctx);
    if (!c) return -1;
    const bool cached = g.w_persistent && ctx->wbf_pack_cache != 0;
    int row;
    if (cached) {
      row = pack_row_lookup(ctx, c, key, K, NP, NXI * u_xi);
      if (row < 0) return -1;
    } else {
      // scratch row (the last one is reserved for it): rebuilt on every call, into the second workspace
      row = WbfPackCache::kRows - 1;
      WbfPackEntry& e = c->e[row];
      key.out = (unsigned short*)msk_workspace2(ctx, NXI * u_xi);
      if (!key.out) return -1;
      key.amax = c->amax + (size_t)row * kWbfAmaxWays;
      e.d = key; e.K = K; e.NP = NP; e.valid = false; e.live = false;
    }
    if (!c->e[row].valid) {
      if (pack_rows_build(ctx, c, std::vector<int>{row}) != 0) return -1;
      if (!cached) c->e[row].valid = false;
    }
    U = (char*)c->e[row].d.out;
    w_amax = c->e[row].d.amax;
  }
  // NP = 2: the source tensor is scaled into fp16 range by a power of two derived on the device from (a bound of) its
  // maximum (kept in the xform header for the weight gradient)
  const float* in_amax = nullptr;
  if (NP != 3) {
    if (g.fuse) in_amax = g.fuse->amax;
    else if (g.in_amax) in_amax = g.in_amax;
    else in_amax = msk_absmax(ctx, g.src, g.sld, g.ck_real > 0 ? g.ck_real : g.CK, (long)g.N * g.SD * g.SH * g.SW, g.xform ? (float*)g.xform : nullptr);
    if (!in_amax) return -1;
  }
  // the kept transform's header carries the maximum it was scaled by (the weight gradient undoes it): copied by the input
  // transform kernel when it arrived in another array
  float* hdr_copy = (NP != 3 && g.xform && !g.fuse && in_amax != (const float*)g.xform) ? (float*)g.xform : nullptr;

  {
    WbfTinArgs ta{};
    ta.src = g.src; ta.sld = g.sld;
    ta.svn = (long)g.DD * g.DH * g.DW; ta.svd = vstr[pm[0]]; ta.svh = vstr[pm[1]]; ta.svw = vstr[pm[2]];
    ta.N = g.N; ta.LD = LD; ta.LH = LH; ta.LW = LW; ta.T = T; ta.CK = g.CK; ta.KC = KC;
    ta.DP = DP; ta.HP = HP; ta.V = V; ta.v_xi = (long)v_xi;
    ta.amax = in_amax;
    ta.amax_copy = hdr_copy;
    ta.c_real = g.ck_real;
    if (g.fuse) {
      if (msk_wbf_transform_dual(ctx, K, NP, ta, *g.fuse, true) != 0) return -1;
      // one-kernel form: the weight gradient (side stream) may start as soon as both transforms are written: fork here,
      // not after the GEMM
      if (g.fuse->Y && ctx->wgrad_async && ctx->side != nullptr && ctx->wgrad_fork == 0) {
        hipEventRecord(ctx->ev_fork, ctx->stream);
        ctx->fork_recorded = true;
      }
    } else if (msk_wbf_transform(ctx, 0, K, NP, ta) != 0) {
      return -1;
    }
  }
  // late fork (option "wgrad_fork" 1): the weight gradient of this layer starts behind the LAST data-gradient launch -- that launch's
  // own completion is the fork point (no marker packet behind it)
  const bool fork_after = g.fuse && g.fuse->Y && ctx->wgrad_async && ctx->side != nullptr && ctx->wgrad_fork == 1 && ctx->fork_attach != 0;
  GemmArgs ga{};
  ga.V = V; ga.U = U; ga.M = M;
  ga.N = g.N; ga.T = T; ga.KC = KC; ga.CN = g.CN; ga.LD = LD; ga.LH = LH; ga.DP = DP; ga.HP = HP;
  ga.tiles_d = tiles_d; ga.tiles_h = tiles_h; ga.ngrp = ngrp; ga.ksplit = ksplit; ga.kc_per = kc_per;
  ga.v_xi = (long)v_xi; ga.v_plane = (long)v_plane; ga.u_xi = (long)u_xi; ga.m_xi = (long)m_xi;
  const char* tag = NP == 3 ? "wbf_gemm_k" : (NP == 2 ? "wbf_gemm_h2_k" : "wbf_gemm_f16_k");
  if (ctx->prof && ctx->prof_shapes) {
    char buf[200];
    snprintf(buf, sizeof(buf), "%s[ck=%d,cn=%d,n=%d,dhw=%dx%dx%d,k=%d,ks=%d%s]", tag, g.CK, g.CN, g.N, g.DD, g.DH, g.DW, K, ksplit,
             fuse_out ? ",fused" : "");
    tag = msk_intern_tag(ctx, buf);
  }
  if (fuse_out) {
    FusedArgs fa{};
    fa.g = ga;
    fa.g.nblk = (int)(base_blocks / NXI);
    fa.dst = g.dst; fa.dld = g.dld;
    fa.dvn = (long)g.DD * g.DH * g.DW; fa.dvd = vstr[pm[0]]; fa.dvh = vstr[pm[1]]; fa.dvw = vstr[pm[2]];
    fa.bias = g.bias; fa.prelu = g.prelu; fa.accumulate = g.accumulate;
    fa.acc_src = g.accumulate ? g.acc_src : nullptr;
    fa.in_amax = in_amax; fa.w_amax = w_amax; fa.scaled = NP != 3 ? 1 : 0;
    fa.stat_partial = SP;
    fa.per_xcd = (fa.g.nblk + 7) / 8;
    fa.LW = LW;
    if (g.dst_lo && g.dst_hi && g.dst_csplit > 0 && g.accumulate && g.dld == 2 * g.dst_csplit && g.CN == 2 * g.dst_csplit && !fuse_stats) {
      fa.st_lo = g.dst_lo; fa.st_hi = g.dst_hi; fa.csplit = g.dst_csplit;
      ctx->dst_split_done = true;
    }
    launch_fused_variant<K, NP>(ctx, tag, variant, fa, fuse_stats, fork_after);   // (the profile's events ride on the dispatch: MSK_LAUNCH_TIMED)
    MSK_LAUNCH_CHECK(ctx);
    if (fuse_stats) {
      if (g.stats_ps) {
        const int tps = fa.g.nblk / g.N;   // tiles per sample: tile id = ((n T + t) tiles_d + td) tiles_h + th
        for (int n = 0; n < g.N; ++n) {
          msk_bn_fin fn{};
          if (g.fin) {
            fn = *g.fin;
            const long o = (long)n * g.fin_stride;
            fn.save_mean += o; fn.save_invstd += o; fn.scale += o; fn.shift += o;
          }
          if (msk_bn_stats_merge(ctx, SP + (size_t)n * tps * g.CN * 3, tps, g.CN, g.stats + (size_t)n * 2 * g.CN, g.fin ? &fn : nullptr) != 0) return -1;
        }
      } else if (msk_bn_stats_merge(ctx, SP, fa.g.nblk, g.CN, g.stats, g.fin) != 0) {
        return -1;
      }
      ctx->stats_fused = true;
    }
    if (g.xform) ctx->xform_written = true;
    return 1;
  }
  launch_gemm_variant<K, NP>(ctx, tag, variant, ga, nblk);
  MSK_LAUNCH_CHECK(ctx);
  {
    ToutArgs oa{};
    oa.M = M; oa.m_xi = (long)m_xi; oa.ksplit = ksplit;
    oa.N = g.N; oa.T = T; oa.LD = LD; oa.LH = LH; oa.LW = LW; oa.CN = g.CN;
    oa.dst = g.dst; oa.dld = g.dld;
    oa.dvn = (long)g.DD * g.DH * g.DW; oa.dvd = vstr[pm[0]]; oa.dvh = vstr[pm[1]]; oa.dvw = vstr[pm[2]];
    oa.bias = g.bias; oa.prelu = g.prelu; oa.accumulate = g.accumulate;
    oa.acc_src = g.accumulate ? g.acc_src : nullptr;
    oa.stat_partial = SP;
    oa.in_amax = in_amax;
    oa.w_amax = w_amax;
    oa.scaled = NP != 3 ? 1 : 0;
    if (fuse_stats) MSK_LAUNCH_TIMED_F(ctx, "wbf_tout_k", false, (wbf_tout_k<true, K>), dim3((unsigned)tout_blocks), dim3(256), 0, oa);
    else MSK_LAUNCH_TIMED_F(ctx, "wbf_tout_k", fork_after, (wbf_tout_k<false, K>), dim3((unsigned)tout_blocks), dim3(256), 0, oa);
    MSK_LAUNCH_CHECK(ctx);
    if (fuse_stats) {
      if (msk_bn_stats_merge(ctx, SP, (int)tout_blocks, g.CN, g.stats, g.fin) != 0) return -1;
      ctx->stats_fused = true;
    }
  }
  if (g.xform) ctx->xform_written = true;
  return 1;
}

}  // namespace

// W tiles per workgroup of the transform kernels: all T of them (one sliding window, every source voxel read once) unless the
// launch then has fewer workgroups than option "wbf_tin_groups" (default 8 per CU): the tiles are cut into chunks, each chunk
// re-reading the K - 1 window positions it shares with its neighbour (chunks of >= 2 tiles: <= (K-1)/8 more source reads).
#ifndef WBF_TIN_MIN_PER
#define WBF_TIN_MIN_PER 2   // smallest chunk (tiles); 1 = A/B: one-tile chunks read their sources twice and lose (transforms 3.0 -> 3.2 ms)
#endif
static int wbf_tiles_per_group(const msk_ctx* ctx, long groups, int T) {
  const long want = ctx->wbf_tin_groups >= 0 ? ctx->wbf_tin_groups : 8L * ctx->num_cu;
  if (groups >= want || T <= WBF_TIN_MIN_PER) return T;
  long chunks = (want + groups - 1) / groups;
  int per = (int)((T + chunks - 1) / chunks);
  if (per < WBF_TIN_MIN_PER) per = WBF_TIN_MIN_PER;
  return per;
}

int msk_wbf_transform(msk_ctx* ctx, int mode, int K, int NP, const WbfTinArgs& ta_in) {
  WbfTinArgs ta = ta_in;
  ta.lane_map = ctx->wbf_tin_map;
  const int pblocks = (ta.DP * ta.HP + 63) / 64;
  ta.t_per = wbf_tiles_per_group(ctx, (long)pblocks * (ta.CK / 32) * ta.N, ta.T);
  const dim3 grid((unsigned)(pblocks * (ta.CK / 32)), ta.N, (unsigned)((ta.T + ta.t_per - 1) / ta.t_per));
  const char* tag = mode == 0 ? "wbf_tin_k" : "wbf_ty_k";
  if (ctx->prof && ctx->prof_shapes) {
    char buf[160];
    snprintf(buf, sizeof(buf), "%s[c=%d,n=%d,l=%dx%dx%d,k=%d,np=%d]", tag, ta.CK, ta.N, ta.LD, ta.LH, ta.LW, K, NP);
    tag = msk_intern_tag(ctx, buf);
  }
  msk_launch_scope ls(ctx, tag);
#define WBF_TIN_LAUNCH(M_, K_, P_) hipLaunchKernelGGL((wbf_tin_k<M_, K_, P_>), grid, dim3(256), 0, ctx->stream, ta)
  if (K == 5 && NP == 2) {
    if (mode == 0) WBF_TIN_LAUNCH(0, 5, 2);
    else WBF_TIN_LAUNCH(1, 5, 2);
  } else if (K == 5) {
    if (mode == 0) WBF_TIN_LAUNCH(0, 5, 3);
    else WBF_TIN_LAUNCH(1, 5, 3);
  } else if (NP == 2) {
    if (mode == 0) WBF_TIN_LAUNCH(0, 3, 2);
    else WBF_TIN_LAUNCH(1, 3, 2);
  } else if (NP == 3) {
    if (mode == 0) WBF_TIN_LAUNCH(0, 3, 3);
    else WBF_TIN_LAUNCH(1, 3, 3);
  } else {
    if (mode == 0) WBF_TIN_LAUNCH(0, 3, 1);
    else WBF_TIN_LAUNCH(1, 3, 1);
  }
#undef WBF_TIN_LAUNCH
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

int msk_wbf_transform_dual(msk_ctx* ctx, int K, int NP, const WbfTinArgs& ta_in, const WbfBnBwd& bn, bool write_v) {
  DualArgs da{};
  da.t = ta_in;
  da.t.lane_map = 1;
  da.y = bn.y; da.yld = bn.yld; da.dout = bn.dout; da.dld = bn.dld;
  da.scale = bn.scale; da.shift = bn.shift; da.alpha = bn.alpha; da.mean = bn.mean; da.invstd = bn.invstd; da.sums = bn.sums;
  da.invM = bn.invM; da.C = ta_in.CK; da.Y = bn.Y; da.y_xi = bn.y_xi; da.amax = bn.amax; da.maxes = bn.maxes;
  da.y_cmax = bn.Y ? bn.y_cmax : nullptr;
  da.coef_stride = bn.coef_stride; da.sums_stride = bn.sums_stride;
  da.bound_shift = ctx->dy_bound_shift;
  const bool write_y = bn.Y != nullptr;
  if (!write_v && !write_y) return 0;
  const int pblocks = (da.t.DP * da.t.HP + 63) / 64;
  da.t.t_per = wbf_tiles_per_group(ctx, (long)pblocks * (da.t.CK / 32) * da.t.N, da.t.T);
  const dim3 grid((unsigned)(pblocks * (da.t.CK / 32)), da.t.N, (unsigned)((da.t.T + da.t.t_per - 1) / da.t.t_per));
  const char* tag = write_v ? (write_y ? "wbf_tin_dual_k" : "wbf_tin_bn_k") : "wbf_ty_bn_k";
  if (ctx->prof && ctx->prof_shapes) {
    char buf[160];
    snprintf(buf, sizeof(buf), "%s[c=%d,n=%d,l=%dx%dx%d,k=%d,np=%d]", tag, da.t.CK, da.t.N, da.t.LD, da.t.LH, da.t.LW, K, NP);
    tag = msk_intern_tag(ctx, buf);
  }
  msk_launch_scope ls(ctx, tag);
#define WBF_DUAL_LAUNCH(K_, P_)                                                                                        \
  do {                                                                                                                 \
    if (write_v && write_y) hipLaunchKernelGGL((wbf_tin_dual_k<K_, P_, true, true>), grid, dim3(256), 0, ctx->stream, da);   \
    else if (write_v) hipLaunchKernelGGL((wbf_tin_dual_k<K_, P_, true, false>), grid, dim3(256), 0, ctx->stream, da);        \
    else hipLaunchKernelGGL((wbf_tin_dual_k<K_, P_, false, true>), grid, dim3(256), 0, ctx->stream, da);                     \
  } while (0)
  if (K == 5 && NP == 2) WBF_DUAL_LAUNCH(5, 2);
  else if (K == 5) WBF_DUAL_LAUNCH(5, 3);
  else if (NP == 2) WBF_DUAL_LAUNCH(3, 2);
  else if (NP == 3) WBF_DUAL_LAUNCH(3, 3);
  else WBF_DUAL_LAUNCH(3, 1);
#undef WBF_DUAL_LAUNCH
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

// Bytes of the transformed input V of a 'same' K^3 convolution over a [n, d, h, w, c] tensor in the shared geometry, or 0
// when the tensor is not eligible.
size_t msk_wbf_xform_bytes(int n, int d, int h, int w, int c, int cout, int K, int NP) {
  WbfGeom geo;
  int mtd, mth;
  wbf_min_tile(cout, &mtd, &mth);
  if (c < 32 || c % 32 || !wbf_pick_geom(d, h, w, mtd, mth, &geo)) return 0;
  const size_t v_xi = (size_t)n * geo.T * (c / 16) * 2 * NP * geo.DP * geo.HP * 16;
  if (v_xi >= 0xFFFFFFF0ull) return 0;
  return (size_t)nxi_of(K) * v_xi + kWbfXformHeader;
}
// the same, 0 unless msk_gconv_wino_bf3 will run the forward convolution c -> cout of that tensor
size_t msk_wbf_fwd_xform_bytes(const msk_ctx* ctx, int n, int d, int h, int w, int c, int cout, int K) {
  WbfGeom geo;
  int mtd, mth;
  wbf_min_tile(cout, &mtd, &mth);
  if (!wbf_pick_geom(d, h, w, mtd, mth, &geo) || !pick_variant(ctx, geo, cout, K)) return 0;
  return msk_wbf_xform_bytes(n, d, h, w, c, cout, K, wbf_pieces(ctx, K));
}

// Returns 1 if handled, 0 if the problem is not eligible, < 0 on error.
static int wino_bf3_impl(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap, bool dry) {
  const bool k5 = g.kd == 5 && g.kh == 5 && g.kw == 5 && g.pd == 2 && g.ph == 2 && g.pw == 2;
  const bool k3 = g.kd == 3 && g.kh == 3 && g.kw == 3 && g.pd == 1 && g.ph == 1 && g.pw == 1;
  if (!k5 && !k3) return 0;
  if (!(g.sd == 1 && g.sh == 1 && g.sw == 1)) return 0;
  if (!(g.SD == g.DD && g.SH == g.DH && g.SW == g.DW)) return 0;
  if (g.CK < 32 || g.CK % 32 || g.CN < 32 || g.CN % 32) return 0;
  if (g.sld % 4 || g.dld % 4 || (((uintptr_t)g.src) & 15) || (((uintptr_t)g.dst) & 15)) return 0;
  if (g.bias && (((uintptr_t)g.bias) & 15)) return 0;
  if (g.prelu && (((uintptr_t)g.prelu) & 15)) return 0;
  const int K = k5 ? 5 : 3;

  // logical axes and plane dims (shared with the weight gradient), then the first tile variant of the class that fits
  WbfGeom geo;
  int mtd, mth;
  wbf_min_tile(g.CN, &mtd, &mth);
  if (!wbf_pick_geom(g.DD, g.DH, g.DW, mtd, mth, &geo)) return 0;
  const Var* bv = pick_variant(ctx, geo, g.CN, K);
  if (!bv) return 0;
  const int np = wbf_pieces(ctx, K);
  if (K == 5) return np == 2 ? run_pipeline<5, 2>(ctx, g, w_canon, A, B, swap, geo, bv, dry) : run_pipeline<5, 3>(ctx, g, w_canon, A, B, swap, geo, bv, dry);
  if (np == 3) return run_pipeline<3, 3>(ctx, g, w_canon, A, B, swap, geo, bv, dry);
  if (np == 2) return run_pipeline<3, 2>(ctx, g, w_canon, A, B, swap, geo, bv, dry);
  return run_pipeline<3, 1>(ctx, g, w_canon, A, B, swap, geo, bv, dry);
}
int msk_gconv_wino_bf3(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap) {
  return wino_bf3_impl(ctx, g, w_canon, A, B, swap, false);
}
// would msk_gconv_wino_bf3 run this problem?  (launches nothing)
bool msk_gconv_wino_bf3_accepts(msk_ctx* ctx, const GConv& g) { return wino_bf3_impl(ctx, g, nullptr, 0, 0, 0, true) == 1; }

// ---- packed-weight cache: invalidation and the batched rebuild (see WbfPackCache above)
void msk_weights_changed_impl(msk_ctx* ctx, const void* p, size_t bytes) {
  msk_small_pack_changed(ctx, p, bytes);
  if (!ctx->wpack || !p) return;
  WbfPackCache* c = (WbfPackCache*)ctx->wpack;
  const char* a0 = (const char*)p;
  const char* a1 = a0 + bytes;
  for (int r = 0; r < WbfPackCache::kRows - 1; ++r) {
    WbfPackEntry& e = c->e[r];
    if (!e.live) continue;
    const char* b0 = (const char*)e.d.w;
    const char* b1 = b0 + (size_t)e.d.count * sizeof(float);
    if (a0 < b1 && b0 < a1) e.valid = false;
  }
}
void msk_weights_freed_impl(msk_ctx* ctx, const void* p, size_t bytes) {
  // msk_free: every row whose weights lie (even partly) inside the freed allocation [p, p + bytes) is DROPPED -- a row that
  // was only invalidated would be rebuilt by the next optimizer call (msk_wbf_prepack_impl) from freed memory, and its
  // packed buffer would stay allocated until the LRU evicts it.  bytes == 0 (range unknown): the rows that begin at p.
  msk_small_pack_freed(ctx, p, bytes);
  if (!ctx->wpack || !p) return;
  WbfPackCache* c = (WbfPackCache*)ctx->wpack;
  const char* a0 = (const char*)p;
  const char* a1 = a0 + bytes;
  for (int r = 0; r < WbfPackCache::kRows - 1; ++r) {
    WbfPackEntry& e = c->e[r];
    if (!e.live) continue;
    const char* b0 = (const char*)e.d.w;
    const char* b1 = b0 + (size_t)e.d.count * sizeof(float);
    if (bytes ? (a0 < b1 && b0 < a1) : b0 == a0) pack_row_drop(ctx, e);
  }
}
int msk_wbf_prepack_impl(msk_ctx* ctx) {
  if (ctx->wbf_prepack && msk_small_prepack(ctx, nullptr, 0) != 0) return -1;
  if (!ctx->wpack || !ctx->wbf_prepack) return 0;
  WbfPackCache* c = (WbfPackCache*)ctx->wpack;
  std::vector<int> rows;
  for (int r = 0; r < WbfPackCache::kRows - 1; ++r) {
    const WbfPackEntry& e = c->e[r];
    if (e.live && !e.valid && e.last_use >= c->epoch - 1) rows.push_back(r);  // rows used since the previous rebuild
  }
  c->epoch += 1;
  return pack_rows_build(ctx, c, rows);
}
// the rows whose weights lie inside [p, p + bytes) only, on the CURRENT stream; the use epoch does not advance
// (msk_sgd_momentum_eager: one call per block of the model, msk_wbf_prepack_impl closes the step)
int msk_wbf_prepack_range_impl(msk_ctx* ctx, const void* p, size_t bytes) {
  if (ctx->wbf_prepack && p && msk_small_prepack(ctx, p, bytes) != 0) return -1;
  if (!ctx->wpack || !ctx->wbf_prepack || !p) return 0;
  WbfPackCache* c = (WbfPackCache*)ctx->wpack;
  const char* a0 = (const char*)p;
  const char* a1 = a0 + bytes;
  std::vector<int> rows;
  for (int r = 0; r < WbfPackCache::kRows - 1; ++r) {
    const WbfPackEntry& e = c->e[r];
    if (!e.live || e.valid || e.last_use < c->epoch - 1) continue;
    const char* b0 = (const char*)e.d.w;
    const char* b1 = b0 + (size_t)e.d.count * sizeof(float);
    if (b0 >= a0 && b1 <= a1) rows.push_back(r);   // wholly inside: a row that straddles the slice waits for the full rebuild
  }
  return pack_rows_build(ctx, c, rows);
}
void msk_wbf_pack_cache_free(msk_ctx* ctx) {
  if (!ctx->wpack) return;
  WbfPackCache* c = (WbfPackCache*)ctx->wpack;
  for (int r = 0; r < WbfPackCache::kRows - 1; ++r)
    if (c->e[r].live && c->e[r].d.out) hipFree(c->e[r].d.out);
  if (c->table) hipFree(c->table);
  if (c->amax) hipFree(c->amax);
  delete c;
  ctx->wpack = nullptr;
}
