// 'Same' 5x5x5 convolution (forward and data gradient of every LUConv layer, vnet.py:36) with a 1-D Winograd
// F(2,5) transform along W: two outputs of a W row cost 6 multiplications per (kd, kh, ci, co) instead of 10,
// i.e. 75 instead of 125 MFMA-MACs per output voxel and channel pair (1.67x fewer than conv_halo_mfma_k).
//
//   y[2t+i] = sum_xi AT[i][xi] * sum_{kd,kh,ci} V_xi[d+kd, h+kh, t][ci] * U_xi[kd,kh][ci][co]
//   V_xi = sum_j BT[xi][j] x[.., 2t+j]   (j = 0..5),     U_xi = sum_kw G[xi][kw] w[kd,kh,kw]
//   interpolation points {0, 1, -1, 2, -2, inf}: fp32 error ~1e-6 of max|y| over K = 800 products
//   (numerical study in DESIGN.md; the direct kernel is ~1e-7) -- inside the 2e-5 conv tolerance of the tests.
//
// Structure: the raw halo tile is staged in LDS exactly like conv_halo_mfma_k (8-channel chunks, [quad][voxel][4],
// rows split by W parity so that the stride-2 accesses x[2t+j] of a wavefront are contiguous); the INPUT TRANSFORM
// runs in registers right before the MFMAs (6 LDS quads -> 6 transformed quads per (kd, kh), ~20 float4 VALU ops
// that issue alongside 24 MFMAs); a wavefront owns 32 (d, h, t) positions and keeps the 6 xi-accumulators, so the
// OUTPUT TRANSFORM is register math in the epilogue.
#include "msk_conv.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct WinoArgs {
  const float* src;
  int sld;
  float* dst;
  int dld;
  int N, D, H, W;         // LOGICAL dims: the kernel tiles 4 x 8 x {8,16} over (D, H, W) and transforms along W ...
  long svn;               // ... which may be any permutation of the tensor's axes: voxel index =
  int svd, svh, svw;      //     n*svn + d*svd + h*svh + w*svw  (identity: H*W, W, 1)
  int CK, CN;
  const float4* um;  // [kd*5+kh][KC][2][xi][npad] float4 (k = kc*8 + h*4 + q)
  unsigned um_bytes;
  unsigned src_bytes;  // extent of src for the staging buffer loads (< 4 GiB: run_gconv chunks the batch)
  int KC, npad;
  const float* bias;
  const float* prelu;  // per-channel slope applied after bias (inference epilogue), or null
  int accumulate;
  int tiles_d, tiles_h, tiles_w, nblk;
  int vec;
  // deterministic split-K over the 8-channel chunks for layers with few tiles (256ch @ 16^3 / 8^3): blockIdx.z owns
  // chunks [z*kc_per, (z+1)*kc_per) and writes an fp32 slab partial[z][voxel][CN]; wino_splitk_reduce_k adds them in order
  int ksplit, kc_per;
  float* partial;
};

// Transformed weights through a raw buffer resource: the (kc, row, xi) part of the address is wave-uniform and goes
// into the scalar offset, the lane part is one constant byte offset -- no 64-bit vector address arithmetic per load
// (the flat-pointer form cost 9 v_lshl_add_u64 per 32 MFMAs).
typedef unsigned int v4u_t __attribute__((vector_size(16)));
__device__ __forceinline__ float4 ubuf_load(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned uniform_bytes) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_bytes, (int)uniform_bytes, 0));
}

__device__ __forceinline__ int xcd_remap_w(int bid, int nb) {
  const int q = nb >> 3, r = nb & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// U[r][kc][h][xi][n][q] = sum_kw G[xi][kw] * w(tap = r*5 + kw (flipped when flip), k = kc*8+h*4+q, n)
__global__ void __launch_bounds__(256)
pack_wino_weights_k(const float* __restrict__ w, int A, int B, int swap, int flip, int CK, int CN, int KC, int npad,
                    int tsd, int tsh, int tsw, float* __restrict__ out) {
  const float G[6][5] = {{0.25f, 0.f, 0.f, 0.f, 0.f},
                         {-1.f / 6, -1.f / 6, -1.f / 6, -1.f / 6, -1.f / 6},
                         {-1.f / 6, 1.f / 6, -1.f / 6, 1.f / 6, -1.f / 6},
                         {1.f / 24, 1.f / 12, 1.f / 6, 1.f / 3, 2.f / 3},
                         {1.f / 24, -1.f / 12, 1.f / 6, -1.f / 3, 2.f / 3},
                         {0.f, 0.f, 0.f, 0.f, 1.f}};
  // one thread per (row, kc, h, n, q): the 5 taps are read ONCE and all 6 xi planes written (one thread per output
  // re-read them 6 times: PMC 429 MB of traffic per launch for a 33 MB weight tensor, 0.12 ms per 256-channel layer)
  const long total = 25L * KC * 2 * npad * 4;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int q = (int)(idx & 3);
    long r_ = idx >> 2;
    const int n = (int)(r_ % npad);
    r_ /= npad;
    const int h = (int)(r_ & 1);
    r_ >>= 1;
    const int kc = (int)(r_ % KC);
    const int row = (int)(r_ / KC);
    const int k = kc * 8 + h * 4 + q;
    double t[5] = {0, 0, 0, 0, 0};
    if (k < CK && n < CN) {
      const int ia = swap ? n : k, ib = swap ? k : n;
      const float* wp = w + ((long)ia * B + ib) * 125;
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) {
        const int tap = (row / 5) * tsd + (row % 5) * tsh + kw * tsw;  // logical (kd, kh, kw) -> canonical tap
        t[kw] = (double)wp[flip ? 124 - tap : tap];
      }
    }
    // out index = ((((row*KC + kc)*2 + h)*6 + xi)*npad + n)*4 + q
    float* o = out + ((((long)(row * KC + kc) * 2 + h) * 6) * npad + n) * 4 + q;
#pragma unroll
    for (int xi = 0; xi < 6; ++xi) {
      double s_ = 0.0;  // transform in double: the weights are packed once per call, cheap
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) s_ += (double)G[xi][kw] * t[kw];
      o[(long)xi * npad * 4] = (float)s_;
    }
  }
}

typedef float f2 __attribute__((ext_vector_type(2)));

// One half (two channels) of a quad as a packed pair: the compiler maps these onto v_pk_fma_f32 / v_pk_add_f32 /
// v_pk_mul_f32, i.e. 13 VALU instructions per half for the whole 6-point input transform.  (PMC: MFMA 69 % + other
// VALU 18 % of the cycles -- fp32 MFMA and the vector ALU share the SIMD's issue, so every transform instruction
// is paid in MFMA time.)
struct Q2 {
  f2 lo, hi;
};
__device__ __forceinline__ Q2 q2(const float4& v) { return Q2{(f2){v.x, v.y}, (f2){v.z, v.w}}; }

__device__ __forceinline__ void wino_bt(const f2 x0, const f2 x1, const f2 x2, const f2 x3, const f2 x4, const f2 x5,
                                        f2 (&v)[6]) {
  const f2 c4 = {4.f, 4.f}, c5 = {5.f, 5.f}, c2 = {2.f, 2.f};
  v[0] = c4 * x0 + (x4 - c5 * x2);
  const f2 pa = x4 - c4 * x2, qa = x3 - c4 * x1;
  v[1] = pa + qa;
  v[2] = pa - qa;
  const f2 pb = x4 - x2, qb = c2 * (x3 - x1);
  v[3] = pb + qb;
  v[4] = pb - qb;
  v[5] = c4 * x1 + (x5 - c5 * x3);
}

// tile: 4 (D) x 8 (H) x 8 (W) outputs = 128 (d, h, t) positions, t = W pair; wave = d plane
__global__ void __launch_bounds__(256, 2)
conv_halo_wino_k(WinoArgs a) {
  constexpr int TD = 4, TH = 8, TW = 8, P = 2;
  constexpr int HD = TD + 2 * P, HH = TH + 2 * P, HW = TW + 2 * P;  // 8 x 12 x 12
  constexpr int HWH = HW / 2;                                         // 6 quads per parity half-row
  constexpr int NV = HD * HH * HW, NVP = NV | 1;
  __shared__ float4 lds[2 * NVP];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  int tile = xcd_remap_w(blockIdx.x, a.nblk);
  const int twi = tile % a.tiles_w;
  tile /= a.tiles_w;
  const int thi = tile % a.tiles_h;
  tile /= a.tiles_h;
  const int tdi = tile % a.tiles_d;
  const int n = tile / a.tiles_d;
  const int d0 = tdi * TD, h0 = thi * TH, w0 = twi * TW;
  const int nt = blockIdx.y;

  // A row of this lane: (dz = wave, hy = li / 4, t = li % 4); LDS index of x[.., 2t + j]:
  //   lh*NVP + ((dz + kd)*HH + hy + kh)*HW + (j & 1)*HWH + t + (j >> 1)
  const int abase = lh * NVP + (wave * HH + (li >> 2)) * HW + (li & 3);

  f32x16 acc[6];
#pragma unroll
  for (int x = 0; x < 6; ++x)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[x][j] = 0.f;

  const long rowstride = (long)a.KC * 2 * 6 * a.npad;  // float4 units between (kd, kh) rows
  const unsigned ulane_off = (unsigned)(lh * 6 * a.npad + nt * 32 + li) * 16u;  // bytes
  const __amdgpu_buffer_rsrc_t ures = __builtin_amdgcn_make_buffer_rsrc((void*)a.um, 0, a.um_bytes, 0x00020000);

  const int kc_begin = a.ksplit > 1 ? (int)blockIdx.z * a.kc_per : 0;
  const int kc_end = a.ksplit > 1 ? min(a.KC, kc_begin + a.kc_per) : a.KC;
  for (int kc = kc_begin; kc < kc_end; ++kc) {
    __syncthreads();
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
          const float* p = a.src + ((long)n * a.svn + (long)gd * a.svd + (long)gh * a.svh + (long)gw * a.svw) * a.sld + c0;
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
        if (it < NV * 2) {
          const int hv = it >> 1, q = it & 1;
          const int hw = hv % HW, rowi = hv / HW;  // rowi = hd*HH + hh
          lds[q * NVP + rowi * HW + (hw & 1) * HWH + (hw >> 1)] = tmp[i];
        }
      }
    }
    __syncthreads();

#pragma unroll 1
    for (int rr = 0; rr < 25; ++rr) {
      const int kd = rr / 5, kh = rr % 5;
      const float4* row = lds + abase + (kd * HH + kh) * HW;
      const Q2 x0 = q2(row[0]), x1 = q2(row[HWH]), x2 = q2(row[1]), x3 = q2(row[HWH + 1]), x4 = q2(row[2]),
               x5 = q2(row[HWH + 2]);
      float4 b[6];
      const unsigned ubase = (unsigned)(((long)kc * 2 * 6 * a.npad + rr * rowstride) * 16);
#pragma unroll
      for (int x = 0; x < 6; ++x) b[x] = ubuf_load(ures, ulane_off, ubase + (unsigned)(x * a.npad) * 16u);
      // V = BT x  (points 0, 1, -1, 2, -2, inf), two channels per packed instruction
      f2 vl[6], vh[6];
      wino_bt(x0.lo, x1.lo, x2.lo, x3.lo, x4.lo, x5.lo, vl);
      wino_bt(x0.hi, x1.hi, x2.hi, x3.hi, x4.hi, x5.hi, vh);
#pragma unroll
      for (int x = 0; x < 6; ++x) {
        acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(vl[x].x, b[x].x, acc[x], 0, 0, 0);
        acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(vl[x].y, b[x].y, acc[x], 0, 0, 0);
        acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(vh[x].x, b[x].z, acc[x], 0, 0, 0);
        acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(vh[x].y, b[x].w, acc[x], 0, 0, 0);
      }
    }
  }

  // output transform + store: y0 = m0+m1+m2+m3+m4, y1 = m1-m2+2(m3-m4)+m5
  const int co = nt * 32 + li;
  if (co < a.CN) {
    const float bv = a.bias ? a.bias[co] : 0.f;
    const float slope = a.prelu ? a.prelu[co] : 1.f;
    const int gd = d0 + wave;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int row = (j & 3) + 8 * (j >> 2) + 4 * lh;  // = hy*4 + t
      const int gh = h0 + (row >> 2), gw = w0 + 2 * (row & 3);
      if (gd < a.D && gh < a.H && gw < a.W) {
        const float m1 = acc[1][j], m2 = acc[2][j], m3 = acc[3][j], m4 = acc[4][j];
        const float y0 = ((acc[0][j] + m1) + (m2 + m3)) + m4;
        const float y1 = ((m1 - m2) + 2.f * (m3 - m4)) + acc[5][j];
        const long vox = (long)n * a.svn + (long)gd * a.svd + (long)gh * a.svh + (long)gw * a.svw;
        if (a.ksplit > 1) {
          float* pp = a.partial + ((long)blockIdx.z * ((long)a.N * a.D * a.H * a.W) + vox) * a.CN + co;
          pp[0] = y0;
          if (gw + 1 < a.W) pp[(long)a.svw * a.CN] = y1;
          continue;
        }
        float* o = a.dst + vox * a.dld + co;
        float r0 = y0 + bv;
        if (a.accumulate) r0 += o[0];
        o[0] = r0 > 0.f ? r0 : slope * r0;
        if (gw + 1 < a.W) {
          float r1 = y1 + bv;
          float* o1 = o + (long)a.svw * a.dld;
          if (a.accumulate) r1 += *o1;
          *o1 = r1 > 0.f ? r1 : slope * r1;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// F(4,5): four outputs of a W row from 8 multiplications (points 0, +-1, +-2, +-1/2, inf) -> 50 of the 125 direct
// MACs per output (F(2,5): 75).  fp32 error ~4.5e-6 of max|y| at K = 800 (tools/winograd_numerics.py; F(2,5) 1.3e-6,
// direct 6e-7).  Same structure as conv_halo_wino_k: tile 4 x 8 x 16 outputs = 128 (d, h, W-quad) positions, a wave
// owns one d plane and the 8 xi accumulators (128 VGPRs); halo rows stored by W residue mod 4 so that the stride-4
// accesses x[4t + j] of a wavefront are contiguous and the four rows of a 16-lane group sit on disjoint banks.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
pack_wino4_weights_k(const float* __restrict__ w, int A, int B, int swap, int flip, int CK, int CN, int KC, int npad,
                     int tsd, int tsh, int tsw, float* __restrict__ out) {
  const double G[8][5] = {{-1, 0, 0, 0, 0},
                          {-2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9},
                          {-2.0 / 9, 2.0 / 9, -2.0 / 9, 2.0 / 9, -2.0 / 9},
                          {1.0 / 90, 1.0 / 45, 2.0 / 45, 4.0 / 45, 8.0 / 45},
                          {1.0 / 90, -1.0 / 45, 2.0 / 45, -4.0 / 45, 8.0 / 45},
                          {32.0 / 45, 16.0 / 45, 8.0 / 45, 4.0 / 45, 2.0 / 45},
                          {32.0 / 45, -16.0 / 45, 8.0 / 45, -4.0 / 45, 2.0 / 45},
                          {0, 0, 0, 0, 1}};
  // one thread per (row, kc, h, n, q): the 5 taps are read ONCE and all 8 xi planes written (the first version had
  // one thread per output and re-read the taps 8 times: 1.5 ms per training step in this kernel)
  const long total = 25L * KC * 2 * npad * 4;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int q = (int)(idx & 3);
    long r_ = idx >> 2;
    const int n = (int)(r_ % npad);
    r_ /= npad;
    const int h = (int)(r_ & 1);
    r_ >>= 1;
    const int kc = (int)(r_ % KC);
    const int row = (int)(r_ / KC);
    const int k = kc * 8 + h * 4 + q;
    double t[5] = {0, 0, 0, 0, 0};
    if (k < CK && n < CN) {
      const int ia = swap ? n : k, ib = swap ? k : n;
      const float* wp = w + ((long)ia * B + ib) * 125;
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) {
        const int tap = (row / 5) * tsd + (row % 5) * tsh + kw * tsw;  // logical (kd, kh, kw) -> canonical tap
        t[kw] = (double)wp[flip ? 124 - tap : tap];
      }
    }
    // out index = ((((row*KC + kc)*2 + h)*8 + xi)*npad + n)*4 + q
    float* o = out + ((((long)(row * KC + kc) * 2 + h) * 8) * npad + n) * 4 + q;
#pragma unroll
    for (int xi = 0; xi < 8; ++xi) {
      double s_ = 0.0;
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) s_ += G[xi][kw] * t[kw];
      o[(long)xi * npad * 4] = (float)s_;
    }
  }
}

__device__ __forceinline__ void wino4_bt(const f2 d0, const f2 d1, const f2 d2, const f2 d3, const f2 d4, const f2 d5,
                                         const f2 d6, const f2 d7, f2 (&v)[8]) {
  const f2 c525 = {5.25f, 5.25f}, c425 = {4.25f, 4.25f}, c025 = {0.25f, 0.25f}, c125 = {1.25f, 1.25f};
  const f2 c05 = {0.5f, 0.5f}, c25 = {2.5f, 2.5f}, c2 = {2.f, 2.f}, c4 = {4.f, 4.f}, c5 = {5.f, 5.f};
  v[0] = (d6 - d0) + c525 * (d2 - d4);
  v[7] = (d7 - d1) + c525 * (d3 - d5);
  const f2 t1 = (d2 + d6) - c425 * d4, t2 = (d1 + d5) - c425 * d3;
  v[1] = t1 + t2;
  v[2] = t1 - t2;
  const f2 t3 = (d6 + c025 * d2) - c125 * d4, t4 = (c05 * d1 - c25 * d3) + c2 * d5;
  v[3] = t3 + t4;
  v[4] = t3 - t4;
  const f2 t5 = (d6 + c4 * d2) - c5 * d4, t6 = (c2 * d1 - c25 * d3) + c05 * d5;
  v[5] = t5 + t6;
  v[6] = t5 - t6;
}

// The same transform with every operation forced onto the packed fp32 pipe (v_pk_add_f32 / v_pk_fma_f32 /
// v_pk_mul_f32, constants as SGPR pairs): left to itself the compiler emits about half of the transform as scalar
// v_add_f32 / v_fma_f32 pairs (41 scalar + 30 packed instructions per 32 MFMAs; fp32 MFMA shares the SIMD's issue with
// them), this is 25 packed instructions per two channels.  2.5*d3 is shared between t4 and t6.
__device__ __forceinline__ f2 pk_add(f2 a, f2 b) {
  f2 r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f2 pk_sub(f2 a, f2 b) {
  f2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f2 pk_mul(f2 a, f2 c) {  // c: wave-uniform constant pair
  f2 r;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "s"(c));
  return r;
}
__device__ __forceinline__ f2 pk_fma(f2 a, f2 c, f2 b) {  // a*c + b
  f2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(c), "v"(b));
  return r;
}
__device__ __forceinline__ f2 pk_fnma(f2 a, f2 c, f2 b) {  // b - a*c
  f2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "s"(c), "v"(b));
  return r;
}
__device__ __forceinline__ f2 pk_fms(f2 a, f2 c, f2 b) {  // a*c - b
  f2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "s"(c), "v"(b));
  return r;
}
__device__ __forceinline__ void wino4_bt_pk(const f2 d0, const f2 d1, const f2 d2, const f2 d3, const f2 d4, const f2 d5,
                                            const f2 d6, const f2 d7, f2 (&v)[8]) {
  const f2 c525 = {5.25f, 5.25f}, c425 = {4.25f, 4.25f}, c025 = {0.25f, 0.25f}, c125 = {1.25f, 1.25f};
  const f2 c05 = {0.5f, 0.5f}, c25 = {2.5f, 2.5f}, c2 = {2.f, 2.f}, c4 = {4.f, 4.f}, c5 = {5.f, 5.f};
  v[0] = pk_fma(pk_sub(d2, d4), c525, pk_sub(d6, d0));
  v[7] = pk_fma(pk_sub(d3, d5), c525, pk_sub(d7, d1));
  const f2 t1 = pk_fnma(d4, c425, pk_add(d2, d6)), t2 = pk_fnma(d3, c425, pk_add(d1, d5));
  v[1] = pk_add(t1, t2);
  v[2] = pk_sub(t1, t2);
  const f2 m3 = pk_mul(d3, c25);
  const f2 t3 = pk_fnma(d4, c125, pk_fma(d2, c025, d6)), t4 = pk_fma(d5, c2, pk_fms(d1, c05, m3));
  v[3] = pk_add(t3, t4);
  v[4] = pk_sub(t3, t4);
  const f2 t5 = pk_fnma(d4, c5, pk_fma(d2, c4, d6)), t6 = pk_fma(d5, c05, pk_fms(d1, c2, m3));
  v[5] = pk_add(t5, t6);
  v[6] = pk_sub(t5, t6);
}

__global__ void __launch_bounds__(256, 2)
conv_halo_wino4_k(WinoArgs a) {
  constexpr int TD = 4, TH = 8, TW = 16, P = 2;
  constexpr int HD = TD + 2 * P, HH = TH + 2 * P, HW = TW + 2 * P;  // 8 x 12 x 20
  constexpr int HWQ = HW / 4;                                         // 5 quads per W residue class
  constexpr int NV = HD * HH * HW, NVP = NV | 1;
  __shared__ float4 lds[2 * NVP];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  int tile = xcd_remap_w(blockIdx.x, a.nblk);
  const int twi = tile % a.tiles_w;
  tile /= a.tiles_w;
  const int thi = tile % a.tiles_h;
  tile /= a.tiles_h;
  const int tdi = tile % a.tiles_d;
  const int n = tile / a.tiles_d;
  const int d0 = tdi * TD, h0 = thi * TH, w0 = twi * TW;
  const int nt = blockIdx.y;

  // A row of this lane: (dz = wave, hy = li / 4, t = li % 4); LDS index of x[.., 4t + j]:
  //   lh*NVP + ((dz + kd)*HH + hy + kh)*HW + (j & 3)*HWQ + t + (j >> 2)
  const int abase = lh * NVP + (wave * HH + (li >> 2)) * HW + (li & 3);

  f32x16 acc[8];
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[x][j] = 0.f;

  const long rowstride = (long)a.KC * 2 * 8 * a.npad;  // float4 units between (kd, kh) rows
  const unsigned ulane_off = (unsigned)(lh * 8 * a.npad + nt * 32 + li) * 16u;  // bytes
  const __amdgpu_buffer_rsrc_t ures = __builtin_amdgcn_make_buffer_rsrc((void*)a.um, 0, a.um_bytes, 0x00020000);

  const int kc_begin = a.ksplit > 1 ? (int)blockIdx.z * a.kc_per : 0;
  const int kc_end = a.ksplit > 1 ? min(a.KC, kc_begin + a.kc_per) : a.KC;
  // Staging, vector path: a thread owns up to two (h, w, channel-quad) columns of the halo tile for the whole kernel and
  // walks the 8 d planes.  Its LDS slot and its byte offset inside the tensor are fixed per tile (computed once, not
  // per element and per channel chunk: the generic loop below spends ~13 VALU instructions per element on index
  // arithmetic, 15 elements per thread and chunk), the plane / channel-chunk part of the address is wave-uniform and
  // rides in the buffer load's scalar offset, and out-of-volume columns read zeros through an out-of-range offset.
  constexpr unsigned kOOBw = 0xFFFFFFF0u;
  const __amdgpu_buffer_rsrc_t sres = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
  int st_lds[2];
  unsigned st_off[2];
  bool st_live[2];
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_) {
    const int item = tid + s_ * 256;  // (hh, hw, q), q fastest: neighbouring lanes read neighbouring 16 B
    const int q = item & 1, col = item >> 1;
    const int hh = col / HW, hw = col % HW;
    const int gh = h0 - P + hh, gw = w0 - P + hw;
    st_live[s_] = item < HH * HW * 2;
    const bool ok = st_live[s_] && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
    st_lds[s_] = q * NVP + hh * HW + (hw & 3) * HWQ + (hw >> 2);
    st_off[s_] = ok ? (unsigned)((((long)n * a.svn + (long)gh * a.svh + (long)gw * a.svw) * a.sld + q * 4) * 4) : kOOBw;
  }

  for (int kc = kc_begin; kc < kc_end; ++kc) {
    __syncthreads();
    if (a.vec) {
      unsigned voff[2];
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) voff[s_] = (kc * 8 + (((tid + s_ * 256) & 1) << 2) < a.CK) ? st_off[s_] : kOOBw;
#pragma unroll
      for (int hd0 = 0; hd0 < HD; hd0 += 4) {
        float4 tmp[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int gd = d0 - P + hd0 + i;
          const bool dok = gd >= 0 && gd < a.D;  // wave-uniform
          const unsigned soff = dok ? (unsigned)(((long)gd * a.svd * a.sld + kc * 8) * 4) : 0u;
#pragma unroll
          for (int s_ = 0; s_ < 2; ++s_)
            tmp[i][s_] = dok ? ubuf_load(sres, voff[s_], soff) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int s_ = 0; s_ < 2; ++s_)
            if (st_live[s_]) lds[st_lds[s_] + (hd0 + i) * HH * HW] = tmp[i][s_];
      }
    } else {
    constexpr int SG = 5;
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
          const float* p = a.src + ((long)n * a.svn + (long)gd * a.svd + (long)gh * a.svh + (long)gw * a.svw) * a.sld + c0;
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
        if (it < NV * 2) {
          const int hv = it >> 1, q = it & 1;
          const int hw = hv % HW, rowi = hv / HW;
          lds[q * NVP + rowi * HW + (hw & 3) * HWQ + (hw >> 2)] = tmp[i];
        }
      }
    }
    }
    __syncthreads();

#pragma unroll 1
    for (int rr = 0; rr < 25; ++rr) {
      const int kd = rr / 5, kh = rr % 5;
      const float4* row = lds + abase + (kd * HH + kh) * HW;
      Q2 x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = q2(row[(j & 3) * HWQ + (j >> 2)]);
      float4 b[8];
      const unsigned ubase = (unsigned)(((long)kc * 2 * 8 * a.npad + rr * rowstride) * 16);
#pragma unroll
      for (int xq = 0; xq < 8; ++xq) b[xq] = ubuf_load(ures, ulane_off, ubase + (unsigned)(xq * a.npad) * 16u);
      f2 vl[8], vh[8];
      wino4_bt_pk(x[0].lo, x[1].lo, x[2].lo, x[3].lo, x[4].lo, x[5].lo, x[6].lo, x[7].lo, vl);
      wino4_bt_pk(x[0].hi, x[1].hi, x[2].hi, x[3].hi, x[4].hi, x[5].hi, x[6].hi, x[7].hi, vh);
      __builtin_amdgcn_s_setprio(2);  // the wavefront in its MFMA burst wins the issue arbitration over its SIMD mate
#pragma unroll
      for (int xq = 0; xq < 8; ++xq) {
        acc[xq] = __builtin_amdgcn_mfma_f32_32x32x2f32(vl[xq].x, b[xq].x, acc[xq], 0, 0, 0);
        acc[xq] = __builtin_amdgcn_mfma_f32_32x32x2f32(vl[xq].y, b[xq].y, acc[xq], 0, 0, 0);
        acc[xq] = __builtin_amdgcn_mfma_f32_32x32x2f32(vh[xq].x, b[xq].z, acc[xq], 0, 0, 0);
        acc[xq] = __builtin_amdgcn_mfma_f32_32x32x2f32(vh[xq].y, b[xq].w, acc[xq], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
    }
  }

  // output transform (AT, 4 x 8) + store
  const int co = nt * 32 + li;
  if (co < a.CN) {
    const float bv = a.bias ? a.bias[co] : 0.f;
    const float slope = a.prelu ? a.prelu[co] : 1.f;
    const int gd = d0 + wave;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int row = (j & 3) + 8 * (j >> 2) + 4 * lh;  // = hy*4 + t
      const int gh = h0 + (row >> 2), gw = w0 + 4 * (row & 3);
      if (gd < a.D && gh < a.H && gw < a.W) {
        const float s12 = acc[1][j] + acc[2][j], d12 = acc[1][j] - acc[2][j];
        const float s34 = acc[3][j] + acc[4][j], d34 = acc[3][j] - acc[4][j];
        const float s56 = acc[5][j] + acc[6][j], d56 = acc[5][j] - acc[6][j];
        float y[4];
        y[0] = ((acc[0][j] + s12) + s34) + s56;
        y[1] = (d12 + 2.f * d34) + 0.5f * d56;
        y[2] = (s12 + 4.f * s34) + 0.25f * s56;
        y[3] = ((d12 + 8.f * d34) + 0.125f * d56) + acc[7][j];
        const long vox = (long)n * a.svn + (long)gd * a.svd + (long)gh * a.svh + (long)gw * a.svw;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (gw + i < a.W) {
            if (a.ksplit > 1) {
              a.partial[((long)blockIdx.z * ((long)a.N * a.D * a.H * a.W) + vox + (long)i * a.svw) * a.CN + co] = y[i];
            } else {
              float* o = a.dst + (vox + (long)i * a.svw) * a.dld + co;
              float r = y[i] + bv;
              if (a.accumulate) r += *o;
              *o = r > 0.f ? r : slope * r;
            }
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// F(4,3): the 3 x 3 x 3 'same' convolutions (UNet3D's DoubleConvs, the deep-supervision heads): four outputs of a row
// from 6 multiplications (points 0, +-1, +-2, inf -- the same six points as F(2,5), hence the same input transform)
// -> 18 of the 27 direct MACs per output... per axis 6/12 = 0.5.  Same structure as conv_halo_wino4_k: tile 4 x 8 x 16
// outputs, halo 6 x 10 x 18 stored by W residue mod 4 (row pitch 20 quads), 9 (kd, kh) rows, 6 xi accumulators.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wino_bt_pk(const f2 d0, const f2 d1, const f2 d2, const f2 d3, const f2 d4, const f2 d5,
                                           f2 (&v)[6]) {
  const f2 c4 = {4.f, 4.f}, c5 = {5.f, 5.f}, c2 = {2.f, 2.f};
  v[0] = pk_fma(d0, c4, pk_fnma(d2, c5, d4));
  const f2 pa = pk_fnma(d2, c4, d4), qa = pk_fnma(d1, c4, d3);
  v[1] = pk_add(pa, qa);
  v[2] = pk_sub(pa, qa);
  const f2 pb = pk_sub(d4, d2), sd = pk_sub(d3, d1);
  v[3] = pk_fma(sd, c2, pb);
  v[4] = pk_fnma(sd, c2, pb);
  v[5] = pk_fma(d1, c4, pk_fnma(d3, c5, d5));
}

// U[r][kc][h][xi][n][q] = sum_kw G[xi][kw] * w(tap, k = kc*8+h*4+q, n), r = kd*3 + kh, canonical tap from the permuted axes
__global__ void __launch_bounds__(256)
pack_wino43_weights_k(const float* __restrict__ w, int A, int B, int swap, int flip, int CK, int CN, int KC, int npad,
                      int tsd, int tsh, int tsw, float* __restrict__ out) {
  const double G[6][3] = {{0.25, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                          {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
  const long total = 9L * KC * 2 * npad * 4;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int q = (int)(idx & 3);
    long r_ = idx >> 2;
    const int n = (int)(r_ % npad);
    r_ /= npad;
    const int h = (int)(r_ & 1);
    r_ >>= 1;
    const int kc = (int)(r_ % KC);
    const int row = (int)(r_ / KC);
    const int k = kc * 8 + h * 4 + q;
    double t[3] = {0, 0, 0};
    if (k < CK && n < CN) {
      const int ia = swap ? n : k, ib = swap ? k : n;
      const float* wp = w + ((long)ia * B + ib) * 27;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int tap = (row / 3) * tsd + (row % 3) * tsh + kw * tsw;
        t[kw] = (double)wp[flip ? 26 - tap : tap];
      }
    }
    float* o = out + ((((long)(row * KC + kc) * 2 + h) * 6) * npad + n) * 4 + q;
#pragma unroll
    for (int xi = 0; xi < 6; ++xi) o[(long)xi * npad * 4] = (float)(G[xi][0] * t[0] + G[xi][1] * t[1] + G[xi][2] * t[2]);
  }
}

__global__ void __launch_bounds__(256, 3)  // 96 accumulator registers leave room for three wavefronts per SIMD
conv_halo_wino43_k(WinoArgs a) {
  constexpr int TD = 4, TH = 8, TW = 16, P = 1;
  constexpr int HD = TD + 2 * P, HH = TH + 2 * P, HW = TW + 2 * P;  // 6 x 10 x 18
  constexpr int HWQ = (HW + 3) / 4, RPW = 4 * HWQ;                    // 5 slots per W residue class, row pitch 20
  constexpr int NV = HD * HH * HW, NVP = (HD * HH * RPW) | 1;
  __shared__ float4 lds[2 * NVP];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  int tile = xcd_remap_w(blockIdx.x, a.nblk);
  const int twi = tile % a.tiles_w;
  tile /= a.tiles_w;
  const int thi = tile % a.tiles_h;
  tile /= a.tiles_h;
  const int tdi = tile % a.tiles_d;
  const int n = tile / a.tiles_d;
  const int d0 = tdi * TD, h0 = thi * TH, w0 = twi * TW;
  const int nt = blockIdx.y;

  // A row of this lane: (dz = wave, hy = li / 4, t = li % 4); LDS index of x[.., 4t + j]:
  //   lh*NVP + ((dz + kd)*HH + hy + kh)*HW + (j & 3)*HWQ + t + (j >> 2)
  const int abase = lh * NVP + (wave * HH + (li >> 2)) * RPW + (li & 3);

  f32x16 acc[6];
#pragma unroll
  for (int x = 0; x < 6; ++x)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[x][j] = 0.f;

  const long rowstride = (long)a.KC * 2 * 6 * a.npad;  // float4 units between (kd, kh) rows
  const unsigned ulane_off = (unsigned)(lh * 6 * a.npad + nt * 32 + li) * 16u;  // bytes
  const __amdgpu_buffer_rsrc_t ures = __builtin_amdgcn_make_buffer_rsrc((void*)a.um, 0, a.um_bytes, 0x00020000);

  const int kc_begin = a.ksplit > 1 ? (int)blockIdx.z * a.kc_per : 0;
  const int kc_end = a.ksplit > 1 ? min(a.KC, kc_begin + a.kc_per) : a.KC;
  // Staging, vector path: a thread owns up to two (h, w, channel-quad) columns of the halo tile for the whole kernel and
  // walks the 8 d planes.  Its LDS slot and its byte offset inside the tensor are fixed per tile (computed once, not
  // per element and per channel chunk: the generic loop below spends ~13 VALU instructions per element on index
  // arithmetic, 15 elements per thread and chunk), the plane / channel-chunk part of the address is wave-uniform and
  // rides in the buffer load's scalar offset, and out-of-volume columns read zeros through an out-of-range offset.
  constexpr unsigned kOOBw = 0xFFFFFFF0u;
  const __amdgpu_buffer_rsrc_t sres = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
  int st_lds[2];
  unsigned st_off[2];
  bool st_live[2];
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_) {
    const int item = tid + s_ * 256;  // (hh, hw, q), q fastest: neighbouring lanes read neighbouring 16 B
    const int q = item & 1, col = item >> 1;
    const int hh = col / HW, hw = col % HW;
    const int gh = h0 - P + hh, gw = w0 - P + hw;
    st_live[s_] = item < HH * HW * 2;
    const bool ok = st_live[s_] && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
    st_lds[s_] = q * NVP + hh * RPW + (hw & 3) * HWQ + (hw >> 2);
    st_off[s_] = ok ? (unsigned)((((long)n * a.svn + (long)gh * a.svh + (long)gw * a.svw) * a.sld + q * 4) * 4) : kOOBw;
  }

  for (int kc = kc_begin; kc < kc_end; ++kc) {
    __syncthreads();
    if (a.vec) {
      unsigned voff[2];
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) voff[s_] = (kc * 8 + (((tid + s_ * 256) & 1) << 2) < a.CK) ? st_off[s_] : kOOBw;
#pragma unroll
      for (int hd0 = 0; hd0 < HD; hd0 += 3) {
        float4 tmp[3][2];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int gd = d0 - P + hd0 + i;
          const bool dok = gd >= 0 && gd < a.D;  // wave-uniform
          const unsigned soff = dok ? (unsigned)(((long)gd * a.svd * a.sld + kc * 8) * 4) : 0u;
#pragma unroll
          for (int s_ = 0; s_ < 2; ++s_)
            tmp[i][s_] = dok ? ubuf_load(sres, voff[s_], soff) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int s_ = 0; s_ < 2; ++s_)
            if (st_live[s_]) lds[st_lds[s_] + (hd0 + i) * HH * RPW] = tmp[i][s_];
      }
    } else {
    constexpr int SG = 5;
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
          const float* p = a.src + ((long)n * a.svn + (long)gd * a.svd + (long)gh * a.svh + (long)gw * a.svw) * a.sld + c0;
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
        if (it < NV * 2) {
          const int hv = it >> 1, q = it & 1;
          const int hw = hv % HW, rowi = hv / HW;
          lds[q * NVP + rowi * RPW + (hw & 3) * HWQ + (hw >> 2)] = tmp[i];
        }
      }
    }
    }
    __syncthreads();

#pragma unroll 1
    for (int rr = 0; rr < 9; ++rr) {
      const int kd = rr / 3, kh = rr % 3;
      const float4* row = lds + abase + (kd * HH + kh) * RPW;
      Q2 x[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) x[j] = q2(row[(j & 3) * HWQ + (j >> 2)]);
      float4 b[6];
      const unsigned ubase = (unsigned)(((long)kc * 2 * 6 * a.npad + rr * rowstride) * 16);
#pragma unroll
      for (int xq = 0; xq < 6; ++xq) b[xq] = ubuf_load(ures, ulane_off, ubase + (unsigned)(xq * a.npad) * 16u);
      f2 vl[6], vh[6];
      wino_bt_pk(x[0].lo, x[1].lo, x[2].lo, x[3].lo, x[4].lo, x[5].lo, vl);
      wino_bt_pk(x[0].hi, x[1].hi, x[2].hi, x[3].hi, x[4].hi, x[5].hi, vh);
      __builtin_amdgcn_s_setprio(2);  // the wavefront in its MFMA burst wins the issue arbitration over its SIMD mate
#pragma unroll
      for (int xq = 0; xq < 6; ++xq) {
        acc[xq] = __builtin_amdgcn_mfma_f32_32x32x2f32(vl[xq].x, b[xq].x, acc[xq], 0, 0, 0);
        acc[xq] = __builtin_amdgcn_mfma_f32_32x32x2f32(vl[xq].y, b[xq].y, acc[xq], 0, 0, 0);
        acc[xq] = __builtin_amdgcn_mfma_f32_32x32x2f32(vh[xq].x, b[xq].z, acc[xq], 0, 0, 0);
        acc[xq] = __builtin_amdgcn_mfma_f32_32x32x2f32(vh[xq].y, b[xq].w, acc[xq], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
    }
  }

  // output transform (AT, 4 x 6) + store
  const int co = nt * 32 + li;
  if (co < a.CN) {
    const float bv = a.bias ? a.bias[co] : 0.f;
    const float slope = a.prelu ? a.prelu[co] : 1.f;
    const int gd = d0 + wave;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int row = (j & 3) + 8 * (j >> 2) + 4 * lh;  // = hy*4 + t
      const int gh = h0 + (row >> 2), gw = w0 + 4 * (row & 3);
      if (gd < a.D && gh < a.H && gw < a.W) {
        const float s12 = acc[1][j] + acc[2][j], d12 = acc[1][j] - acc[2][j];
        const float s34 = acc[3][j] + acc[4][j], d34 = acc[3][j] - acc[4][j];
        float y[4];
        y[0] = (acc[0][j] + s12) + s34;
        y[1] = d12 + 2.f * d34;
        y[2] = s12 + 4.f * s34;
        y[3] = (d12 + 8.f * d34) + acc[5][j];
        const long vox = (long)n * a.svn + (long)gd * a.svd + (long)gh * a.svh + (long)gw * a.svw;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (gw + i < a.W) {
            if (a.ksplit > 1) {
              a.partial[((long)blockIdx.z * ((long)a.N * a.D * a.H * a.W) + vox + (long)i * a.svw) * a.CN + co] = y[i];
            } else {
              float* o = a.dst + (vox + (long)i * a.svw) * a.dld + co;
              float r = y[i] + bv;
              if (a.accumulate) r += *o;
              *o = r > 0.f ? r : slope * r;
            }
          }
        }
      }
    }
  }
}

__global__ void __launch_bounds__(256)
wino_splitk_reduce_k(const float* __restrict__ partial, int ksplit, long voxels, int CN, const float* __restrict__ bias,
                     const float* __restrict__ prelu, float* __restrict__ dst, int dld, int accumulate) {
  const long total = voxels * CN;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long v = i / CN;
    const int c = (int)(i - v * CN);
    float s = bias ? bias[c] : 0.f;
    for (int z = 0; z < ksplit; ++z) s += partial[(long)z * total + i];  // fixed order
    float* o = dst + v * dld + c;
    if (accumulate) s += *o;
    if (prelu && s < 0.f) s *= prelu[c];
    *o = s;
  }
}

}  // namespace

int msk_gconv_halo_wino(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap) {
  const bool k5 = g.kd == 5 && g.kh == 5 && g.kw == 5 && g.pd == 2 && g.ph == 2 && g.pw == 2;
  const bool k3 = g.kd == 3 && g.kh == 3 && g.kw == 3 && g.pd == 1 && g.ph == 1 && g.pw == 1;  // F(4,3) only
  if (!(k5 || k3)) return 0;
  if (!(g.sd == 1 && g.sh == 1 && g.sw == 1)) return 0;
  if (!(g.SD == g.DD && g.SH == g.DH && g.SW == g.DW)) return 0;
  if (g.CK < 8 || g.CN < 8) return 0;                       // tiny-channel layers have their own kernels
  if (k3 && ctx->conv_impl == 14) return 0;
  // The kernels tile 4 x 8 x {8,16} over LOGICAL axes (d, h, w) and transform along w; any permutation of the tensor's
  // axes can play those roles (addresses are strided per voxel, the weights are packed with the taps permuted the same
  // way).  The anisotropic MRI slabs (512 x 512 x 12, W = 12 / 8 / 4) get their transform along H this way.
  // Preference: F(4,5) (w % 16 == 0) over F(2,5); among equals the least padding of a ragged d, then the identity.
  const int dims[3] = {g.DD, g.DH, g.DW};
  static const int kPerms[6][3] = {{0, 1, 2}, {1, 0, 2}, {0, 2, 1}, {2, 0, 1}, {1, 2, 0}, {2, 1, 0}};  // logical (d,h,w) <- axis
  int best = -1, best_f45 = 0;
  long best_cost = 0;
  for (int i = 0; i < 6; ++i) {
    const int ld_ = dims[kPerms[i][0]], lh_ = dims[kPerms[i][1]], lw_ = dims[kPerms[i][2]];
    if (lh_ % 8 || lw_ % 8) continue;                          // whole tiles in h and w; d may be ragged (bounds-checked)
    const int f = (lw_ % 16 == 0) && ctx->conv_impl != 14;     // 14 = F(2,5) only (A/B)
    if (k3 && !f) continue;                                    // 3^3: only the 16-wide F(4,3) kernel exists
    const long cost = (long)((ld_ + 3) / 4) * 4 * lh_ * lw_;   // padded volume
    if (cost * 2 > (long)ld_ * lh_ * lw_ * 3) continue;        // more than 1.5x padding: the direct kernel wins
    if (best < 0 || f > best_f45 || (f == best_f45 && cost < best_cost)) {
      best = i;
      best_f45 = f;
      best_cost = cost;
    }
  }
  if (best < 0) return 0;
  const int* pm = kPerms[best];
  const int LD = dims[pm[0]], LH = dims[pm[1]], LW = dims[pm[2]];
  const int vstr[3] = {g.DH * g.DW, g.DW, 1};   // voxel strides of the tensor's (D, H, W)
  const int tstr[3] = {k3 ? 9 : 25, k3 ? 3 : 5, 1};   // tap strides of the canonical weight's (kd, kh, kw)
  const bool f45 = best_f45 != 0;
  const int KC = (g.CK + 7) / 8;
  const int npad = ((g.CN + 31) / 32) * 32;
  const int twid = f45 ? 16 : 8, nxi = k3 ? 6 : (f45 ? 8 : 6), nrows = k3 ? 9 : 25;
  const long nblk = (long)g.N * ((LD + 3) / 4) * (LH / 8) * (LW / twid);
  // the direct kernel splits K when the tiling cannot fill the chip; leave those small layers to it
  if (nblk > 0x7fffffff) return 0;
  const size_t ubytes = (size_t)nxi * nrows * KC * 2 * npad * 4 * sizeof(float);
  float* um = (float*)msk_workspace2(ctx, ubytes);
  if (!um) return -1;
  {
    msk_launch_scope ls(ctx, "pack_weights_wino");
    long blocks = ((long)(ubytes / sizeof(float)) + 255) / 256;
    if (blocks > 8L * ctx->num_cu) blocks = 8L * ctx->num_cu;
    if (k3)
      hipLaunchKernelGGL(pack_wino43_weights_k, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, w_canon, A, B, swap,
                         g.transposed ? 1 : 0, g.CK, g.CN, KC, npad, tstr[pm[0]], tstr[pm[1]], tstr[pm[2]], um);
    else if (f45)
      hipLaunchKernelGGL(pack_wino4_weights_k, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, w_canon, A, B, swap,
                         g.transposed ? 1 : 0, g.CK, g.CN, KC, npad, tstr[pm[0]], tstr[pm[1]], tstr[pm[2]], um);
    else
      hipLaunchKernelGGL(pack_wino_weights_k, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, w_canon, A, B, swap,
                         g.transposed ? 1 : 0, g.CK, g.CN, KC, npad, tstr[pm[0]], tstr[pm[1]], tstr[pm[2]], um);
    MSK_LAUNCH_CHECK(ctx);
  }
  WinoArgs a{};
  a.src = g.src; a.sld = g.sld; a.dst = g.dst; a.dld = g.dld;
  a.N = g.N; a.D = LD; a.H = LH; a.W = LW; a.CK = g.CK; a.CN = g.CN;
  a.svn = (long)g.DD * g.DH * g.DW; a.svd = vstr[pm[0]]; a.svh = vstr[pm[1]]; a.svw = vstr[pm[2]];
  a.um = reinterpret_cast<const float4*>(um); a.KC = KC; a.npad = npad;
  if (ubytes >= 0xFFFFFFF0ull) return 0;  // 32-bit buffer offsets (52 MB for 256 -> 256 channels)
  a.um_bytes = (unsigned)ubytes;
  {
    const size_t sb = (size_t)g.N * g.SD * g.SH * g.SW * g.sld * sizeof(float);
    if (sb >= 0xFFFFFFF0ull) return 0;
    a.src_bytes = (unsigned)sb;
  }
  a.bias = g.bias; a.prelu = g.prelu; a.accumulate = g.accumulate;
  a.tiles_d = (LD + 3) / 4; a.tiles_h = LH / 8; a.tiles_w = LW / twid; a.nblk = (int)nblk;
  a.vec = (g.CK % 4 == 0) && (g.sld % 4 == 0) && (((uintptr_t)g.src) % 16 == 0);
  // split K when the (M, N) tiling alone cannot fill the chip (~4 workgroups per CU wanted), like conv_halo_mfma_k
  a.ksplit = 1;
  a.kc_per = KC;
  a.partial = nullptr;
  const long mn_blocks = nblk * (npad / 32);
  const long voxels = (long)g.N * g.DD * g.DH * g.DW;
  if (mn_blocks < 2L * ctx->num_cu && KC >= 2) {
    long want = (4L * ctx->num_cu + mn_blocks - 1) / mn_blocks;
    if (want > KC) want = KC;
    a.kc_per = (int)((KC + want - 1) / want);
    a.ksplit = (KC + a.kc_per - 1) / a.kc_per;
    if (a.ksplit > 1) {
      a.partial = (float*)msk_workspace(ctx, (size_t)a.ksplit * voxels * g.CN * sizeof(float));
      if (!a.partial) return -1;
    }
  }
  const char* tag = k3 ? "conv_halo_wino43_k" : (f45 ? "conv_halo_wino4_k" : "conv_halo_wino_k");
  if (ctx->prof && ctx->prof_shapes) {
    char buf[200];
    snprintf(buf, sizeof(buf), "%s[ck=%d,cn=%d,n=%d,dhw=%dx%dx%d,acc=%d]", tag, g.CK, g.CN, g.N, g.DD, g.DH, g.DW,
             g.accumulate);
    tag = msk_intern_tag(ctx, buf);
  }
  {
    msk_launch_scope ls(ctx, tag);
    if (k3) hipLaunchKernelGGL(conv_halo_wino43_k, dim3((unsigned)nblk, npad / 32, a.ksplit), dim3(256), 0, ctx->stream, a);
    else if (f45) hipLaunchKernelGGL(conv_halo_wino4_k, dim3((unsigned)nblk, npad / 32, a.ksplit), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(conv_halo_wino_k, dim3((unsigned)nblk, npad / 32, a.ksplit), dim3(256), 0, ctx->stream, a);
    MSK_LAUNCH_CHECK(ctx);
  }
  if (a.ksplit > 1) {
    msk_launch_scope ls(ctx, "conv_splitk_reduce");
    long blocks = (voxels * g.CN + 255) / 256;
    if (blocks > 8L * ctx->num_cu) blocks = 8L * ctx->num_cu;
    hipLaunchKernelGGL(wino_splitk_reduce_k, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const float*)a.partial,
                       a.ksplit, voxels, g.CN, g.bias, g.prelu, g.dst, g.dld, g.accumulate);
    MSK_LAUNCH_CHECK(ctx);
  }
  return 1;
}
