// Shared pieces of the bf16x3 Winograd pipelines (msk_conv_wbf.hip: forward / data gradient, msk_wgrad_wbf.hip: weight
// gradient): exact three-way bf16 split and the stage-1 transform kernels' argument block.
#pragma once
#include "msk_conv.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// x = hi + mid + lo exactly (x fp32, pieces bf16, round-to-nearest-even at each step), two values at a time
__device__ __forceinline__ void wbf_split3_pair(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
  f32x2 x = {x0, x1};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
  f32x2 r = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u)};
  mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
  f32x2 r2 = {r.x - __uint_as_float(mid << 16), r.y - __uint_as_float(mid & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

// Stage 1 (wbf_tin_k<MODE>): MODE 0  V = B^T x  (8 transformed values per 4 inputs, sliding 8-wide window along w)
//                            MODE 1  Y = A dy   (8 values per 4 output gradients: the adjoint of the output transform)
// both split into three bf16 pieces and written as V[xi][n][t][kc][piece][khalf][DP][HP] (16-byte slots of 8 channels,
// slot (dp, hp) = position (dp - 2, hp - 2), zero outside the volume).
struct WbfTinArgs {
  const float* src;
  int sld;
  long svn;
  int svd, svh, svw;  // voxel strides of the logical axes
  int N, LD, LH, LW, T, CK, KC;
  int DP, HP;
  char* V;
  long v_xi;  // bytes between xi planes
};
int msk_wbf_transform(msk_ctx* ctx, int mode, const WbfTinArgs& a);
