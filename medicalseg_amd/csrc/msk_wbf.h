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

// NP = 2: fp16 two-piece split  x*s = h + l,  h = fp16(x*s), l = fp16(x*s - h)  (22 significand bits where l is a normal
// fp16 number).  A product then is  a*b = ha*hb + ha*lb + la*hb + O(2^-22):  THREE fp16 MFMAs into the SAME fp32 accumulator
// instead of six bf16 ones.  fp16's exponent range is managed per TENSOR: s is a power of two derived on the device from
// (an upper bound of) the tensor's max |value| (wbf_scale_of: the max lands at 2^9..2^10, the Winograd transforms amplify by
// <= 21.25, fp16 overflows at 2^16); elements more than 2^12 below the max get a subnormal l, i.e. an ABSOLUTE error of
// 2^-25 in scaled units = 2^-35 of the tensor's max -- far below the fp32 rounding of the sums they enter.  The results are
// divided by the product of the operand scales in the output stage (exact: powers of two).
typedef _Float16 wbf_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void wbf_split2h_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
  f32x2 x = {x0, x1};
  const wbf_f16x2 h = __builtin_convertvector(x, wbf_f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  const f32x2 hf = __builtin_convertvector(h, f32x2);
  f32x2 r = {x0 - hf.x, x1 - hf.y};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, wbf_f16x2));
}
// Round 3: the format of the TRANSFORMED ACTIVATION / GRADIENT planes (V, Y) keeps the low piece SCALED by 2^11:
//     x*s = h + l,   h = fp16(x*s),   l' = fp16((x*s - h) * 2^11)
// |x*s - h| <= 2^-11 |h|, so l' has the magnitude of h and is a NORMAL fp16 number whenever h is (down to 2^-14 in scaled
// units = 2^-24 of the tensor's maximum) -- the plain low piece went subnormal 2^12 below the maximum, which left elements
// 2^17 / 2^20 below it with 17 / 14 significant bits (tests/test_gpu_wbf.py::test_wbf_fp16_split_sparse_outliers_keep_the_bulk:
// bulk error 1.2e-5 / 1.0e-4; now fp32 class).  The matrix stage multiplies l' by the partner's high piece times 2^-11, made in
// registers (v_pk_mul_f16, exact for partners >= 2^-3 in scaled units): still three MFMAs into one accumulator.
constexpr float kWbfLoScale = 2048.f;
__device__ __forceinline__ void wbf_split2hs_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
  f32x2 x = {x0, x1};
  const wbf_f16x2 h = __builtin_convertvector(x, wbf_f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  const f32x2 hf = __builtin_convertvector(h, f32x2);
  f32x2 r = {(x0 - hf.x) * kWbfLoScale, (x1 - hf.y) * kWbfLoScale};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, wbf_f16x2));
}
typedef _Float16 wbf_f16x8 __attribute__((ext_vector_type(8)));
// high piece of the partner operand times 2^-11 (for the product with a scaled low piece)
__device__ __forceinline__ uint4 wbf_hi_down(uint4 h) {
  const wbf_f16x8 v = __builtin_bit_cast(wbf_f16x8, h) * (_Float16)(1.0f / kWbfLoScale);
  return __builtin_bit_cast(uint4, v);
}
// A caller-owned transformed-input buffer (msk_conv3d_fwd_ex's xform) starts with a header of kWbfXformHeader bytes whose
// two amax arrays hold the max |x| the transform was scaled by and the max |w| of the layer's weights (NP = 2; unused
// otherwise): the weight gradient that consumes the buffer later undoes the first, the data gradient of the same step reuses
// the second (msk_conv3d_bwd_bnact: the weights have not changed since the forward pass).
constexpr int kWbfXformHeader = 512;  // two amax arrays: max |x| and max |w| the forward pass scaled by
// Power-of-two scale of a tensor from (an upper bound of) its max |value| on the DEVICE (no host round trip): amax * scale
// lands in [2^9, 2^10), the Winograd transforms amplify by <= 21.25, fp16 overflows at 65504 = 2^16.  NULL / 0 / inf -> 1.
// An "amax" is an array of kWbfAmaxWays floats whose maximum counts: producers spread their atomics over the ways by block
// index (a thousand blocks updating ONE address cost as much as the pass they ride on), consumers take the max of all.
constexpr int kWbfAmaxWays = 64;
__device__ __forceinline__ float wbf_amax_of(const float* amax) {
  const float4* p = reinterpret_cast<const float4*>(amax);
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < kWbfAmaxWays / 4; ++i) {
    const float4 q = p[i];
    m = fmaxf(fmaxf(m, fmaxf(q.x, q.y)), fmaxf(q.z, q.w));
  }
  return m;
}
__device__ __forceinline__ float wbf_scale_from(float a) {
  if (!(a > 0.f) || !(a < 3.0e38f)) return 1.f;
  int e;
  (void)frexpf(a, &e);  // a = m * 2^e, m in [0.5, 1)
  return ldexpf(1.f, 10 - e);
}
__device__ __forceinline__ float wbf_scale_of(const float* amax) {
  if (!amax) return 1.f;
  return wbf_scale_from(wbf_amax_of(amax));
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
  int lane_map;
  const float* amax;  // NP = 2: device scalar, (bound of) max |value| of the source tensor -> wbf_scale_of; NULL = unscaled
  float* amax_copy;   // non-null: block 0 copies the amax array there (the header of a kept transform) -- no memcpy command
  int t_per;          // W tiles per workgroup (grid.z = ceil(T / t_per) chunks; msk_wbf_transform* fill it): planes of few positions and many
                      // tiles (MRI level 256 x 256 x 9 tiled along 256: 82 workgroups of 64 tiles) otherwise leave most of the chip idle
  int c_real;         // > 0: only the first c_real channels of a source voxel exist (c_real % 4 == 0), the rest of the CK are zeros -- the
                      // zero-padded problems of msk_conv.hip (20-class heads) without a padded copy of the tensor; 0 = all CK
  float* cmax;        // MODE 1, NP = 2, non-null: zeroed array [CK]; max |dy| PER CHANNEL is folded into it (wbf_cmax_commit) for the
                      // weight gradient's per-channel renormalisation (msk_wgrad_wbf.hip)
};
// Per-channel maxima of the GRADIENT tensor dy, measured by the kernel that writes its transform Y = A dy (round 4).  The
// weight-gradient kernel multiplies both fp16 pieces of output channel c by 2^s(c), s(c) = floor(log2(max |dy| / max |dy[c]|))
// in [0, 15], before the matrix instructions and divides its sums by 2^s(c): every channel then sits as high in fp16's range
// as the tensor's loudest one, and the in-register factors 2^-11 of the cross terms (both applied on this side) stay exact for
// channels ANY factor below the loudest (round 3: exact only within 2^13, one bit lost per factor of two beyond).
__device__ __forceinline__ int wbf_chan_shift(float tensor_amax, float chan_max) {
  if (!(chan_max > 0.f) || !(tensor_amax > 0.f) || !(tensor_amax < 3.0e38f)) return 0;
  int ea, ec;
  (void)frexpf(tensor_amax, &ea);
  (void)frexpf(chan_max, &ec);
  const int s = ea - ec;        // chan_max * 2^s < 2^ea: below the tensor's power-of-two ceiling
  return s < 0 ? 0 : (s > 15 ? 15 : s);
}
// mx[j] = this thread's max |value| of channel cg*8 + j (0 for idle threads).  Every lane of the wavefront must call it.
// wave_cg: all 64 lanes share cg (one channel group per wavefront), else cg varies with lane & 3.
#ifndef WBF_CMAX_PRECHECK
#define WBF_CMAX_PRECHECK 1   // 0: A/B only (tools/ab_build.sh ... -DWBF_CMAX_PRECHECK=0)
#endif
__device__ __forceinline__ void wbf_cmax_commit(float* cmax, int cg, float (&mx)[8], bool wave_cg) {
  unsigned* q = reinterpret_cast<unsigned*>(cmax) + cg * 8;
#if WBF_CMAX_PRECHECK
  // (only launches of more workgroups than are resident together: in a single round every workgroup reads the initial zeros and
  // the reads are pure latency at the kernel's tail -- 128^3 batch 2: 546 workgroups, +0.05 ms per step with the reads)
  const bool precheck = gridDim.x * gridDim.y * gridDim.z > 1024u;
  unsigned cur[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};   // requested before the wavefront reduction below, which covers part of their latency
  if (precheck) {
#pragma unroll
    for (int j = 0; j < 8; ++j) cur[j] = __hip_atomic_load(q + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#endif
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float m = mx[j];
#pragma unroll
    for (int o = 32; o >= 4; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (wave_cg) {
      m = fmaxf(m, __shfl_xor(m, 2, 64));
      m = fmaxf(m, __shfl_xor(m, 1, 64));
    }
    mx[j] = m;
  }
  const int lane = threadIdx.x & 63;
  if (lane < (wave_cg ? 1 : 4)) {
    // the 8 C words of a tensor sit in one or two cache lines = one or two L2 channels, and atomics on a line are served one
    // after the other (measured: ~4 ns each; 4160 workgroups of the 512 x 512 x 12 slab = 133 k atomics = 0.5 of the 0.79 ms of
    // its A dy transform).  The maximum only grows: relaxed reads first (all 8 in flight together, above), the atomic only where this
    // wavefront would raise the value -- after the first few workgroups almost none does.  (A stale read costs one needless
    // atomic, never a missed one.)
#if WBF_CMAX_PRECHECK
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (mx[j] > 0.f && __float_as_uint(mx[j]) > cur[j]) atomicMax(q + j, __float_as_uint(mx[j]));  // non-negative floats order like their bits
#else
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (mx[j] > 0.f) atomicMax(q + j, __float_as_uint(mx[j]));
#endif
  }
}
// K = 5 | 3 (Winograd F(4,5) / F(4,3)); NP = 3 (exact bf16 split) | 1 (fp16 operands, K = 3 only)
int msk_wbf_transform(msk_ctx* ctx, int mode, int K, int NP, const WbfTinArgs& a);

// Backward of conv -> BatchNorm (batch statistics) -> PReLU fused into the transforms of dy (msk_conv3d_bwd_bnact):
// dy = scale * (du - sums[c]/M - xhat * sums[C + c]/M), du = dout * (u > 0 ? 1 : alpha), u = scale*y + shift,
// xhat = (y - mean) * invstd  -- the arithmetic of msk_affine_act_bwd_apply (bn_mode 1, no residual) -- is evaluated in the
// registers of ONE kernel that writes both transforms of dy, B^T dy (data gradient) and A dy (weight gradient): dy itself
// never reaches HBM.
struct WbfBnBwd {
  const float* y;     // convolution output (pre-BatchNorm), voxel stride yld
  int yld;
  const float* dout;  // gradient w.r.t. the unit's output, voxel stride dld
  int dld;
  const float *scale, *shift, *alpha, *mean, *invstd, *sums;  // [C] each, sums [2C]
  float invM;
  char* Y;            // second output: the A dy transform in the layout of the first (null: not written)
  const float* amax;  // NP = 2: amax array bounding max |dy| (msk_bn_bwd_bound, or written by the dual kernel: below), else NULL
  long y_xi;
  // NP = 2, one-kernel form: the bound  max|scale| * (max|du| + max|sums[c]/M| + max|xhat| * max|sums[C+c]/M|)  is evaluated
  // by the dual transform itself from the two amax arrays the reduce pass left (every block the same arithmetic on the same
  // inputs; block 0 stores it to `amax` for the kernels behind it) -- no separate bound kernel on the critical path
  const float* maxes;
  float* y_cmax;      // NP = 2: zeroed array [C] that receives max |dy| per channel (WbfTinArgs::cmax), or null
  int coef_stride;    // InstanceNorm (msk_conv3d_bwd_inact): sample n's scale / shift / mean / invstd lie n * coef_stride floats further,
  int sums_stride;    // its sums n * sums_stride floats further; 0 = one set for the batch (BatchNorm)
  // msk_wgrad_c1 only: the unit's pre-activation also adds its (one-channel, tiled) INPUT -- in_tr, vnet.py:75-78: the weight
  // gradient's A operand itself, taken from the kernel's LDS halo
  int res_is_input;
};
// writes B^T dy to a.V when write_v and A dy to bn.Y when that is non-null
int msk_wbf_transform_dual(msk_ctx* ctx, int K, int NP, const WbfTinArgs& a, const WbfBnBwd& bn, bool write_v);
// pieces per value for a K^3 convolution under the context's precision option ("conv_fp16": fp16 operands for K = 3)
// and "conv_split" 2: the fp16 two-piece split above)
inline int wbf_pieces(const msk_ctx* ctx, int K) { return (K == 3 && ctx->conv_fp16) ? 1 : (ctx->conv_split == 2 ? 2 : 3); }

// Geometry shared by the forward / data-gradient pipeline and the weight gradient (so that V = B^T x written by the forward
// pass can be handed to the weight gradient, msk_conv3d_fwd_ex / msk_conv3d_wgrad_ex): which tensor axes play the
// logical (d, h, w) roles -- the transform runs along w in T = ceil(w / 4) tiles; (d, h) carry the position tiles -- and the padded
// plane dims of the transformed tensor.  Depends on the spatial dims only.
struct WbfGeom {
  int perm[3];        // logical (d, h, w) <- tensor axis
  int LD, LH, LW, T;
  int DP, HP;         // plane dims: tile-rounded + 4 halo slots
  double max_pad;     // the padding limit this geometry was accepted under (kWbfMaxPad, or kWbfMaxPadLast on the second try)
};
// (minTD, minTH): the coarsest position tile a consumer of this geometry needs -- a function of the layer's OUTPUT channel
// count (wbf_min_tile), so that the forward pass and the weight gradient of one layer agree.
inline void wbf_min_tile(int cout, int* td, int* th) {
  *td = cout == 32 ? 16 : 8;
  *th = cout <= 64 ? 16 : 8;
}
// Padding of the (d, h) position planes up to whole tiles that is still worth it: the 16-bit pipeline is 4-5x faster than the
// fp32 Winograd / direct kernels a declined layer falls back to, so up to 80 % padded matrix work wins clearly (MRI level
// 256 x 256 x 9: 9 -> 16 planes, 12.9 -> ~5 ms for its three kernels; round 2) and even the 4x of the 2-voxel-deep MRI bottom
// level (256 channels at 32 x 32 x 2: planes 32 x 2 -> 32 x 8) still does (round 4, A/B on one box: step 30.8 -> 30.0 ms;
// 35 % was the round-1 break-even against the bf16x3 pipeline).
#ifndef WBF_RAGGED_W
#define WBF_RAGGED_W 1   // 0 = A/B: the transform axis must be a multiple of 4 (rounds 1-3)
#endif
constexpr bool kWbfRaggedW = WBF_RAGGED_W != 0;
constexpr double kWbfMaxPad = 1.8;       // the limit every shape is tried with first
constexpr double kWbfMaxPadLast = 4.0;   // second try for shapes nothing accepted (the geometry remembers which limit it was made with)
inline bool wbf_pick_geom(int D, int H, int W, int minTD, int minTH, WbfGeom* out) {
  static const int kPerms[6][3] = {{0, 1, 2}, {1, 0, 2}, {0, 2, 1}, {2, 0, 1}, {1, 2, 0}, {2, 1, 0}};
  const int dims[3] = {D, H, W};
  int best = -1;
  double best_cost = 0, limit = kWbfMaxPad;
  for (int pass = 0; pass < 2 && best < 0; ++pass) {
    limit = pass == 0 ? kWbfMaxPad : kWbfMaxPadLast;
    for (int i = 0; i < 6; ++i) {
      const int ld = dims[kPerms[i][0]], lh = dims[kPerms[i][1]], lw = dims[kPerms[i][2]];
      // the transform axis may be RAGGED (round 4): its last tile of 4 outputs is then partly outside the volume -- the transform
      // kernels read zeros there and the output transforms store and count only what exists -- and the padded matrix work of
      // that tile enters limit and cost like the position tiles' (MRI level 256 x 256 x 9: transform along 9 -> 12 with whole
      // 256 x 256 planes = 1.33x instead of planes 256 x 9 -> 16 = 1.78x; bottom level 32 x 32 x 2: 2x instead of 4x)
      const int lw4 = (lw + 3) / 4 * 4;
      if (lw % 4 && !kWbfRaggedW) continue;
      if ((double)((ld + minTD - 1) / minTD * minTD) * ((lh + minTH - 1) / minTH * minTH) * lw4 > limit * (double)ld * lh * lw) continue;
      const double cost = (double)((ld + 7) / 8 * 8) * ((lh + 7) / 8 * 8) * lw4 / ((double)ld * lh * lw);  // padding at the finest tile
      if (best < 0 || cost < best_cost - 1e-9) {
        best = i;
        best_cost = cost;
      }
    }
  }
  if (best < 0) return false;
  for (int j = 0; j < 3; ++j) out->perm[j] = kPerms[best][j];
  out->LD = dims[out->perm[0]];
  out->LH = dims[out->perm[1]];
  out->LW = dims[out->perm[2]];
  out->T = (out->LW + 3) / 4;
  out->max_pad = limit;
  const int rd = out->LD >= 16 ? 16 : 8, rh = out->LH >= 32 ? 32 : (out->LH >= 16 ? 16 : 8);
  out->DP = (out->LD + rd - 1) / rd * rd + 4;
  out->HP = (out->LH + rh - 1) / rh * rh + 4;
  return true;
}
// a (TD x TH) position tiling fits the planes and wastes no more of the matrix work on padding than the geometry's limit
inline bool wbf_tile_ok(const WbfGeom& g, int TD, int TH) {
  const int td = (g.LD + TD - 1) / TD * TD, th = (g.LH + TH - 1) / TH * TH;
  // the same inequality wbf_pick_geom accepted the geometry under: the ragged last W tile counts as padded work too (advisor,
  // round 4: without the LW factor a ragged axis let a tiling through at max_pad * LW4 / LW of padded work)
  const int lw4 = (g.LW + 3) / 4 * 4;
  return td + 4 <= g.DP && th + 4 <= g.HP && (double)td * th * lw4 <= g.max_pad * (double)g.LD * g.LH * g.LW;
}
size_t msk_wbf_xform_bytes(int n, int d, int h, int w, int c, int cout, int K, int NP);
size_t msk_wbf_fwd_xform_bytes(const msk_ctx* ctx, int n, int d, int h, int w, int c, int cout, int K);
// merge of per-block BatchNorm partial records [nb][C][3] = (n, mean, M2) into stats[2C] (msk_elementwise.hip)
int msk_bn_stats_merge(msk_ctx* ctx, const float* partial, int nb, int C, float* stats, const msk_bn_fin* fin = nullptr);  // fin: + msk_bn_finalize(world 1) in the same launch

