// Volumetric preprocessing on gfx950: resample (scipy.ndimage.zoom order 0/1
// semantics), HU / min-max / max normalisation, label remap.  Replaces the
// numpy|CuPy code of tools/preprocess_utils/{geometry,values}.py.  All HBM-bound:
// one coalesced read stream + one write stream, no LDS.
#include <type_traits>

#include "msk_common.h"

namespace {

constexpr int kThreads = 256;

inline int ew_blocks(size_t total, int num_cu) {
  size_t b = (total + kThreads - 1) / kThreads;
  size_t cap = (size_t)num_cu * 16;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

// coordinate of output index o along an axis: o * (n_in-1)/(n_out-1)  
__device__ __forceinline__ double axis_coord(int o, int n_in, int n_out) {
  return n_out > 1 ? (double)o * ((double)(n_in - 1) / (double)(n_out - 1)) : 0.0;
}

template <typename T, int ORDER>
__global__ void __launch_bounds__(kThreads)
resample_k(const T* __restrict__ src_full, int fh, int fw, int ci, int cj, int ck, int sd, int sh, int sw,
           T* __restrict__ dst, int dd, int dh, int dw) {
  // the source is the crop box [ci, ci+sd) x [cj, cj+sh) x [ck, ck+sw) of a volume with row pitches (fh, fw):
  // functional.py:103-110 resized_crop_3d = crop_3d + resize_3d; the plain resample is the full box
  const size_t total = (size_t)dd * dh * dw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ow = (int)(i % dw);
    const int oh = (int)((i / dw) % dh);
    const int od = (int)(i / ((size_t)dw * dh));
    const double cd = axis_coord(od, sd, dd), ch = axis_coord(oh, sh, dh), cw = axis_coord(ow, sw, dw);
    if constexpr (ORDER == 0) {
      int id = (int)floor(cd + 0.5), ih = (int)floor(ch + 0.5), iw = (int)floor(cw + 0.5);
      id = min(max(id, 0), sd - 1);
      ih = min(max(ih, 0), sh - 1);
      iw = min(max(iw, 0), sw - 1);
      dst[i] = src_full[((size_t)(id + ci) * fh + (ih + cj)) * fw + (iw + ck)];
    } else {
      int d0 = min(max((int)floor(cd), 0), sd - 1), h0 = min(max((int)floor(ch), 0), sh - 1),
          w0 = min(max((int)floor(cw), 0), sw - 1);
      const int d1 = min(d0 + 1, sd - 1), h1 = min(h0 + 1, sh - 1), w1 = min(w0 + 1, sw - 1);
      const double td = cd - d0, th = ch - h0, tw = cw - w0;
      auto at = [&](int a, int b, int c) { return (double)src_full[((size_t)(a + ci) * fh + (b + cj)) * fw + (c + ck)]; };
      // separable: d, then h, then w (matches oracle/preprocess_numpy.py)
      const double v00 = at(d0, h0, w0) * (1.0 - td) + at(d1, h0, w0) * td;
      const double v01 = at(d0, h0, w1) * (1.0 - td) + at(d1, h0, w1) * td;
      const double v10 = at(d0, h1, w0) * (1.0 - td) + at(d1, h1, w0) * td;
      const double v11 = at(d0, h1, w1) * (1.0 - td) + at(d1, h1, w1) * td;
      const double v0 = v00 * (1.0 - th) + v10 * th;
      const double v1 = v01 * (1.0 - th) + v11 * th;
      const double v = v0 * (1.0 - tw) + v1 * tw;
      if constexpr (!std::is_same<T, float>::value) {
        dst[i] = (T)rint(v);
      } else {
        dst[i] = (T)v;
      }
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
flip3d_k(const T* __restrict__ src, T* __restrict__ dst, int d, int h, int w, int axis) {
  const size_t total = (size_t)d * h * w;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int ow = (int)(i % w), oh = (int)((i / w) % h), od = (int)(i / ((size_t)w * h));
    if (axis == 0) od = d - 1 - od;
    else if (axis == 1) oh = h - 1 - oh;
    else ow = w - 1 - ow;
    dst[i] = src[((size_t)od * h + oh) * w + ow];
  }
}

// scipy.ndimage.rotate(order 0|1, mode='constant', reshape=False) on the (a0, a1) plane:
// in = M @ out + shift in double like scipy's NI_GeometricTransform (no FMA contraction so that
// right-angle rotations stay on the grid); coordinates outside [0, n-1] -> cval, no interpolation
// beyond the edges; integer outputs round half away from zero.
template <typename T, int ORDER>
__global__ void __launch_bounds__(kThreads)
rotate3d_k(const T* __restrict__ src, T* __restrict__ dst, int d, int h, int w, int a0, int a1, double m00, double m01,
           double m10, double m11, double s0, double s1, double cval) {
  const size_t total = (size_t)d * h * w;
  const int dims[3] = {d, h, w};
  const int n0 = dims[a0], n1 = dims[a1];
  const size_t strides[3] = {(size_t)h * w, (size_t)w, 1};
  const size_t st0 = strides[a0], st1 = strides[a1];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int idx[3];
    idx[2] = (int)(i % w);
    idx[1] = (int)((i / w) % h);
    idx[0] = (int)(i / ((size_t)w * h));
    const double o0 = (double)idx[a0], o1 = (double)idx[a1];
    const double c0 = __dadd_rn(__dadd_rn(s0, __dmul_rn(o0, m00)), __dmul_rn(o1, m01));
    const double c1 = __dadd_rn(__dadd_rn(s1, __dmul_rn(o0, m10)), __dmul_rn(o1, m11));
    double v = cval;
    if (!(c0 < 0.0 || c0 > (double)(n0 - 1) || c1 < 0.0 || c1 > (double)(n1 - 1))) {
      const size_t base = i - (size_t)idx[a0] * st0 - (size_t)idx[a1] * st1;  // the other axis' offset
      if constexpr (ORDER == 0) {
        const int i0 = min(max((int)floor(c0 + 0.5), 0), n0 - 1), i1 = min(max((int)floor(c1 + 0.5), 0), n1 - 1);
        v = (double)src[base + i0 * st0 + i1 * st1];
      } else {
        const int f0 = (int)floor(c0), f1 = (int)floor(c1);
        const double t0 = c0 - f0, t1 = c1 - f1;
        const int g0 = min(f0 + 1, n0 - 1), g1 = min(f1 + 1, n1 - 1);  // weight 0 when clamped
        auto at = [&](int p, int q) { return (double)src[base + p * st0 + q * st1]; };
        v = __dmul_rn(at(f0, f1), __dmul_rn(1.0 - t0, 1.0 - t1));
        v = __dadd_rn(v, __dmul_rn(at(f0, g1), __dmul_rn(1.0 - t0, t1)));
        v = __dadd_rn(v, __dmul_rn(at(g0, f1), __dmul_rn(t0, 1.0 - t1)));
        v = __dadd_rn(v, __dmul_rn(at(g0, g1), __dmul_rn(t0, t1)));
      }
    }
    if constexpr (!std::is_same<T, float>::value) {
      dst[i] = (T)(v > 0.0 ? floor(v + 0.5) : ceil(v - 0.5));
    } else {
      dst[i] = (T)v;
    }
  }
}

__global__ void __launch_bounds__(kThreads)
hu_norm_k(const float* __restrict__ src, float* __restrict__ dst, size_t n, float hu_min, float scale, float hu_nan) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = src[i];
    if (v != v) v = hu_nan;            // np.nan_to_num(nan=HU_nan); +-inf -> clipped below like numpy's huge finite values
    v = (v - hu_min) / scale;          // (image - HU_min) / ((HU_max - HU_min) / 255)
    v = fminf(fmaxf(v, 0.f), 255.f);   // np.clip(image, 0, 255)
    dst[i] = v;
  }
}

__global__ void __launch_bounds__(kThreads)
minmax_partial_k(const float* __restrict__ src, size_t n, float* __restrict__ partial /*[nb][2]*/) {
  __shared__ float smin[kThreads / 64], smax[kThreads / 64];
  float lo = INFINITY, hi = -INFINITY;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = src[i];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_down(lo, o, 64));
    hi = fmaxf(hi, __shfl_down(hi, o, 64));
  }
  const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
  if (lane == 0) {
    smin[wave] = lo;
    smax[wave] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kThreads / 64; ++w) {
      lo = fminf(lo, smin[w]);
      hi = fmaxf(hi, smax[w]);
    }
    partial[2 * blockIdx.x] = lo;
    partial[2 * blockIdx.x + 1] = hi;
  }
}
__global__ void minmax_final_k(float* partial, int nb) {
  // one wavefront: the lanes stride over the block partials, then a shuffle tree (min / max are exact in any order).  Round 4 had
  // ONE thread walk up to 1024 partials -- a chain of dependent loads: 55 of msk_max_norm's 75 us at 512 x 512 x 12
  float lo = INFINITY, hi = -INFINITY;
  for (int b = threadIdx.x; b < nb; b += 64) {
    lo = fminf(lo, partial[2 * b]);
    hi = fmaxf(hi, partial[2 * b + 1]);
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_down(lo, o, 64));
    hi = fmaxf(hi, __shfl_down(hi, o, 64));
  }
  if (threadIdx.x == 0) {
    partial[0] = lo;
    partial[1] = hi;
  }
}

// mode 0: (x - lo)/(hi - lo) clipped to [0,1] with lo/hi from `mm` (device) or the given bounds
// mode 1: x / hi when hi > 0 (transform.py:67-69)
__global__ void __launch_bounds__(kThreads)
norm_apply_k(const float* __restrict__ src, float* __restrict__ dst, size_t n, const float* __restrict__ mm,
             float blo, float bhi, int mode) {
  const float lo = mm ? mm[0] : blo, hi = mm ? mm[1] : bhi;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = src[i];
    if (mode == 0) {
      v = (v - lo) / (hi - lo);
      v = fminf(fmaxf(v, 0.f), 1.f);
    } else {
      if (hi > 0.f) v = v / hi;
    }
    dst[i] = v;
  }
}

__global__ void __launch_bounds__(kThreads)
label_remap_k(int32_t* __restrict__ label, size_t n, const int32_t* __restrict__ keys,
              const int32_t* __restrict__ vals, int npairs) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int32_t v = label[i];
    for (int k = 0; k < npairs; ++k)
      if (v == keys[k]) v = vals[k];  // sequential passes, like the reference's loop over map_dict
    label[i] = v;
  }
}

}  // namespace

extern "C" {

static int crop_resample(msk_ctx* ctx, const void* src, int fd, int fh, int fw, int ci, int cj, int ck, int sd, int sh,
                         int sw, void* dst, int dd, int dh, int dw, int order, int dtype) {
  MSK_REQUIRE(ctx, order == 0 || order == 1, "order must be 0 or 1");
  MSK_REQUIRE(ctx, dtype == 0 || dtype == 1, "dtype must be 0 (float32) or 1 (int32)");
  MSK_REQUIRE(ctx, sd > 0 && sh > 0 && sw > 0 && dd > 0 && dh > 0 && dw > 0, "empty volume");
  MSK_REQUIRE(ctx, ci >= 0 && cj >= 0 && ck >= 0 && ci + sd <= fd && cj + sh <= fh && ck + sw <= fw,
              "crop box outside the volume");
  const size_t total = (size_t)dd * dh * dw;
  const int nb = ew_blocks(total, ctx->num_cu);
  msk_launch_scope ls(ctx, order == 0 ? "resample3d_order0" : "resample3d_order1");
  if (dtype == 0) {
    if (order == 0)
      hipLaunchKernelGGL((resample_k<float, 0>), dim3(nb), dim3(kThreads), 0, ctx->stream, (const float*)src, fh, fw, ci, cj, ck, sd, sh, sw, (float*)dst, dd, dh, dw);
    else
      hipLaunchKernelGGL((resample_k<float, 1>), dim3(nb), dim3(kThreads), 0, ctx->stream, (const float*)src, fh, fw, ci, cj, ck, sd, sh, sw, (float*)dst, dd, dh, dw);
  } else {
    if (order == 0)
      hipLaunchKernelGGL((resample_k<int32_t, 0>), dim3(nb), dim3(kThreads), 0, ctx->stream, (const int32_t*)src, fh, fw, ci, cj, ck, sd, sh, sw, (int32_t*)dst, dd, dh, dw);
    else
      hipLaunchKernelGGL((resample_k<int32_t, 1>), dim3(nb), dim3(kThreads), 0, ctx->stream, (const int32_t*)src, fh, fw, ci, cj, ck, sd, sh, sw, (int32_t*)dst, dd, dh, dw);
  }
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

int msk_resample3d(msk_ctx* ctx, const void* src, int sd, int sh, int sw, void* dst, int dd, int dh, int dw,
                   int order, int dtype) {
  return crop_resample(ctx, src, sd, sh, sw, 0, 0, 0, sd, sh, sw, dst, dd, dh, dw, order, dtype);
}

int msk_crop_resample3d(msk_ctx* ctx, const void* src, int sd, int sh, int sw, int i, int j, int k, int cd, int ch,
                        int cw, void* dst, int dd, int dh, int dw, int order, int dtype) {
  return crop_resample(ctx, src, sd, sh, sw, i, j, k, cd, ch, cw, dst, dd, dh, dw, order, dtype);
}

int msk_flip3d(msk_ctx* ctx, const void* src, void* dst, int d, int h, int w, int axis, int dtype) {
  MSK_REQUIRE(ctx, axis >= 0 && axis <= 2, "axis must be 0, 1 or 2");
  MSK_REQUIRE(ctx, dtype == 0 || dtype == 1, "dtype must be 0 (float32) or 1 (int32)");
  MSK_REQUIRE(ctx, src != dst, "flip is out of place");
  const size_t total = (size_t)d * h * w;
  if (total == 0) return 0;
  msk_launch_scope ls(ctx, "flip3d");
  // float32 and int32 are both 4-byte moves
  hipLaunchKernelGGL((flip3d_k<uint32_t>), dim3(ew_blocks(total, ctx->num_cu)), dim3(kThreads), 0, ctx->stream,
                     (const uint32_t*)src, (uint32_t*)dst, d, h, w, axis);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

int msk_rotate3d(msk_ctx* ctx, const void* src, void* dst, int d, int h, int w, int axis_a, int axis_b, double angle_deg,
                 int order, double cval, int dtype) {
  MSK_REQUIRE(ctx, order == 0 || order == 1, "order must be 0 or 1");
  MSK_REQUIRE(ctx, dtype == 0 || dtype == 1, "dtype must be 0 (float32) or 1 (int32)");
  MSK_REQUIRE(ctx, axis_a >= 0 && axis_a <= 2 && axis_b >= 0 && axis_b <= 2 && axis_a != axis_b, "bad rotation plane");
  MSK_REQUIRE(ctx, src != dst, "rotate is out of place");
  MSK_REQUIRE(ctx, d > 0 && h > 0 && w > 0, "empty volume");
  const int a0 = axis_a < axis_b ? axis_a : axis_b, a1 = axis_a < axis_b ? axis_b : axis_a;  // scipy sorts the axes
  // scipy.special.cosdg / sindg: exact at multiples of 90 degrees
  double c, s;
  double am = fmod(angle_deg, 360.0);
  if (am < 0) am += 360.0;
  if (fmod(am, 90.0) == 0.0) {
    static const double kc[4] = {1.0, 0.0, -1.0, 0.0}, ks[4] = {0.0, 1.0, 0.0, -1.0};
    const int q = ((int)(am / 90.0)) & 3;
    c = kc[q];
    s = ks[q];
  } else {
    const double r = angle_deg * (M_PI / 180.0);
    c = cos(r);
    s = sin(r);
  }
  const int dims[3] = {d, h, w};
  const double c0 = (dims[a0] - 1) / 2.0, c1 = (dims[a1] - 1) / 2.0;
  const double m00 = c, m01 = s, m10 = -s, m11 = c;
  const double s0 = c0 - (m00 * c0 + m01 * c1), s1 = c1 - (m10 * c0 + m11 * c1);
  const size_t total = (size_t)d * h * w;
  const int nb = ew_blocks(total, ctx->num_cu);
  msk_launch_scope ls(ctx, "rotate3d");
  if (dtype == 0) {
    if (order == 0)
      hipLaunchKernelGGL((rotate3d_k<float, 0>), dim3(nb), dim3(kThreads), 0, ctx->stream, (const float*)src, (float*)dst, d, h, w, a0, a1, m00, m01, m10, m11, s0, s1, cval);
    else
      hipLaunchKernelGGL((rotate3d_k<float, 1>), dim3(nb), dim3(kThreads), 0, ctx->stream, (const float*)src, (float*)dst, d, h, w, a0, a1, m00, m01, m10, m11, s0, s1, cval);
  } else {
    if (order == 0)
      hipLaunchKernelGGL((rotate3d_k<int32_t, 0>), dim3(nb), dim3(kThreads), 0, ctx->stream, (const int32_t*)src, (int32_t*)dst, d, h, w, a0, a1, m00, m01, m10, m11, s0, s1, cval);
    else
      hipLaunchKernelGGL((rotate3d_k<int32_t, 1>), dim3(nb), dim3(kThreads), 0, ctx->stream, (const int32_t*)src, (int32_t*)dst, d, h, w, a0, a1, m00, m01, m10, m11, s0, s1, cval);
  }
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

int msk_hu_norm(msk_ctx* ctx, const float* src, float* dst, size_t count, float hu_min, float hu_max, float hu_nan) {
  if (count == 0) return 0;
  const float scale = (float)(((double)hu_max - (double)hu_min) / 255.0);
  msk_launch_scope ls(ctx, "hu_norm");
  hipLaunchKernelGGL(hu_norm_k, dim3(ew_blocks(count, ctx->num_cu)), dim3(kThreads), 0, ctx->stream, src, dst, count, hu_min, scale, hu_nan);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

static int minmax(msk_ctx* ctx, const float* src, size_t count, float** mm) {
  int nb = ew_blocks(count, ctx->num_cu);
  if (nb > 1024) nb = 1024;
  float* partial = (float*)msk_workspace(ctx, (size_t)nb * 2 * sizeof(float));
  if (!partial) return -1;
  {
    msk_launch_scope ls(ctx, "minmax_partial");
    hipLaunchKernelGGL(minmax_partial_k, dim3(nb), dim3(kThreads), 0, ctx->stream, src, count, partial);
    MSK_LAUNCH_CHECK(ctx);
  }
  {
    msk_launch_scope ls(ctx, "minmax_final");
    hipLaunchKernelGGL(minmax_final_k, dim3(1), dim3(64), 0, ctx->stream, partial, nb);
    MSK_LAUNCH_CHECK(ctx);
  }
  *mm = partial;
  return 0;
}

int msk_minmax_norm(msk_ctx* ctx, const float* src, float* dst, size_t count, int use_bounds, float min_val, float max_val) {
  if (count == 0) return 0;
  float* mm = nullptr;
  if (!use_bounds) {
    if (minmax(ctx, src, count, &mm) != 0) return -1;
  }
  msk_launch_scope ls(ctx, "minmax_norm");
  hipLaunchKernelGGL(norm_apply_k, dim3(ew_blocks(count, ctx->num_cu)), dim3(kThreads), 0, ctx->stream, src, dst, count, (const float*)mm, min_val, max_val, 0);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

int msk_max_norm(msk_ctx* ctx, const float* src, float* dst, size_t count) {
  if (count == 0) return 0;
  float* mm = nullptr;
  if (minmax(ctx, src, count, &mm) != 0) return -1;
  msk_launch_scope ls(ctx, "max_norm");
  hipLaunchKernelGGL(norm_apply_k, dim3(ew_blocks(count, ctx->num_cu)), dim3(kThreads), 0, ctx->stream, src, dst, count, (const float*)mm, 0.f, 0.f, 1);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

int msk_label_remap(msk_ctx* ctx, int32_t* label, size_t count, const int32_t* keys, const int32_t* vals, int npairs) {
  if (count == 0 || npairs == 0) return 0;
  msk_launch_scope ls(ctx, "label_remap");
  hipLaunchKernelGGL(label_remap_k, dim3(ew_blocks(count, ctx->num_cu)), dim3(kThreads), 0, ctx->stream, label, count, keys, vals, npairs);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

}  // extern "C"
