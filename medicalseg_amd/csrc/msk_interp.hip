// Trilinear resize of the deep-supervision heads (reference vnet_deepsup.py:268-277:
// F.interpolate(d, size=x.shape[2:], mode='trilinear'), i.e. align_corners=False, align_mode=0
// [PADDLE]): per axis ratio = n_in/n_out, src = max(ratio*(o+0.5)-0.5, 0), i0 = floor(src),
// i1 = min(i0+1, n_in-1), lam = src-i0.  NDHWC fp32, few channels (ncls): HBM-bound on the
// full-resolution side.
//   forward : one pass, 8-corner gather (the coarse source stays in L2)
//   backward: the adjoint, separable -- one GATHER pass per resized axis (W, H, D), each source
//             element summing its pre-image in a fixed order: deterministic, no atomics
#include "msk_common.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ void axis_src(int o, float ratio, int n_in, int& i0, int& i1, float& lam) {
  const float src = fmaxf(ratio * ((float)o + 0.5f) - 0.5f, 0.f);
  i0 = min((int)src, n_in - 1);
  i1 = min(i0 + 1, n_in - 1);
  lam = src - (float)i0;
}

// V = 4: a thread owns four consecutive channels of an output voxel (float4 loads / stores; C % 4 == 0, aligned tensors):
// the index arithmetic and the eight corner addresses are shared by four values (20-class heads at 512 x 512 x 12: 0.50 ->
// ~0.2 ms).  `total` counts threads' work items = voxels * C / V.
template <int V>
__global__ void __launch_bounds__(kThreads)
interp_fwd_k(const float* __restrict__ src, int sld, int SD, int SH, int SW, float* __restrict__ dst, int dld, int DD,
             int DH, int DW, int C, long total, float rd, float rh, float rw) {
  const int CQ = C / V;
  for (long t = (long)blockIdx.x * kThreads + threadIdx.x; t < total; t += (long)gridDim.x * kThreads) {
    const int c = (int)(t % CQ) * V;
    long v = t / CQ;
    const int w = (int)(v % DW);
    v /= DW;
    const int h = (int)(v % DH);
    v /= DH;
    const int d = (int)(v % DD);
    const long n = v / DD;
    int d0, d1, h0, h1, w0, w1;
    float ld_, lh, lw;
    axis_src(d, rd, SD, d0, d1, ld_);
    axis_src(h, rh, SH, h0, h1, lh);
    axis_src(w, rw, SW, w0, w1, lw);
    const float* b = src + (long)n * SD * SH * SW * sld + c;
    float* o = dst + (t / CQ) * dld + c;
    if constexpr (V == 4) {
      auto at = [&](int dd, int hh, int ww) { return *reinterpret_cast<const float4*>(b + (((long)dd * SH + hh) * SW + ww) * sld); };
      auto mix = [](const float4 p, const float4 q, float l) {  // the scalar kernel's expression, per component
        return make_float4(p.x * (1.f - l) + q.x * l, p.y * (1.f - l) + q.y * l, p.z * (1.f - l) + q.z * l, p.w * (1.f - l) + q.w * l);
      };
      const float4 x00 = mix(at(d0, h0, w0), at(d0, h0, w1), lw), x01 = mix(at(d0, h1, w0), at(d0, h1, w1), lw);
      const float4 x10 = mix(at(d1, h0, w0), at(d1, h0, w1), lw), x11 = mix(at(d1, h1, w0), at(d1, h1, w1), lw);
      *reinterpret_cast<float4*>(o) = mix(mix(x00, x01, lh), mix(x10, x11, lh), ld_);
    } else {
      auto at = [&](int dd, int hh, int ww) { return b[(((long)dd * SH + hh) * SW + ww) * sld]; };
      const float x00 = at(d0, h0, w0) * (1.f - lw) + at(d0, h0, w1) * lw;
      const float x01 = at(d0, h1, w0) * (1.f - lw) + at(d0, h1, w1) * lw;
      const float x10 = at(d1, h0, w0) * (1.f - lw) + at(d1, h0, w1) * lw;
      const float x11 = at(d1, h1, w0) * (1.f - lw) + at(d1, h1, w1) * lw;
      const float y0 = x00 * (1.f - lh) + x01 * lh;
      const float y1 = x10 * (1.f - lh) + x11 * lh;
      *o = y0 * (1.f - ld_) + y1 * ld_;
    }
  }
}

// g: [outer][n_out][inner][C] (voxel stride gld)  ->  out: [outer][n_in][inner][C] (stride old_)
template <int V>
__global__ void __launch_bounds__(kThreads)
interp_axis_bwd_k(const float* __restrict__ g, int gld, float* __restrict__ out, int old_, long total, int n_out, int n_in,
                  long inner, int C, float ratio, int accumulate) {
  const float inv = 1.f / ratio;
  const int CQ = C / V;
  for (long t = (long)blockIdx.x * kThreads + threadIdx.x; t < total; t += (long)gridDim.x * kThreads) {
    const int c = (int)(t % CQ) * V;
    long v = t / CQ;
    const long iv = v % inner;
    v /= inner;
    const int s = (int)(v % n_in);
    const long o = v / n_in;
    // destination indices whose i0 or i1 can equal s: src(d) in (s-1, s+1), one slot of margin
    int lo = (int)floorf(((float)s - 0.5f) * inv - 0.5f) - 1;
    int hi = (int)ceilf(((float)s + 1.5f) * inv - 0.5f) + 1;
    lo = max(lo, 0);
    hi = min(hi, n_out - 1);
    const float* gp = g + ((long)o * n_out * inner + iv) * gld + c;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (int d = lo; d <= hi; ++d) {
      int i0, i1;
      float lam;
      axis_src(d, ratio, n_in, i0, i1, lam);
      float wgt = 0.f;
      if (i0 == s) wgt += 1.f - lam;
      if (i1 == s) wgt += lam;
      if (wgt != 0.f) {
        if constexpr (V == 4) {
          const float4 q = *reinterpret_cast<const float4*>(gp + (long)d * inner * gld);
          acc[0] = fmaf(wgt, q.x, acc[0]); acc[1] = fmaf(wgt, q.y, acc[1]); acc[2] = fmaf(wgt, q.z, acc[2]); acc[3] = fmaf(wgt, q.w, acc[3]);
        } else {
          acc[0] = fmaf(wgt, gp[(long)d * inner * gld], acc[0]);
        }
      }
    }
    float* op = out + (((long)o * n_in + s) * inner + iv) * old_ + c;
    if constexpr (V == 4) {
      float4 r = make_float4(acc[0], acc[1], acc[2], acc[3]);
      if (accumulate) {
        const float4 p = *reinterpret_cast<const float4*>(op);
        r.x += p.x; r.y += p.y; r.z += p.z; r.w += p.w;
      }
      *reinterpret_cast<float4*>(op) = r;
    } else {
      *op = accumulate ? *op + acc[0] : acc[0];
    }
  }
}

inline int blocks_for(long total, int num_cu) {
  long b = (total + kThreads - 1) / kThreads;
  const long cap = (long)num_cu * 16;
  if (b > cap) b = cap;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" {

int msk_interp_trilinear_fwd(msk_ctx* ctx, msk_tensor src, msk_tensor dst) {
  MSK_REQUIRE(ctx, src.p && dst.p, "null tensor");
  MSK_REQUIRE(ctx, src.n == dst.n && src.c == dst.c, "batch/channel mismatch");
  MSK_REQUIRE(ctx, src.d > 0 && src.h > 0 && src.w > 0 && dst.d > 0 && dst.h > 0 && dst.w > 0, "empty volume");
  const bool v4 = dst.c % 4 == 0 && src.ld % 4 == 0 && dst.ld % 4 == 0 && ((((uintptr_t)src.p) | ((uintptr_t)dst.p)) & 15) == 0;
  const long total = msk_voxels(dst) * dst.c / (v4 ? 4 : 1);
  msk_launch_scope ls(ctx, "interp_trilinear_fwd");
  if (v4)
    hipLaunchKernelGGL(interp_fwd_k<4>, dim3(blocks_for(total, ctx->num_cu)), dim3(kThreads), 0, ctx->stream,
                       (const float*)src.p, src.ld, src.d, src.h, src.w, (float*)dst.p, dst.ld, dst.d, dst.h, dst.w, dst.c,
                       total, (float)src.d / dst.d, (float)src.h / dst.h, (float)src.w / dst.w);
  else
    hipLaunchKernelGGL(interp_fwd_k<1>, dim3(blocks_for(total, ctx->num_cu)), dim3(kThreads), 0, ctx->stream,
                       (const float*)src.p, src.ld, src.d, src.h, src.w, (float*)dst.p, dst.ld, dst.d, dst.h, dst.w, dst.c,
                       total, (float)src.d / dst.d, (float)src.h / dst.h, (float)src.w / dst.w);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

int msk_interp_scratch_bytes(msk_ctx* ctx, msk_tensor src, msk_tensor dst, size_t* bytes) {
  MSK_REQUIRE(ctx, bytes != nullptr, "null out pointer");
  const size_t t1 = (size_t)dst.n * dst.d * dst.h * src.w * src.c;
  const size_t t2 = (size_t)dst.n * dst.d * src.h * src.w * src.c;
  *bytes = (t1 + t2) * sizeof(float);
  return 0;
}

int msk_interp_trilinear_bwd(msk_ctx* ctx, msk_tensor ddst, msk_tensor dsrc, int accumulate, void* scratch,
                             size_t scratch_bytes) {
  MSK_REQUIRE(ctx, ddst.p && dsrc.p, "null tensor");
  MSK_REQUIRE(ctx, dsrc.n == ddst.n && dsrc.c == ddst.c, "batch/channel mismatch");
  const int C = dsrc.c;
  size_t need = 0;
  msk_interp_scratch_bytes(ctx, dsrc, ddst, &need);
  const bool rw = ddst.w != dsrc.w, rh = ddst.h != dsrc.h;
  const bool need_scratch = rw || rh;  // the D pass always runs last and writes dsrc
  MSK_REQUIRE(ctx, !need_scratch || (scratch && scratch_bytes >= need), "scratch too small (msk_interp_scratch_bytes)");
  float* t1 = (float*)scratch;
  float* t2 = t1 + (size_t)ddst.n * ddst.d * ddst.h * dsrc.w * C;

  const float* cur = (const float*)ddst.p;
  int cur_ld = ddst.ld, cur_h = ddst.h, cur_w = ddst.w;
  const bool v4 = C % 4 == 0 && ddst.ld % 4 == 0 && dsrc.ld % 4 == 0 &&
                  ((((uintptr_t)ddst.p) | ((uintptr_t)dsrc.p) | ((uintptr_t)t1) | ((uintptr_t)t2)) & 15) == 0;
  const int VV = v4 ? 4 : 1;
#define MSK_AXIS_BWD(...)                                                                        \
  do {                                                                                           \
    if (v4) hipLaunchKernelGGL(interp_axis_bwd_k<4>, __VA_ARGS__);                               \
    else hipLaunchKernelGGL(interp_axis_bwd_k<1>, __VA_ARGS__);                                  \
  } while (0)
  msk_launch_scope ls(ctx, "interp_trilinear_bwd");
  if (rw) {  // [N*Dd*Hd][Wd][1][C] -> [N*Dd*Hd][Ws][1][C]
    const long total = (long)ddst.n * ddst.d * ddst.h * dsrc.w * C / VV;
    MSK_AXIS_BWD(dim3(blocks_for(total, ctx->num_cu)), dim3(kThreads), 0, ctx->stream, cur, cur_ld, t1, C, total, ddst.w, dsrc.w,
                 1L, C, (float)dsrc.w / ddst.w, 0);
    MSK_LAUNCH_CHECK(ctx);
    cur = t1; cur_ld = C; cur_w = dsrc.w;
  }
  if (rh) {  // [N*Dd][Hd][W][C] -> [N*Dd][Hs][W][C]
    const long total = (long)ddst.n * ddst.d * dsrc.h * cur_w * C / VV;
    MSK_AXIS_BWD(dim3(blocks_for(total, ctx->num_cu)), dim3(kThreads), 0, ctx->stream, cur, cur_ld, t2, C, total, ddst.h, dsrc.h,
                 (long)cur_w, C, (float)dsrc.h / ddst.h, 0);
    MSK_LAUNCH_CHECK(ctx);
    cur = t2; cur_ld = C; cur_h = dsrc.h;
  }
  {  // [N][Dd][H*W][C] -> [N][Ds][H*W][C] (identity map when Dd == Ds)
    const long total = (long)ddst.n * dsrc.d * cur_h * cur_w * C / VV;
    MSK_AXIS_BWD(dim3(blocks_for(total, ctx->num_cu)), dim3(kThreads), 0, ctx->stream, cur, cur_ld, (float*)dsrc.p, dsrc.ld, total,
                 ddst.d, dsrc.d, (long)cur_h * cur_w, C, (float)dsrc.d / ddst.d, accumulate);
    MSK_LAUNCH_CHECK(ctx);
  }
#undef MSK_AXIS_BWD
  return 0;
}

}  // extern "C"
