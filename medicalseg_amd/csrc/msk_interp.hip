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

__global__ void __launch_bounds__(kThreads)
interp_fwd_k(const float* __restrict__ src, int sld, int SD, int SH, int SW, float* __restrict__ dst, int dld, int DD,
             int DH, int DW, int C, long total, float rd, float rh, float rw) {
  for (long t = (long)blockIdx.x * kThreads + threadIdx.x; t < total; t += (long)gridDim.x * kThreads) {
    const int c = (int)(t % C);
    long v = t / C;
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
    auto at = [&](int dd, int hh, int ww) { return b[(((long)dd * SH + hh) * SW + ww) * sld]; };
    const float x00 = at(d0, h0, w0) * (1.f - lw) + at(d0, h0, w1) * lw;
    const float x01 = at(d0, h1, w0) * (1.f - lw) + at(d0, h1, w1) * lw;
    const float x10 = at(d1, h0, w0) * (1.f - lw) + at(d1, h0, w1) * lw;
    const float x11 = at(d1, h1, w0) * (1.f - lw) + at(d1, h1, w1) * lw;
    const float y0 = x00 * (1.f - lh) + x01 * lh;
    const float y1 = x10 * (1.f - lh) + x11 * lh;
    dst[(t / C) * dld + c] = y0 * (1.f - ld_) + y1 * ld_;
  }
}

// g: [outer][n_out][inner][C] (voxel stride gld)  ->  out: [outer][n_in][inner][C] (stride old_)
__global__ void __launch_bounds__(kThreads)
interp_axis_bwd_k(const float* __restrict__ g, int gld, float* __restrict__ out, int old_, long total, int n_out, int n_in,
                  long inner, int C, float ratio, int accumulate) {
  const float inv = 1.f / ratio;
  for (long t = (long)blockIdx.x * kThreads + threadIdx.x; t < total; t += (long)gridDim.x * kThreads) {
    const int c = (int)(t % C);
    long v = t / C;
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
    float acc = 0.f;
    for (int d = lo; d <= hi; ++d) {
      int i0, i1;
      float lam;
      axis_src(d, ratio, n_in, i0, i1, lam);
      float wgt = 0.f;
      if (i0 == s) wgt += 1.f - lam;
      if (i1 == s) wgt += lam;
      if (wgt != 0.f) acc = fmaf(wgt, gp[(long)d * inner * gld], acc);
    }
    float* op = out + (((long)o * n_in + s) * inner + iv) * old_ + c;
    *op = accumulate ? *op + acc : acc;
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
  const long total = msk_voxels(dst) * dst.c;
  msk_launch_scope ls(ctx, "interp_trilinear_fwd");
  hipLaunchKernelGGL(interp_fwd_k, dim3(blocks_for(total, ctx->num_cu)), dim3(kThreads), 0, ctx->stream,
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
  msk_launch_scope ls(ctx, "interp_trilinear_bwd");
  if (rw) {  // [N*Dd*Hd][Wd][1][C] -> [N*Dd*Hd][Ws][1][C]
    const long total = (long)ddst.n * ddst.d * ddst.h * dsrc.w * C;
    hipLaunchKernelGGL(interp_axis_bwd_k, dim3(blocks_for(total, ctx->num_cu)), dim3(kThreads), 0, ctx->stream, cur,
                       cur_ld, t1, C, total, ddst.w, dsrc.w, 1L, C, (float)dsrc.w / ddst.w, 0);
    MSK_LAUNCH_CHECK(ctx);
    cur = t1; cur_ld = C; cur_w = dsrc.w;
  }
  if (rh) {  // [N*Dd][Hd][W][C] -> [N*Dd][Hs][W][C]
    const long total = (long)ddst.n * ddst.d * dsrc.h * cur_w * C;
    hipLaunchKernelGGL(interp_axis_bwd_k, dim3(blocks_for(total, ctx->num_cu)), dim3(kThreads), 0, ctx->stream, cur,
                       cur_ld, t2, C, total, ddst.h, dsrc.h, (long)cur_w, C, (float)dsrc.h / ddst.h, 0);
    MSK_LAUNCH_CHECK(ctx);
    cur = t2; cur_ld = C; cur_h = dsrc.h;
  }
  {  // [N][Dd][H*W][C] -> [N][Ds][H*W][C] (identity map when Dd == Ds)
    const long total = (long)ddst.n * dsrc.d * cur_h * cur_w * C;
    hipLaunchKernelGGL(interp_axis_bwd_k, dim3(blocks_for(total, ctx->num_cu)), dim3(kThreads), 0, ctx->stream, cur,
                       cur_ld, (float*)dsrc.p, dsrc.ld, total, ddst.d, dsrc.d, (long)cur_h * cur_w, C,
                       (float)dsrc.d / ddst.d, accumulate);
    MSK_LAUNCH_CHECK(ctx);
  }
  return 0;
}

}  // extern "C"
