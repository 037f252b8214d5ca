// Weight gradient of the kernel == stride convolutions (every VNet down / up convolution, vnet.py:98,133; UNet3D's
// too): each input voxel meets exactly ONE tap of exactly one output voxel, so
//     dW[cb][ca][tap] = sum over OUTPUT voxels o of  A[o*s + tap][ca] * B[o][cb]
// reads A once and B once when one wavefront keeps the accumulators of all taps (2x2x2: 8 x 16 registers).  The
// generic kernel (wgrad_mfma_k) hands every (kd, kh) tap row to a different workgroup, i.e. streams both tensors
// kd*kh = 4 times and spends ~20 VALU instructions of cursor / range arithmetic per 2 MFMAs: PMC MFMA pipe 20 % busy,
// 1.3 ms per VNet step for ~0.25 ms of HBM time.  Here: no range checks (k == s, no padding: every tap of every output
// voxel is inside the volume), one base offset per step, tap offsets wave-uniform.
#include "msk_conv.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr unsigned kOOBk = 0xFFFFFFF0u;

__device__ __forceinline__ float ks_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}

// KT taps per wavefront (tap group tg covers taps [tg*KT, tg*KT + KT)); grid.x = (tap group, ca tile, cb tile),
// grid.y * 4 + wave = split of the output-voxel range.
template <int KT>
__global__ void __launch_bounds__(256, 2)
wgrad_ks_mfma_k(WGrad g, int splits, float* __restrict__ partial, unsigned a_bytes, unsigned b_bytes) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int ca_tiles = (g.CA + 31) >> 5, cb_tiles = (g.CB + 31) >> 5;
  const int taps = g.kd * g.kh * g.kw;
  int b = blockIdx.x;
  const int cbt = b % cb_tiles;
  b /= cb_tiles;
  const int cat = b % ca_tiles;
  const int tg = b / ca_tiles;
  const int split = blockIdx.y * 4 + wave;
  if (split >= splits) return;

  const long M = (long)g.N * g.BD * g.BH * g.BW;
  long per = (M + splits - 1) / splits;
  per = (per + 1) & ~1L;  // even: the two lane halves (voxel parity) stay in step
  const long m0 = (long)split * per;
  long m1 = m0 + per;
  if (m1 > M) m1 = M;

  const int ca = cat * 32 + li, cb = cbt * 32 + li;
  const bool ca_ok = ca < g.CA, cb_ok = cb < g.CB;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, b_bytes, 0x00020000);

  // byte offset of tap t relative to the voxel (o*s): wave-uniform -> scalar operand of the load
  unsigned tapoff[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    const int tap = tg * KT + t;
    const int a = tap / (g.kh * g.kw), bb = (tap / g.kw) % g.kh, c = tap % g.kw;
    tapoff[t] = tap < taps ? (unsigned)(((a * g.AH + bb) * g.AW + c) * g.ald) * 4u : 0u;
  }
  const int ntaps = min(KT, taps - tg * KT);

  f32x16 acc[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

  // output-voxel cursor of this lane half
  long m = m0 + lh;
  int cw, chh, cd, cn;
  {
    const unsigned mu = (unsigned)(m < M ? m : 0);
    const unsigned t1 = mu / (unsigned)g.BW, t2 = t1 / (unsigned)g.BH;
    cw = (int)(mu - t1 * (unsigned)g.BW);
    chh = (int)(t1 - t2 * (unsigned)g.BH);
    cn = (int)(t2 / (unsigned)g.BD);
    cd = (int)(t2 - (unsigned)cn * (unsigned)g.BD);
  }
  constexpr int U = 4;  // steps per batch: the next batch's loads are issued before this batch's MFMAs
  float av_n[U][KT], bv_n[U];
  auto load_batch = [&](long mm) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long mu = mm + 2 * u;
      const bool live = mu < m1;
      const unsigned abase = (unsigned)((((cn * g.AD + cd * g.sd) * g.AH + chh * g.sh) * g.AW + cw * g.sw) * g.ald + ca) * 4u;
      const unsigned voff = (live && ca_ok) ? abase : kOOBk;
#pragma unroll
      for (int t = 0; t < KT; ++t) av_n[u][t] = t < ntaps ? ks_load(ra, voff, tapoff[t]) : 0.f;
      bv_n[u] = ks_load(rb, (live && cb_ok) ? ((unsigned)mu * (unsigned)g.bld + (unsigned)cb) * 4u : kOOBk, 0u);
      cw += 2;  // next voxel of this parity
      while (cw >= g.BW) {
        cw -= g.BW;
        if (++chh >= g.BH) {
          chh = 0;
          if (++cd >= g.BD) {
            cd = 0;
            ++cn;
          }
        }
      }
    }
  };
  load_batch(m);
  for (; m - lh < m1; m += 2 * U) {  // uniform trip count across the wave
    float av[U][KT], bv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int t = 0; t < KT; ++t) av[u][t] = av_n[u][t];
      bv[u] = bv_n[u];
    }
    load_batch(m + 2 * U);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < KT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][t], bv[u], acc[t], 0, 0, 0);
  }

  if (cb_ok) {
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const int tap = tg * KT + t;
      if (tap >= taps) break;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = (j & 3) + 8 * (j >> 2) + 4 * lh;
        const int oca = cat * 32 + row;
        if (oca < g.CA) partial[(((long)split * taps + tap) * g.CA + oca) * g.CB + cb] = acc[t][j];
      }
    }
  }
}

}  // namespace

// returns 1 when handled, 0 when not eligible, < 0 on error
int msk_wgrad_ks(msk_ctx* ctx, const WGrad& g) {
  if (!(g.kd == g.sd && g.kh == g.sh && g.kw == g.sw && g.pd == 0 && g.ph == 0 && g.pw == 0)) return 0;
  const int taps = g.kd * g.kh * g.kw;
  if (taps < 2 || taps > 32) return 0;
  if ((g.BD - 1) * g.sd + g.kd > g.AD || (g.BH - 1) * g.sh + g.kh > g.AH || (g.BW - 1) * g.sw + g.kw > g.AW) return 0;
  const long M = (long)g.N * g.BD * g.BH * g.BW;
  const size_t abytes = (size_t)g.N * g.AD * g.AH * g.AW * g.ald * sizeof(float);
  const size_t bbytes = (size_t)M * g.bld * sizeof(float);
  if (M >= (1L << 31) || abytes >= 0xFFFFFFF0ull || bbytes >= 0xFFFFFFF0ull) return 0;
  constexpr int KT = 8;
  const int tgroups = (taps + KT - 1) / KT;
  const int ca_tiles = (g.CA + 31) / 32, cb_tiles = (g.CB + 31) / 32;
  const long tasks = (long)tgroups * ca_tiles * cb_tiles;
  const size_t per = (size_t)taps * g.CA * g.CB * sizeof(float);
  // ~2 rounds of (CUs x 8 resident wavefronts), at least 64 output voxels per wavefront, slabs <= 256 MiB
  long splits = (2L * 8 * ctx->num_cu + tasks - 1) / tasks;
  const long maxs = M / 64 > 0 ? M / 64 : 1;
  if (splits > maxs) splits = maxs;
  if (splits < 1) splits = 1;
  splits = (splits + 3) & ~3L;
  while (splits > 4 && splits * per > ((size_t)1 << 28)) splits -= 4;
  float* partial = (float*)msk_workspace(ctx, (size_t)splits * per);
  if (!partial) return -1;
  {
    const char* tag = "wgrad_ks_mfma";
    if (ctx->prof && ctx->prof_shapes) {
      char buf[160];
      snprintf(buf, sizeof(buf), "wgrad_ks_mfma[ca=%d,cb=%d,k=%dx%dx%d,M=%ld,splits=%ld]", g.CA, g.CB, g.kd, g.kh, g.kw, M, splits);
      tag = msk_intern_tag(ctx, buf);
    }
    msk_launch_scope ls(ctx, tag);
    hipLaunchKernelGGL((wgrad_ks_mfma_k<KT>), dim3((unsigned)tasks, (unsigned)((splits + 3) / 4)), dim3(256), 0, ctx->stream, g,
                       (int)splits, partial, (unsigned)abytes, (unsigned)bbytes);
    MSK_LAUNCH_CHECK(ctx);
  }
  long per_vox = (M + splits - 1) / splits;
  per_vox = (per_vox + 1) & ~1L;
  const int used = (int)((M + per_vox - 1) / per_vox);
  const int rc = msk_wgrad_reduce(ctx, partial, used, taps, g.CA, g.CB, g.dw, g.accumulate);
  return rc == 0 ? 1 : rc;
}
