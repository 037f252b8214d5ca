// Weight gradient of the kernel == stride convolutions (every VNet down / up convolution, vnet.py:98,133; UNet3D's
// too): each input voxel meets exactly ONE tap of exactly one output voxel, so
//     dW[cb][ca][tap] = sum over OUTPUT voxels o of  A[o*s + tap][ca] * B[o][cb]
// reads A once and B once when one wavefront keeps the accumulators of all taps (2x2x2: 8 x 16 registers).  The
// generic kernel (wgrad_mfma_k) hands every (kd, kh) tap row to a different workgroup, i.e. streams both tensors
// kd*kh = 4 times and spends ~20 VALU instructions of cursor / range arithmetic per 2 MFMAs: PMC MFMA pipe 20 % busy,
// 1.3 ms per VNet step for ~0.25 ms of HBM time.  Here: no range checks (k == s, no padding: every tap of every output
// voxel is inside the volume), one base offset per step, tap offsets wave-uniform.
#include "msk_conv.h"
#ifdef KS_PROBE_NOMFMA   // knock-out probe (timing only, wrong results): the fp32 matrix instructions of this file become register moves
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a_, b_, c_, x_, y_, z_) ks_probe_keep((a_), (b_), (c_))
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a_, b_, c_, x_, y_, z_) ks_probe_keep((a_), (b_), (c_))
template <typename T> __device__ __forceinline__ T ks_probe_keep(float a, float b, T c) { asm volatile("" ::"v"(a), "v"(b)); c[0] += a * 1e-30f + b * 1e-30f; return c; }
#endif

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


// ---- fine levels (CA <= 32): rows = (tap, ca) -------------------------------------------------------------------------------
// With 16 channels the kernel above leaves half of every MFMA's 32 rows empty (and half of every A load's lanes idle), and
// one workgroup per 32-column tile re-reads the big fine-resolution tensor A once per tile: 0.232 ms for the 64 -> 16
// up-convolution at 128^3 (403 MB: 1.2 TB/s; the fp32 matrix pipe alone needs 0.11 ms of it) plus 0.07 ms for a separate
// pass over the same tensor for the bias gradient.  Here the GEMM's rows are the flattened (tap, ca) pairs -- a 32-row tile
// of a 16-channel tensor is two W-adjacent fine voxels: 128 contiguous bytes, 256 with the other voxel parity -- so
//   * every MFMA row and every load lane is live, and one wavefront owns RT row tiles x ALL CT column tiles: A and B are
//     read exactly once (row groups of a bigger tap x channel product read disjoint parts of A and share the small B);
//   * the bias gradient (column sums of B for a convolution, of A over the taps for a transposed one) is accumulated from
//     the operands already in registers and travels as a tail of the partial slab: no second pass over the tensor;
//   * the four wavefronts of a workgroup add their accumulators through LDS in a fixed order before one of them writes:
//     a quarter of the partial slabs (16 MB instead of 67 MB at 128^3) for the reduction kernels to read back.
struct Ks2Args {
  float* partial;     // [slabs][pitch]: taps*CA*CB sums in (tap, ca, cb) order, then the bias sums
  long pitch;         // floats per slab (multiple of 4)
  int slabs;          // grid.y; 4 wavefronts = 4 voxel ranges each
  int bias_mode;      // 0 none, 1 column sums of B, 2 sums of A over the taps (single row group only)
  unsigned a_bytes, b_bytes;
};

template <int RT, int CT>
__global__ void __launch_bounds__(256, 2)
wgrad_ks2_k(WGrad g, Ks2Args a) {
  constexpr int NA = RT * CT;
  __shared__ float red[NA * 16 * 64 + (RT + CT) * 64];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int taps = g.kd * g.kh * g.kw, R = taps * g.CA;
  const int rg = blockIdx.x;
  const int splits = a.slabs * 4;
  const int split = blockIdx.y * 4 + wave;

  const long M = (long)g.N * g.BD * g.BH * g.BW;
  long per = (M + splits - 1) / splits;
  per = (per + 1) & ~1L;  // even: the two lane halves (voxel parity) stay in step
  const long m0 = (long)split * per;
  long m1 = m0 + per;
  if (m1 > M) m1 = M;     // m0 >= M: an empty range, the wavefront only takes part in the reduction

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, a.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, a.b_bytes, 0x00020000);

  // this lane's row of every row tile: byte offset of (tap, ca) from the voxel o*s
  unsigned rowoff[RT];
  bool rowok[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int r = (rg * RT + rt) * 32 + li;
    rowok[rt] = r < R;
    const int tap = rowok[rt] ? r / g.CA : 0;
    const int ca = rowok[rt] ? r - tap * g.CA : 0;
    const int ta = tap / (g.kh * g.kw), tb = (tap / g.kw) % g.kh, tc = tap % g.kw;
    rowoff[rt] = (unsigned)(((ta * g.AH + tb) * g.AW + tc) * g.ald + ca) * 4u;
  }
  bool colok[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) colok[ct] = ct * 32 + li < g.CB;

  f32x16 acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[rt][ct][j] = 0.f;
  float asum[RT], bsum[CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) asum[rt] = 0.f;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) bsum[ct] = 0.f;

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
  float av_n[U][RT], bv_n[U][CT];
  auto load_batch = [&](long mm) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long mu = mm + 2 * u;
      const bool live = mu < m1;
      const unsigned abase = (unsigned)(((cn * g.AD + cd * g.sd) * g.AH + chh * g.sh) * g.AW + cw * g.sw) * (unsigned)g.ald * 4u;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) av_n[u][rt] = ks_load(ra, (live && rowok[rt]) ? abase + rowoff[rt] : kOOBk, 0u);
      const unsigned bbase = (unsigned)mu * (unsigned)g.bld * 4u + (unsigned)li * 4u;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) bv_n[u][ct] = ks_load(rb, (live && colok[ct]) ? bbase + ct * 128u : kOOBk, 0u);
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
    float av[U][RT], bv[U][CT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) av[u][rt] = av_n[u][rt];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) bv[u][ct] = bv_n[u][ct];
    }
    load_batch(m + 2 * U);
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][rt], bv[u][ct], acc[rt][ct], 0, 0, 0);
      if (a.bias_mode == 2) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) asum[rt] += av[u][rt];
      } else if (a.bias_mode == 1) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) bsum[ct] += bv[u][ct];
      }
    }
  }

  // wavefronts 3 -> 2 -> 1 -> 0 hand their sums down through LDS (a fixed order: deterministic)
  float* rsum = red + NA * 16 * 64;
  for (int w = 3; w >= 1; --w) {
    if (wave == w) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int j = 0; j < 16; ++j) red[((rt * CT + ct) * 16 + j) * 64 + lane] = acc[rt][ct][j];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) rsum[rt * 64 + lane] = asum[rt];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) rsum[(RT + ct) * 64 + lane] = bsum[ct];
    }
    __syncthreads();
    if (wave == w - 1) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int j = 0; j < 16; ++j) acc[rt][ct][j] += red[((rt * CT + ct) * 16 + j) * 64 + lane];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) asum[rt] += rsum[rt * 64 + lane];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) bsum[ct] += rsum[(RT + ct) * 64 + lane];
    }
    __syncthreads();
  }

  float* slab = a.partial + (long)blockIdx.y * a.pitch;
  if (wave == 0) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int cb = ct * 32 + li;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int r = (rg * RT + rt) * 32 + (j & 3) + 8 * (j >> 2) + 4 * lh;   // = tap*CA + ca
          if (r < R && cb < g.CB) slab[(long)r * g.CB + cb] = acc[rt][ct][j];
        }
      }
    if (a.bias_mode == 1 && rg == 0) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const float s = bsum[ct] + __shfl_xor(bsum[ct], 32);
        if (lh == 0 && ct * 32 + li < g.CB) slab[(long)R * g.CB + ct * 32 + li] = s;
      }
    }
    if (a.bias_mode == 2) {   // single row group: every (tap, ca) row is here; fold the voxel parities, then the taps below
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const float s = asum[rt] + __shfl_xor(asum[rt], 32);
        if (lh == 0) red[rt * 32 + li] = s;
      }
    }
  }
  __syncthreads();
  if (a.bias_mode == 2 && wave == 0 && lane < g.CA) {
    float s = 0.f;
    for (int tap = 0; tap < taps; ++tap) s += red[tap * g.CA + lane];
    slab[(long)R * g.CB + lane] = s;
  }
}


// fine levels: rows = (tap, ca); 1 handled, 0 not eligible, < 0 error
int wgrad_ks2(msk_ctx* ctx, const WGrad& g, int taps, long M, size_t abytes, size_t bbytes, bool k_eq_s) {
  if (ctx->ks_legacy & 1) return 0;                         // the one-tap-per-tile kernel everywhere (A/B)
  if (g.CA > 32 || g.CB > 128 || g.CB % 4 || M < 4096) return 0;
  const int rtiles = (taps * g.CA + 31) / 32, cbt = (g.CB + 31) / 32;
  const int CT = cbt <= 1 ? 1 : (cbt == 2 ? 2 : 4);
  int RT = 8 / CT;
  while (RT > 1 && RT / 2 >= rtiles) RT >>= 1;
  const int rgroups = (rtiles + RT - 1) / RT;
  // the bias gradient rides along when this problem has one (run_wgrad): column sums of B, or of A when one row group holds all taps
  int bias_mode = 0;
  // (sums of A over every tap count each fine voxel once only when kernel == stride)
  if (g.db) bias_mode = g.db_src == 1 ? 1 : ((rgroups == 1 && k_eq_s) ? 2 : 0);
  const int nbias = bias_mode == 1 ? g.CB : (bias_mode == 2 ? g.CA : 0);
  const long per = (long)taps * g.CA * g.CB;
  const long pitch = per + ((nbias + 3) & ~3);
  // one round of resident workgroups (2 per CU), at least 64 output voxels per wavefront
  long slabs = 2L * ctx->num_cu / rgroups;
  if (slabs > M / 256) slabs = M / 256;
  if (slabs < 1) slabs = 1;
  float* partial = (float*)msk_workspace(ctx, (size_t)slabs * pitch * sizeof(float));
  if (!partial) return -1;
  Ks2Args a{partial, pitch, (int)slabs, bias_mode, (unsigned)abytes, (unsigned)bbytes};
  {
    const char* tag = "wgrad_ks2_mfma";
    if (ctx->prof && ctx->prof_shapes) {
      char buf[160];
      snprintf(buf, sizeof(buf), "wgrad_ks2_mfma[ca=%d,cb=%d,k=%dx%dx%d,M=%ld,slabs=%ld,rt=%d,ct=%d,bias=%d]", g.CA, g.CB, g.kd, g.kh, g.kw, M,
               slabs, RT, CT, bias_mode);
      tag = msk_intern_tag(ctx, buf);
    }
    msk_launch_scope ls(ctx, tag);
    const dim3 grid((unsigned)rgroups, (unsigned)slabs);
#define KS2(rt, ct) hipLaunchKernelGGL((wgrad_ks2_k<rt, ct>), grid, dim3(256), 0, ctx->stream, g, a)
    switch (RT * 8 + CT) {
      case 8 * 8 + 1: KS2(8, 1); break;
      case 4 * 8 + 1: KS2(4, 1); break;
      case 2 * 8 + 1: KS2(2, 1); break;
      case 1 * 8 + 1: KS2(1, 1); break;
      case 4 * 8 + 2: KS2(4, 2); break;
      case 2 * 8 + 2: KS2(2, 2); break;
      case 1 * 8 + 2: KS2(1, 2); break;
      case 2 * 8 + 4: KS2(2, 4); break;
      default: KS2(1, 4); break;
    }
#undef KS2
    MSK_LAUNCH_CHECK(ctx);
  }
  // slabs whose four voxel ranges are all empty were still written (zeros): every slab counts
  const int rc = msk_wgrad_reduce_ex(ctx, partial, (int)slabs, pitch, taps, g.CA, g.CB, g.dw, g.accumulate, nbias, g.db, g.accumulate);
  if (rc != 0) return rc;
  if (bias_mode) ctx->wgrad_db_done = true;
  return 1;
}

}  // namespace

// returns 1 when handled, 0 when not eligible, < 0 on error
int msk_wgrad_ks(msk_ctx* ctx, const WGrad& g) {
  // unpadded windows inside A; round 4: any stride (the anisotropic MRI levels overlap along W, see msk_gconv_ks_fwd)
  if (!(g.pd == 0 && g.ph == 0 && g.pw == 0)) return 0;
  const bool k_eq_s = g.kd == g.sd && g.kh == g.sh && g.kw == g.sw;
  const int taps = g.kd * g.kh * g.kw;
  if (taps < 2 || taps > 32) return 0;
  if ((g.BD - 1) * g.sd + g.kd > g.AD || (g.BH - 1) * g.sh + g.kh > g.AH || (g.BW - 1) * g.sw + g.kw > g.AW) return 0;
  const long M = (long)g.N * g.BD * g.BH * g.BW;
  const size_t abytes = (size_t)g.N * g.AD * g.AH * g.AW * g.ald * sizeof(float);
  const size_t bbytes = (size_t)M * g.bld * sizeof(float);
  if (M >= (1L << 31) || abytes >= 0xFFFFFFF0ull || bbytes >= 0xFFFFFFF0ull) return 0;
  if (int r2 = wgrad_ks2(ctx, g, taps, M, abytes, bbytes, k_eq_s)) return r2;
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
