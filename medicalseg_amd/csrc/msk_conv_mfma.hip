// MFMA (fp32-input v_mfma_f32_32x32x2_f32) convolution kernels for gfx950.
//
// (1) conv_halo_mfma_k -- 'same' stride-1 odd-kernel convolutions (the 5x5x5 LUConv
//     layers = 99.8% of VNet FLOPs; also their data gradients, which are the same
//     convolution with flipped/transposed weights).  Implicit GEMM
//        M = output voxels (32 per MFMA tile), N = out channels (32), K = taps x Cin.
//     Per workgroup: a TDxTHxTW output tile; for each 8-channel K-chunk the input halo
//     tile is staged ONCE in LDS as [quad][voxel][4] (conflict-free ds_read_b128 with
//     an immediate per-tap offset) and reused by all k^3 taps.  Weights are pre-packed
//     so a wavefront's B fragment is one contiguous 1 KiB global load (L2 resident,
//     identical for every workgroup), double-buffered one tap-row ahead.
//     K order inside a chunk is (h, q): lane (i, h) feeds channel 8kc+4h+q to the q-th
//     MFMA -- A and B agree, and any K permutation is legal in a GEMM sum.
//
// (2) wgrad_mfma_k -- weight gradients: M = Cin tile (32), N = Cout tile (32),
//     K = voxels.  Both operands are one coalesced dword per lane straight from global
//     (lanes run along channels in NDHWC), the dy fragment is reused across the kw taps
//     of a tap-row, accumulators (kw x 16 VGPRs) stay in registers for the whole voxel
//     range, split-K partials are reduced deterministically by wgrad_reduce_k.
#include "msk_conv.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct HaloArgs {
  const float* src;
  int sld;
  float* dst;
  int dld;
  int N, D, H, W;
  int CK, CN;
  const float4* wm;
  int KC, npad;
  const float* bias;
  int accumulate;
  int tiles_d, tiles_h, tiles_w;
  int nblk;
  int vec;
};

__device__ __forceinline__ int xcd_remap(int bid, int nb) {
  // bijective "block b runs on XCD b%8" -> contiguous chunk per XCD (guide T1)
  const int q = nb >> 3, r = nb & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int TD, int TH, int TW, int KS>
__global__ void __launch_bounds__(256, 2) conv_halo_mfma_k(HaloArgs a) {
  constexpr int P = KS / 2;
  constexpr int HD = TD + 2 * P, HH = TH + 2 * P, HW = TW + 2 * P;
  constexpr int NV = HD * HH * HW;
  constexpr int NVP = NV | 1;  // odd plane pitch
  constexpr int MR = TD * TH * TW / 128;
  static_assert(TD * TH * TW % 128 == 0, "tile must hold 4 waves x MR x 32 voxels");
  __shared__ float4 lds[2 * NVP];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  int tile = xcd_remap(blockIdx.x, a.nblk);
  const int twi = tile % a.tiles_w;
  tile /= a.tiles_w;
  const int thi = tile % a.tiles_h;
  tile /= a.tiles_h;
  const int tdi = tile % a.tiles_d;
  const int n = tile / a.tiles_d;
  const int d0 = tdi * TD, h0 = thi * TH, w0 = twi * TW;
  const int nt = blockIdx.y;

  int abase[MR];
#pragma unroll
  for (int r = 0; r < MR; ++r) {
    const int l = (wave * MR + r) * 32 + li;
    const int dz = l / (TH * TW), hy = (l / TW) % TH, wx = l % TW;
    abase[r] = (dz * HH + hy) * HW + wx + lh * NVP;
  }
  f32x16 acc[MR];
#pragma unroll
  for (int r = 0; r < MR; ++r)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[r][j] = 0.f;

  const long tapstride = (long)a.KC * 2 * a.npad;  // float4 units
  const float4* wlane = a.wm + ((long)lh * a.npad + nt * 32 + li);

  for (int kc = 0; kc < a.KC; ++kc) {
    __syncthreads();
    // ---- stage the halo tile for channels [8kc, 8kc+8) ----
    for (int it = tid; it < NV * 2; it += 256) {
      const int hv = it >> 1, q = it & 1;
      const int hd = hv / (HH * HW), rem = hv % (HH * HW), hh = rem / HW, hw = rem % HW;
      const int gd = d0 - P + hd, gh = h0 - P + hh, gw = w0 - P + hw;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int c0 = kc * 8 + q * 4;
      if (gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W && c0 < a.CK) {
        const float* p = a.src + ((((long)n * a.D + gd) * a.H + gh) * a.W + gw) * a.sld + c0;
        if (a.vec) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          v.x = p[0];
          if (c0 + 1 < a.CK) v.y = p[1];
          if (c0 + 2 < a.CK) v.z = p[2];
          if (c0 + 3 < a.CK) v.w = p[3];
        }
      }
      lds[q * NVP + hv] = v;
    }
    __syncthreads();

    const float4* wk = wlane + (long)kc * 2 * a.npad;
    float4 bcur[KS], bnxt[KS];
#pragma unroll
    for (int kw = 0; kw < KS; ++kw) bcur[kw] = wk[(long)kw * tapstride];
    for (int rr = 0; rr < KS * KS; ++rr) {
      if (rr + 1 < KS * KS) {
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) bnxt[kw] = wk[(long)((rr + 1) * KS + kw) * tapstride];
      }
      const int kd = rr / KS, kh = rr % KS;
      const int rowoff = (kd * HH + kh) * HW;
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) {
        float4 av[MR];
#pragma unroll
        for (int r = 0; r < MR; ++r) av[r] = lds[abase[r] + rowoff + kw];
        const float4 b = bcur[kw];
#pragma unroll
        for (int r = 0; r < MR; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r].x, b.x, acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < MR; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r].y, b.y, acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < MR; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r].z, b.z, acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < MR; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r].w, b.w, acc[r], 0, 0, 0);
      }
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) bcur[kw] = bnxt[kw];
    }
  }

  // ---- epilogue: D[row = voxel][col = out channel] ----
  const int co = nt * 32 + li;
  if (co < a.CN) {
    const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
    for (int r = 0; r < MR; ++r) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = (j & 3) + 8 * (j >> 2) + 4 * lh;
        const int l = (wave * MR + r) * 32 + row;
        const int dz = l / (TH * TW), hy = (l / TW) % TH, wx = l % TW;
        const int gd = d0 + dz, gh = h0 + hy, gw = w0 + wx;
        if (gd < a.D && gh < a.H && gw < a.W) {
          float* o = a.dst + ((((long)n * a.D + gd) * a.H + gh) * a.W + gw) * a.dld + co;
          float v = acc[r][j] + bv;
          if (a.accumulate) v += *o;
          *o = v;
        }
      }
    }
  }
}

template <int TD, int TH, int TW, int KS>
int launch_halo(msk_ctx* ctx, HaloArgs& a, int ntiles_n) {
  a.tiles_d = msk_cdiv(a.D, TD);
  a.tiles_h = msk_cdiv(a.H, TH);
  a.tiles_w = msk_cdiv(a.W, TW);
  const long nblk = (long)a.N * a.tiles_d * a.tiles_h * a.tiles_w;
  if (nblk > 0x7fffffff) return msk_fail(ctx, __FILE__, __LINE__, "conv_halo", "grid too large");
  a.nblk = (int)nblk;
  msk_launch_scope ls(ctx, KS == 5 ? "conv_halo_mfma_k5" : "conv_halo_mfma_k3");
  hipLaunchKernelGGL((conv_halo_mfma_k<TD, TH, TW, KS>), dim3((unsigned)nblk, ntiles_n), dim3(256), 0, ctx->stream, a);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

// ---------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------
template <int KW>
__global__ void __launch_bounds__(256, 2) wgrad_mfma_k(WGrad g, int splits, float* __restrict__ partial) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int ca_tiles = (g.CA + 31) >> 5, cb_tiles = (g.CB + 31) >> 5;
  int b = blockIdx.x;
  const int cbt = b % cb_tiles;
  b /= cb_tiles;
  const int cat = b % ca_tiles;
  const int rr = b / ca_tiles;  // tap row = kd*kh_count + kh
  const int kd = rr / g.kh, kh = rr % g.kh;
  const int split = blockIdx.y * 4 + wave;
  if (split >= splits) return;

  const long M = (long)g.N * g.BD * g.BH * g.BW;
  long per = (M + splits - 1) / splits;
  per = (per + 1) & ~1L;  // even, so lane halves stay in step
  const long m0 = (long)split * per;
  long m1 = m0 + per;
  if (m1 > M) m1 = M;

  const int ca = cat * 32 + li, cb = cbt * 32 + li;
  const bool ca_ok = ca < g.CA, cb_ok = cb < g.CB;

  f32x16 acc[KW];
#pragma unroll
  for (int k = 0; k < KW; ++k)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;

  // coordinates of this lane-half's first voxel
  long m = m0 + lh;
  int ow = (int)(m % g.BW);
  int oh = (int)((m / g.BW) % g.BH);
  int od = (int)((m / ((long)g.BW * g.BH)) % g.BD);
  int n = (int)(m / ((long)g.BW * g.BH * g.BD));

  // software pipeline: operands of step s+1 are in flight while step s's MFMAs issue
  float av_n[KW], bv_n;
  auto load_step = [&](long mm) {
    const bool live = mm < m1;
    const int id = od * g.sd - g.pd + kd, ih = oh * g.sh - g.ph + kh;
    const bool rowok = live && id >= 0 && id < g.AD && ih >= 0 && ih < g.AH;
    const float* arow = g.A + ((((long)n * g.AD + id) * g.AH + ih) * g.AW) * g.ald + ca;
    const int iw0 = ow * g.sw - g.pw;
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      const int iw = iw0 + k;
      av_n[k] = (rowok && ca_ok && iw >= 0 && iw < g.AW) ? arow[(long)iw * g.ald] : 0.f;
    }
    bv_n = (live && cb_ok) ? g.B[mm * g.bld + cb] : 0.f;
  };
  load_step(m);
  for (; m - lh < m1; m += 2) {  // uniform trip count across the wave
    float av[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) av[k] = av_n[k];
    const float bv = bv_n;
    // advance two voxels and issue the next step's loads
    ow += 2;
    while (ow >= g.BW) {
      ow -= g.BW;
      if (++oh >= g.BH) {
        oh = 0;
        if (++od >= g.BD) {
          od = 0;
          ++n;
        }
      }
    }
    load_step(m + 2);
#pragma unroll
    for (int k = 0; k < KW; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k], bv, acc[k], 0, 0, 0);
  }

  const int taps = g.kd * g.kh * g.kw;
  if (cb_ok) {
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      const int tap = rr * g.kw + k;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = (j & 3) + 8 * (j >> 2) + 4 * lh;
        const int oca = cat * 32 + row;
        if (oca < g.CA) partial[(((long)split * taps + tap) * g.CA + oca) * g.CB + cb] = acc[k][j];
      }
    }
  }
}

template <int KW>
int launch_wgrad(msk_ctx* ctx, const WGrad& g, int splits, float* partial) {
  const int ca_tiles = (g.CA + 31) / 32, cb_tiles = (g.CB + 31) / 32;
  const int rows = g.kd * g.kh;
  msk_launch_scope ls(ctx, "wgrad_mfma");
  hipLaunchKernelGGL((wgrad_mfma_k<KW>), dim3(rows * ca_tiles * cb_tiles, (splits + 3) / 4), dim3(256), 0, ctx->stream,
                     g, splits, partial);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

}  // namespace

int msk_gconv_halo_mfma(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap) {
  // eligibility: cubic odd kernel (3 or 5), stride 1, 'same' padding, same spatial dims
  const int ks = g.kd;
  if (!(g.kd == g.kh && g.kh == g.kw && (ks == 3 || ks == 5))) return 0;
  if (!(g.sd == 1 && g.sh == 1 && g.sw == 1)) return 0;
  if (!(g.pd == ks / 2 && g.ph == ks / 2 && g.pw == ks / 2)) return 0;
  if (!(g.SD == g.DD && g.SH == g.DH && g.SW == g.DW)) return 0;
  // With stride 1 and p = k/2 the transposed gather equals a forward gather with flipped taps.
  const int flip = g.transposed ? 1 : 0;
  const int taps = ks * ks * ks;
  const int KC = (g.CK + 7) / 8;
  const int npad = ((g.CN + 31) / 32) * 32;
  const size_t wbytes = (size_t)taps * KC * 2 * npad * 4 * sizeof(float);
  float* wm = (float*)msk_workspace2(ctx, wbytes);
  if (!wm) return -1;
  if (msk_pack_weights(ctx, w_canon, A, B, taps, swap, flip, ks, ks, ks, 1, g.CK, g.CN, KC, npad, wm) != 0) return -1;

  HaloArgs a{};
  a.src = g.src; a.sld = g.sld; a.dst = g.dst; a.dld = g.dld;
  a.N = g.N; a.D = g.DD; a.H = g.DH; a.W = g.DW;
  a.CK = g.CK; a.CN = g.CN;
  a.wm = reinterpret_cast<const float4*>(wm);
  a.KC = KC; a.npad = npad;
  a.bias = g.bias; a.accumulate = g.accumulate;
  a.vec = (g.CK % 4 == 0) && (g.sld % 4 == 0) && (((uintptr_t)g.src) % 16 == 0);
  const int ntn = npad / 32;
  int rc;
  if (ks == 5) {
    if (g.DW >= 32) rc = launch_halo<2, 4, 32, 5>(ctx, a, ntn);
    else if (g.DW >= 16) rc = launch_halo<2, 8, 16, 5>(ctx, a, ntn);
    else rc = launch_halo<4, 8, 8, 5>(ctx, a, ntn);
  } else {
    if (g.DW >= 32) rc = launch_halo<2, 4, 32, 3>(ctx, a, ntn);
    else if (g.DW >= 16) rc = launch_halo<2, 8, 16, 3>(ctx, a, ntn);
    else rc = launch_halo<4, 8, 8, 3>(ctx, a, ntn);
  }
  return rc == 0 ? 1 : rc;
}

int msk_wgrad_mfma(msk_ctx* ctx, const WGrad& g) {
  if (g.kw < 1 || g.kw > 5) return 0;
  const int taps = g.kd * g.kh * g.kw;
  const long M = (long)g.N * g.BD * g.BH * g.BW;
  const int ca_tiles = (g.CA + 31) / 32, cb_tiles = (g.CB + 31) / 32;
  const long tasks = (long)g.kd * g.kh * ca_tiles * cb_tiles;
  long splits = (4096 + tasks - 1) / tasks;
  const long maxs = M / 256 > 0 ? M / 256 : 1;
  if (splits > maxs) splits = maxs;
  if (splits < 1) splits = 1;
  splits = (splits + 3) & ~3L;
  // keep the partial slab below 1 GiB
  const size_t per = (size_t)taps * g.CA * g.CB * sizeof(float);
  while (splits > 4 && splits * per > ((size_t)1 << 30)) splits -= 4;
  float* partial = (float*)msk_workspace(ctx, (size_t)splits * per);
  if (!partial) return -1;
  int rc;
  switch (g.kw) {
    case 1: rc = launch_wgrad<1>(ctx, g, (int)splits, partial); break;
    case 2: rc = launch_wgrad<2>(ctx, g, (int)splits, partial); break;
    case 3: rc = launch_wgrad<3>(ctx, g, (int)splits, partial); break;
    case 4: rc = launch_wgrad<4>(ctx, g, (int)splits, partial); break;
    default: rc = launch_wgrad<5>(ctx, g, (int)splits, partial); break;
  }
  if (rc != 0) return rc;
  // splits whose voxel range is empty wrote nothing: count only the populated ones
  long per_vox = (M + splits - 1) / splits;
  per_vox = (per_vox + 1) & ~1L;
  const int used = (int)((M + per_vox - 1) / per_vox);
  rc = msk_wgrad_reduce(ctx, partial, used, taps, g.CA, g.CB, g.dw, g.accumulate);
  return rc == 0 ? 1 : rc;
}
