// 'Same' k^3 convolution with a TINY reduction-channel count (CK <= 4) on the MFMA pipe: the data gradient
// of out_tr.conv1 (ncls -> 32 channels, vnet.py:165) and in_tr.conv1 forward-type problems.
//
// conv_halo_mfma_k walks K in 8-channel chunks, so CK = 3 would run at 3/8 MFMA efficiency; the VALU kernel
// (conv_halo_valu_k) needs CK*CN wave-uniform weights per tap and measured 1.93 ms for 3 -> 32 @ 2x128^3
// (0.64 ms of FMA time).  Here the GEMM K dimension enumerates (tap, channel) pairs TIGHTLY: one
// v_mfma_f32_32x32x2_f32 consumes the same channel of two taps (lane half 0: tap t, lane half 1: tap
// t + ceil(taps/2)), so K = 125*CK has no padding beyond one tap.  The halo tile is staged planar
// ([channel][voxel], odd pitch) so every A operand is a conflict-free ds_read_b32 at
// voxel + tap_offset[t], the offsets coming from a small LDS table.
#include "msk_conv.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct TKArgs {
  const float* src;
  int sld;
  float* dst;
  int dld;
  int N, D, H, W;
  int CN, npad;
  const float* wt;  // [HT][CK][2][npad]
  unsigned wt_bytes, src_bytes;
  const float* bias;
  int accumulate;
  int tiles_d, tiles_h, tiles_w, nblk;
};

__device__ __forceinline__ int xcd_remap_tk(int bid, int nb) {
  const int q = nb >> 3, r = nb & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__global__ void __launch_bounds__(256)
pack_tightk_weights_k(const float* __restrict__ w, int A, int B, int ks, int swap, int flip, int CK, int CN, int npad,
                      float* __restrict__ out) {
  const int taps = ks * ks * ks, HT = (taps + 1) / 2;
  const long total = (long)HT * CK * 2 * npad;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int n = (int)(idx % npad);
    long r = idx / npad;
    const int h = (int)(r & 1);
    r >>= 1;
    const int c = (int)(r % CK);
    const int ti = (int)(r / CK);
    const int tap = h * HT + ti;
    float v = 0.f;
    if (tap < taps && n < CN) {
      const int st = flip ? taps - 1 - tap : tap;  // flipping all three axes reverses the linear tap index
      const int ia = swap ? n : c, ib = swap ? c : n;
      v = w[((long)ia * B + ib) * taps + st];
    }
    out[idx] = v;
  }
}

template <int TD, int TH, int TW, int KS, int CK>
__global__ void __launch_bounds__(256)
conv_halo_tightk_k(TKArgs a) {
  constexpr int P = KS / 2;
  constexpr int HD = TD + 2 * P, HH = TH + 2 * P, HW = TW + 2 * P;
  constexpr int NV = HD * HH * HW, NVP = NV | 1;
  constexpr int TAPS = KS * KS * KS, HT = (TAPS + 1) / 2;
  constexpr int MR = TD * TH * TW / 128;
  __shared__ float lds[CK * NVP];
  __shared__ int toff[2 * HT];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  int tile = xcd_remap_tk(blockIdx.x, a.nblk);
  const int twi = tile % a.tiles_w;
  tile /= a.tiles_w;
  const int thi = tile % a.tiles_h;
  tile /= a.tiles_h;
  const int tdi = tile % a.tiles_d;
  const int n = tile / a.tiles_d;
  const int d0 = tdi * TD, h0 = thi * TH, w0 = twi * TW;
  const int nt = blockIdx.y;

  for (int i = tid; i < 2 * HT; i += 256) {
    const int tap = (i / HT) * HT + (i % HT);
    toff[i] = tap < TAPS ? ((tap / (KS * KS)) * HH + (tap / KS) % KS) * HW + tap % KS : 0;
  }
  const __amdgpu_buffer_rsrc_t sres = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
  for (int hv = tid; hv < NV; hv += 256) {
    const int hd = hv / (HH * HW), rem = hv % (HH * HW), hh = rem / HW, hw = rem % HW;
    const int gd = d0 - P + hd, gh = h0 - P + hh, gw = w0 - P + hw;
    const bool in = gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
    // 32-bit offsets into a buffer resource (tensors are chunked below 4 GiB); halo voxels outside the volume read
    // zeros through an out-of-range offset
    const unsigned voff = in ? (unsigned)((((n * a.D + gd) * a.H + gh) * a.W + gw) * a.sld) * 4u : 0xFFFFFFF0u;
#pragma unroll
    for (int c = 0; c < CK; ++c)
      lds[c * NVP + hv] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sres, (int)voff, 4 * c, 0));
  }
  __syncthreads();

  int abase[MR];
#pragma unroll
  for (int r = 0; r < MR; ++r) {
    const int l = (wave * MR + r) * 32 + li;
    abase[r] = ((l / (TH * TW)) * HH + (l / TW) % TH) * HW + l % TW;
  }
  f32x16 acc[MR];
#pragma unroll
  for (int r = 0; r < MR; ++r)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[r][j] = 0.f;

  // weights through a raw buffer resource: the (tap pair, channel) part of the address is wave-uniform and rides in
  // the scalar offset, the lane part is one constant (a flat pointer cost ~3 VALU instructions per load)
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);
  const unsigned wlane = (unsigned)(lh * a.npad + nt * 32 + li) * 4u;
  const int* tl = toff + lh * HT;
#pragma unroll 3
  for (int ti = 0; ti < HT; ++ti) {
    const int off = tl[ti];
#pragma unroll
    for (int c = 0; c < CK; ++c) {
      const float b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
          wres, (int)wlane, (int)((unsigned)((ti * CK + c) * 2 * a.npad) * 4u), 0));
#pragma unroll
      for (int r = 0; r < MR; ++r)
        acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[c * NVP + abase[r] + off], b, acc[r], 0, 0, 0);
    }
  }

  const int co = nt * 32 + li;
  if (co < a.CN) {
    const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
    for (int r = 0; r < MR; ++r) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = (j & 3) + 8 * (j >> 2) + 4 * lh;
        const int l = (wave * MR + r) * 32 + row;
        const int gd = d0 + l / (TH * TW), gh = h0 + (l / TW) % TH, gw = w0 + l % TW;
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
int launch_tk(msk_ctx* ctx, TKArgs& a, int CK, int ntn, const char* tag) {
  a.tiles_d = msk_cdiv(a.D, TD);
  a.tiles_h = msk_cdiv(a.H, TH);
  a.tiles_w = msk_cdiv(a.W, TW);
  const long nblk = (long)a.N * a.tiles_d * a.tiles_h * a.tiles_w;
  if (nblk > 0x7fffffff) return msk_fail(ctx, __FILE__, __LINE__, "conv_halo_tightk", "grid too large");
  a.nblk = (int)nblk;
  msk_launch_scope ls(ctx, tag);
  dim3 grid((unsigned)nblk, ntn);
  switch (CK) {
    case 1: hipLaunchKernelGGL((conv_halo_tightk_k<TD, TH, TW, KS, 1>), grid, dim3(256), 0, ctx->stream, a); break;
    case 2: hipLaunchKernelGGL((conv_halo_tightk_k<TD, TH, TW, KS, 2>), grid, dim3(256), 0, ctx->stream, a); break;
    case 3: hipLaunchKernelGGL((conv_halo_tightk_k<TD, TH, TW, KS, 3>), grid, dim3(256), 0, ctx->stream, a); break;
    default: hipLaunchKernelGGL((conv_halo_tightk_k<TD, TH, TW, KS, 4>), grid, dim3(256), 0, ctx->stream, a); break;
  }
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

}  // namespace

int msk_gconv_halo_tightk(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap) {
  const int ks = g.kd;
  if (!(g.kd == g.kh && g.kh == g.kw && ks == 5)) return 0;
  if (!(g.sd == 1 && g.sh == 1 && g.sw == 1 && g.pd == 2 && g.ph == 2 && g.pw == 2)) return 0;
  if (!(g.SD == g.DD && g.SH == g.DH && g.SW == g.DW)) return 0;
  // CK = 1 (in_tr.conv1, N = 16 half-padded) measured 0.39 ms here vs 0.35 ms on the VALU kernel; a tiny CN as well
  // belongs to the VALU kernel
  if (g.CK < 2 || g.CK > 4 || g.CN < 8) return 0;
  const int taps = ks * ks * ks, HT = (taps + 1) / 2;
  const int npad = ((g.CN + 31) / 32) * 32;
  float* wt = (float*)msk_workspace2(ctx, (size_t)HT * g.CK * 2 * npad * sizeof(float));
  if (!wt) return -1;
  {
    msk_launch_scope ls(ctx, "pack_weights_tightk");
    hipLaunchKernelGGL(pack_tightk_weights_k, dim3(64), dim3(256), 0, ctx->stream, w_canon, A, B, ks, swap,
                       g.transposed ? 1 : 0, g.CK, g.CN, npad, wt);
    MSK_LAUNCH_CHECK(ctx);
  }
  TKArgs a{};
  a.src = g.src; a.sld = g.sld; a.dst = g.dst; a.dld = g.dld;
  a.N = g.N; a.D = g.DD; a.H = g.DH; a.W = g.DW;
  a.CN = g.CN; a.npad = npad; a.wt = wt; a.bias = g.bias; a.accumulate = g.accumulate;
  a.wt_bytes = (unsigned)((size_t)HT * g.CK * 2 * npad * sizeof(float));
  {
    const size_t sb = (size_t)g.N * g.SD * g.SH * g.SW * g.sld * sizeof(float);
    if (sb >= 0xFFFFFFF0ull) return 0;
    a.src_bytes = (unsigned)sb;
  }
  const char* tag = "conv_halo_tightk";
  if (ctx->prof && ctx->prof_shapes) {
    char buf[160];
    snprintf(buf, sizeof(buf), "conv_halo_tightk[ck=%d,cn=%d,n=%d,dhw=%dx%dx%d,acc=%d]", g.CK, g.CN, g.N, g.DD, g.DH, g.DW,
             g.accumulate);
    tag = msk_intern_tag(ctx, buf);
  }
  int rc;
  if (g.DW >= 32 && g.DW % 32 < 1) rc = launch_tk<2, 4, 32, 5>(ctx, a, g.CK, npad / 32, tag);
  else if (g.DW >= 16 && g.DW % 16 < 1) rc = launch_tk<2, 8, 16, 5>(ctx, a, g.CK, npad / 32, tag);
  else rc = launch_tk<4, 8, 8, 5>(ctx, a, g.CK, npad / 32, tag);
  return rc == 0 ? 1 : rc;
}
