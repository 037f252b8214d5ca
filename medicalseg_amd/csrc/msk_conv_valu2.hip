// 'Same' 5^3 convolution with a TINY output-channel count (CN <= 4, out_tr.conv1: 32 -> ncls, vnet.py:165) on
// the VALU, two W-adjacent output voxels per thread.
//
// PMC on the one-voxel-per-thread kernel (conv_halo_valu_k, 2.3 ms for 32 -> 3 @ 2x128^3, the same time for
// CN = 1..4): VALU active 47 %, as many SALU as VALU instructions, LDS 21 % active, no bank conflicts -- the
// kernel is ISSUE bound (per tap and input channel: one LDS quad, CK*CN scalar weights, 2 FMA instructions).
// With two voxels per thread a kernel row of 5 taps needs 6 LDS quads instead of 10 (x[w+1..w+4] serve both
// voxels), every scalar weight feeds twice the FMAs, and the odd output channel of the two voxels shares one
// v_pk_fma_f32: 1.5 instead of 2 VALU instructions per voxel, tap and input channel for CN = 3.
//
// LDS halo layout: rows split by W parity ([parity][12]) so that the stride-2 accesses x[2*tw + k] of a wavefront
// are contiguous; row pitch 24 quads = 384 B = 128 (mod 256) keeps the two rows a 16-lane group touches on
// disjoint banks.
#include "msk_conv.h"

typedef float f2 __attribute__((ext_vector_type(2)));

namespace {

struct V2Args {
  const float* src;
  int sld;
  float* dst;
  int dld;
  int N, D, H, W, CNout;
  const float* wp;  // [tap][CK][CN] (CN = compute width: an odd output count is padded with a zero column)
  const float* bias;
  int accumulate;
  int tiles_d, tiles_h, tiles_w, nblk;
  unsigned src_bytes;  // extent of src for the staging buffer loads (< 4 GiB: run_gconv chunks the batch)
};

__device__ __forceinline__ int xcd_remap_v2(int bid, int nb) {
  const int q = nb >> 3, r = nb & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int KS, int CK, int CN>
__global__ void __launch_bounds__(256)
conv_halo_valu2_k(V2Args a) {
  constexpr int TD = 4, TH = 8, TW = 16, P = KS / 2;
  constexpr int HD = TD + 2 * P, HH = TH + 2 * P, HW = TW + 2 * P;  // 8 x 12 x 20
  constexpr int HWH = 12, RP = 2 * HWH;                              // padded half row, row pitch (quads)
  constexpr int NV = HD * HH * HW;
  constexpr int QC = CK / 4;
  constexpr int NP = CN / 2, ODD = CN & 1;
  static_assert(CK % 4 == 0 && HW / 2 <= HWH, "quad-aligned channels");
  __shared__ float4 lds[HD * HH * RP];

  const int tid = threadIdx.x;
  int tile = xcd_remap_v2(blockIdx.x, a.nblk);
  const int twi = tile % a.tiles_w;
  tile /= a.tiles_w;
  const int thi = tile % a.tiles_h;
  tile /= a.tiles_h;
  const int tdi = tile % a.tiles_d;
  const int n = tile / a.tiles_d;
  const int d0 = tdi * TD, h0 = thi * TH, w0 = twi * TW;
  const int tw = tid & 7, hy = (tid >> 3) & 7, dz = tid >> 6;
  const int base = (dz * HH + hy) * RP + tw;

  f2 accp[2][NP > 0 ? NP : 1];
  f2 accx = {0.f, 0.f};
#pragma unroll
  for (int v = 0; v < 2; ++v)
#pragma unroll
    for (int p = 0; p < (NP > 0 ? NP : 1); ++p) accp[v][p] = (f2){0.f, 0.f};

  // Staging: a thread owns ONE (h, w) column of the halo tile (12 x 20 = 240 columns) for the whole kernel: its LDS
  // slot and its byte offset in the tensor are computed once per tile, the plane and the channel quad of a pass ride in
  // the buffer load's scalar offset, columns outside the volume read zeros through an out-of-range offset.  (The
  // generic loop re-derived (d, h, w) with divisions and a 64-bit address for each of its 60 elements per thread --
  // in a kernel that is VALU-issue bound.)
  const __amdgpu_buffer_rsrc_t sres = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
  const bool st_live = tid < HH * HW;
  const int st_hh = tid / HW, st_hw = tid % HW;
  const int st_gh = h0 - P + st_hh, st_gw = w0 - P + st_hw;
  const int st_lds = st_hh * RP + (st_hw & 1) * HWH + (st_hw >> 1);
  const unsigned st_off = (st_live && st_gh >= 0 && st_gh < a.H && st_gw >= 0 && st_gw < a.W)
                              ? (unsigned)((((long)n * a.D * a.H + st_gh) * a.W + st_gw) * a.sld * 4)
                              : 0xFFFFFFF0u;
#pragma unroll 1
  for (int qc = 0; qc < QC; ++qc) {
    const int c0 = qc * 4;
    __syncthreads();
    {
      float4 tmp[HD];
#pragma unroll
      for (int hd = 0; hd < HD; ++hd) {
        const int gd = d0 - P + hd;
        const bool dok = gd >= 0 && gd < a.D;  // wave-uniform
        const unsigned soff = dok ? (unsigned)(((long)gd * a.H * a.W * a.sld + c0) * 4) : 0u;
        tmp[hd] = dok ? __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(sres, (int)st_off, (int)soff, 0))
                      : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (st_live) {
#pragma unroll
        for (int hd = 0; hd < HD; ++hd) lds[hd * HH * RP + st_lds] = tmp[hd];
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int rr = 0; rr < KS * KS; ++rr) {
      const float4* row = lds + base + ((rr / KS) * HH + (rr % KS)) * RP;
      float4 x[KS + 1];
#pragma unroll
      for (int k = 0; k <= KS; ++k) x[k] = row[(k & 1) * HWH + (k >> 1)];
      const float* w = a.wp + ((long)rr * KS * CK + c0) * CN;  // wave-uniform: scalar loads
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) {
        const float xa[4] = {x[kw].x, x[kw].y, x[kw].z, x[kw].w};
        const float xb[4] = {x[kw + 1].x, x[kw + 1].y, x[kw + 1].z, x[kw + 1].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float* wc = w + ((long)kw * CK + c) * CN;
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            const f2 wv = {wc[2 * p], wc[2 * p + 1]};
            accp[0][p] += (f2){xa[c], xa[c]} * wv;
            accp[1][p] += (f2){xb[c], xb[c]} * wv;
          }
          if (ODD) accx += (f2){xa[c], xb[c]} * (f2){wc[CN - 1], wc[CN - 1]};
        }
      }
    }
  }

  const int gd = d0 + dz, gh = h0 + hy;
  if (gd < a.D && gh < a.H) {
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int gw = w0 + 2 * tw + v;
      if (gw < a.W) {
        float* o = a.dst + ((((long)n * a.D + gd) * a.H + gh) * a.W + gw) * a.dld;
#pragma unroll
        for (int j = 0; j < CN; ++j) {
          if (j >= a.CNout) break;
          float r = (ODD && j == CN - 1) ? (v == 0 ? accx.x : accx.y) : (j & 1 ? accp[v][j >> 1].y : accp[v][j >> 1].x);
          r += a.bias ? a.bias[j] : 0.f;
          if (a.accumulate) r += o[j];
          o[j] = r;
        }
      }
    }
  }
}

}  // namespace

int msk_gconv_halo_valu2(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap) {
  const int ks = g.kd;
  if (!(g.kd == g.kh && g.kh == g.kw && ks == 5)) return 0;
  if (!(g.sd == 1 && g.sh == 1 && g.sw == 1 && g.pd == 2 && g.ph == 2 && g.pw == 2)) return 0;
  if (!(g.SD == g.DD && g.SH == g.DH && g.SW == g.DW)) return 0;
  if (!(g.CK == 32 && g.CN >= 1 && g.CN <= 4)) return 0;  // the out_tr.conv1 class
  if (g.DW < 16) return 0;                                  // narrow slabs keep the 4x8x8 one-voxel kernel
  if (!((g.sld % 4 == 0) && (((uintptr_t)g.src) % 16 == 0))) return 0;
  const int taps = ks * ks * ks;
  // CN = 3 runs as 4 columns with a zero one: measured 1.69 ms vs 1.85 ms for the odd-width instantiation
  // (pairs of output channels map straight onto v_pk_fma_f32)
  const int cw = g.CN == 3 ? 4 : g.CN;
  const float* wp = msk_pack_weights_get(ctx, w_canon, A, B, taps, swap, g.transposed ? 1 : 0, ks, ks, ks, 0, g.CK, g.CN, 0, cw);
  if (!wp) return -1;
  V2Args a{};
  a.src = g.src; a.sld = g.sld; a.dst = g.dst; a.dld = g.dld;
  a.N = g.N; a.D = g.DD; a.H = g.DH; a.W = g.DW; a.CNout = g.CN;
  a.wp = wp; a.bias = g.bias; a.accumulate = g.accumulate;
  a.tiles_d = msk_cdiv(a.D, 4); a.tiles_h = msk_cdiv(a.H, 8); a.tiles_w = msk_cdiv(a.W, 16);
  const long nblk = (long)a.N * a.tiles_d * a.tiles_h * a.tiles_w;
  if (nblk > 0x7fffffff) return 0;
  {
    const size_t sb = (size_t)g.N * g.SD * g.SH * g.SW * g.sld * sizeof(float);
    if (sb >= 0xFFFFFFF0ull) return 0;
    a.src_bytes = (unsigned)sb;
  }
  a.nblk = (int)nblk;
  const char* tag = "conv_halo_valu2";
  if (ctx->prof && ctx->prof_shapes) {
    char buf[160];
    snprintf(buf, sizeof(buf), "conv_halo_valu2[ck=%d,cn=%d,n=%d,dhw=%dx%dx%d,acc=%d]", g.CK, g.CN, g.N, g.DD, g.DH, g.DW, g.accumulate);
    tag = msk_intern_tag(ctx, buf);
  }
  msk_launch_scope ls(ctx, tag);
  switch (cw) {
    case 1: hipLaunchKernelGGL((conv_halo_valu2_k<5, 32, 1>), dim3((unsigned)nblk), dim3(256), 0, ctx->stream, a); break;
    case 2: hipLaunchKernelGGL((conv_halo_valu2_k<5, 32, 2>), dim3((unsigned)nblk), dim3(256), 0, ctx->stream, a); break;
    default: hipLaunchKernelGGL((conv_halo_valu2_k<5, 32, 4>), dim3((unsigned)nblk), dim3(256), 0, ctx->stream, a); break;
  }
  MSK_LAUNCH_CHECK(ctx);
  return 1;
}
