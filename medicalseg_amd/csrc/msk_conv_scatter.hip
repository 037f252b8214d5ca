// Transposed gather with kernel == stride (every VNet up-convolution forward, vnet.py:133-150, and the
// data gradient of every down-convolution, :98): each SOURCE voxel feeds exactly k^3 destination voxels
// and no destination voxel has two contributions.  As a GEMM:
//     out[m][(tap, cn)] = sum_k src[m][k] * W[tap][k][cn],   m = source voxel,
// i.e. the taps are folded into N.  The parity-class gather kernel (gconv_gather_mfma_k) re-read the
// source once per class (8x) and wrote every other destination voxel (64-byte islands): measured
// 0.69 ms for 32ch@64^3 -> 16ch@128^3 against ~0.1 ms of HBM time.  Here a workgroup reads its 128
// source voxels ONCE (A fragments stay in registers), loops over the N tiles, and a wavefront's stores
// cover contiguous destination rows (tap pairs along W are adjacent voxels).
#include "msk_conv.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

// Bf[kc][h][jpad][4]: k = kc*8 + h*4 + q, j = tap*CN + cn (zero padded to jpad, K to 8*KC)
__global__ void __launch_bounds__(256)
pack_scatter_weights_k(const float* __restrict__ w, int A, int B, int taps, int swap, int CK, int CN, int KC, int jpad,
                       float4* __restrict__ out) {
  const long total = (long)KC * 2 * jpad;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int j = (int)(idx % jpad);
    const int h = (int)((idx / jpad) & 1);
    const int kc = (int)(idx / (2L * jpad));
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (j < taps * CN) {
      const int tap = j / CN, n = j % CN;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = kc * 8 + h * 4 + q;
        if (k < CK) {
          const int ia = swap ? n : k, ib = swap ? k : n;  // canonical w[a][b][tap]
          v[q] = w[((long)ia * B + ib) * taps + tap];
        }
      }
    }
    out[idx] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

constexpr int NTG = 4;   // N tiles (of 32 columns) per workgroup pass: 64 accumulator registers
constexpr int KCB = 8;   // 8-channel chunks held in registers at a time (64 channels)

__global__ void __launch_bounds__(256, 3)  // 172 registers without the bound: 4 over the three-wavefront limit
convT_scatter_mfma_k(GConv g, const float4* __restrict__ bf, int KC, int jpad, int vec) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const long M = (long)g.N * g.SD * g.SH * g.SW;
  const long m = (long)blockIdx.x * 128 + wave * 32 + li;   // this lane's A row (source voxel)
  const int nt0 = blockIdx.y * NTG;
  const int ntiles = jpad / 32;

  f32x16 acc[NTG];
#pragma unroll
  for (int t = 0; t < NTG; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

  const float* arow = g.src + (m < M ? m : 0) * g.sld;
  for (int kc0 = 0; kc0 < KC; kc0 += KCB) {
    float4 a[KCB];
#pragma unroll
    for (int i = 0; i < KCB; ++i) {
      const int c0 = (kc0 + i) * 8 + lh * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M && kc0 + i < KC && c0 < g.CK) {
        if (vec) {
          v = *reinterpret_cast<const float4*>(arow + c0);
        } else {
          v.x = arow[c0];
          if (c0 + 1 < g.CK) v.y = arow[c0 + 1];
          if (c0 + 2 < g.CK) v.z = arow[c0 + 2];
          if (c0 + 3 < g.CK) v.w = arow[c0 + 3];
        }
      }
      a[i] = v;
    }
#pragma unroll
    for (int i = 0; i < KCB; ++i) {
      if (kc0 + i < KC) {  // wave-uniform
        const float4* bk = bf + ((long)(kc0 + i) * 2 + lh) * jpad + li;
#pragma unroll
        for (int t = 0; t < NTG; ++t) {
          if (nt0 + t < ntiles) {
            const float4 b = bk[(nt0 + t) * 32];
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b.x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b.y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b.z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b.w, acc[t], 0, 0, 0);
          }
        }
      }
    }
  }

  // D[row = source voxel][col = (tap, cn)] -> dst[(n, d*sd+a, h*sh+b, w*sw+c)][cn]
  unsigned rowbase[16];
  bool rowok[16];
  const long mw = (long)blockIdx.x * 128 + wave * 32;
  // The lane's first row is decoded with divisions, the other 15 follow by carries (rows advance by 1, 1, 1, 5):
  // 64 run-time divisions per lane cost more than the MFMA loop of these k = s layers.
  unsigned w_, h_, d_, n_;
  {
    const long m0 = mw + 4 * lh;
    const unsigned r = (unsigned)(m0 < M ? m0 : 0);
    const unsigned t1 = r / (unsigned)g.SW, t2 = t1 / (unsigned)g.SH;
    w_ = r - t1 * (unsigned)g.SW;
    h_ = t1 - t2 * (unsigned)g.SH;
    n_ = t2 / (unsigned)g.SD;
    d_ = t2 - n_ * (unsigned)g.SD;
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j > 0) {
      w_ += (j & 3) ? 1u : 5u;
      while (w_ >= (unsigned)g.SW) {
        w_ -= (unsigned)g.SW;
        if (++h_ >= (unsigned)g.SH) {
          h_ = 0;
          if (++d_ >= (unsigned)g.SD) {
            d_ = 0;
            ++n_;
          }
        }
      }
    }
    const long mr = mw + (j & 3) + 8 * (j >> 2) + 4 * lh;
    rowok[j] = mr < M;
    rowbase[j] = (((n_ * g.DD + d_ * g.sd) * g.DH + h_ * g.sh) * g.DW + w_ * g.sw) * g.dld;
  }
  const int taps = g.kd * g.kh * g.kw;
#pragma unroll
  for (int t = 0; t < NTG; ++t) {
    const int col = (nt0 + t) * 32 + li;
    if (nt0 + t < ntiles && col < taps * g.CN) {
      const int tap = col / g.CN, cn = col - tap * g.CN;
      const int ta = tap / (g.kh * g.kw), tb = (tap / g.kw) % g.kh, tc = tap % g.kw;
      const unsigned tapoff = ((unsigned)(ta * g.DH + tb) * g.DW + tc) * g.dld + cn;
      const float bv = g.bias ? g.bias[cn] : 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (rowok[j]) {
          float* o = g.dst + rowbase[j] + tapoff;
          float v = acc[t][j] + bv;
          if (g.accumulate) v += *o;
          *o = v;
        }
      }
    }
  }
}

}  // namespace

int msk_gconv_scatter_mfma(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap) {
  // eligibility: transposed gather, kernel == stride, no padding, dst = src * stride exactly
  if (!g.transposed) return 0;
  if (!(g.kd == g.sd && g.kh == g.sh && g.kw == g.sw && g.pd == 0 && g.ph == 0 && g.pw == 0)) return 0;
  if (!(g.DD == g.SD * g.sd && g.DH == g.SH * g.sh && g.DW == g.SW * g.sw)) return 0;
  const int taps = g.kd * g.kh * g.kw;
  if (taps < 2) return 0;
  const long M = (long)g.N * g.SD * g.SH * g.SW;
  const unsigned long dst_elems = (unsigned long)g.N * g.DD * g.DH * g.DW * g.dld;
  if (M >= (1L << 31) || dst_elems >= (1UL << 32)) return 0;  // 32-bit row arithmetic in the epilogue
  // deep levels (<= 16^3 sources) have too few 128-voxel tiles to fill 256 CUs: the parity-class kernel, which
  // also parallelises over the classes, measured faster there (0.08 vs 0.14 ms at 256->128 @ 8^3)
  if (M < 16384 && ctx->conv_impl != 7) return 0;
  const int KC = (g.CK + 7) / 8;
  const int jpad = ((taps * g.CN + 31) / 32) * 32;
  float4* bf = (float4*)msk_workspace2(ctx, (size_t)KC * 2 * jpad * sizeof(float4));
  if (!bf) return -1;
  {
    msk_launch_scope ls(ctx, "pack_weights_scatter");
    long blocks = ((long)KC * 2 * jpad + 255) / 256;
    if (blocks > 4L * ctx->num_cu) blocks = 4L * ctx->num_cu;
    hipLaunchKernelGGL(pack_scatter_weights_k, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, w_canon, A, B, taps, swap,
                       g.CK, g.CN, KC, jpad, bf);
    MSK_LAUNCH_CHECK(ctx);
  }
  const int vec = (g.CK % 4 == 0) && (g.sld % 4 == 0) && (((uintptr_t)g.src) % 16 == 0);
  const char* tag = "convT_scatter_mfma";
  if (ctx->prof && ctx->prof_shapes) {
    char buf[160];
    snprintf(buf, sizeof(buf), "convT_scatter_mfma[ck=%d,cn=%d,k=%dx%dx%d,dst=%dx%dx%dx%d]", g.CK, g.CN, g.kd, g.kh, g.kw, g.N,
             g.DD, g.DH, g.DW);
    tag = msk_intern_tag(ctx, buf);
  }
  msk_launch_scope ls(ctx, tag);
  dim3 grid((unsigned)((M + 127) / 128), (jpad / 32 + NTG - 1) / NTG);
  hipLaunchKernelGGL(convT_scatter_mfma_k, grid, dim3(256), 0, ctx->stream, g, (const float4*)bf, KC, jpad, vec);
  MSK_LAUNCH_CHECK(ctx);
  return 1;
}
