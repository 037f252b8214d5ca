// Transposed gather with kernel == stride (every VNet up-convolution forward, vnet.py:133-150, and the
// data gradient of every down-convolution, :98): each SOURCE voxel feeds exactly k^3 destination voxels
// and no destination voxel has two contributions.  As a GEMM:
//     out[m][(tap, cn)] = sum_k src[m][k] * W[tap][k][cn],   m = source voxel,
// i.e. the taps are folded into N.  The parity-class gather kernel (gconv_gather_mfma_k) re-read the
// source once per class (8x) and wrote every other destination voxel (64-byte islands): measured
// 0.69 ms for 32ch@64^3 -> 16ch@128^3 against ~0.1 ms of HBM time.  Here a workgroup reads its 128
// source voxels ONCE (A fragments stay in registers), loops over the N tiles, and a wavefront's stores
// cover contiguous destination rows (tap pairs along W are adjacent voxels).
#include <optional>

#include "msk_conv.h"
#ifdef KS_PROBE_NOMFMA   // knock-out probe (timing only, wrong results): the fp32 matrix instructions of this file become register moves
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a_, b_, c_, x_, y_, z_) ks_probe_keep((a_), (b_), (c_))
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a_, b_, c_, x_, y_, z_) ks_probe_keep((a_), (b_), (c_))
template <typename T> __device__ __forceinline__ T ks_probe_keep(float a, float b, T c) { asm volatile("" ::"v"(a), "v"(b)); c[0] += a * 1e-30f + b * 1e-30f; return c; }
#endif
#include "msk_wbf.h"   // msk_bn_stats_merge

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

constexpr int KCB = 8;   // 8-channel chunks held in registers at a time (64 channels)

// D is produced TRANSPOSED: rows = (tap, cn) columns of the GEMM above (weights as the MFMA A operand), columns = the
// wavefront's 32 source voxels (x as the B operand).  A lane then owns ONE source voxel -- one address decode instead of
// sixteen -- and four consecutive rows of a register quad are four consecutive output channels of one tap: 16-byte
// stores (and 16-byte loads for the accumulate) instead of 64 scalar ones per N tile group.
template <int NTG>  // N tiles (of 32 rows) per workgroup pass: 16 accumulator registers each
__global__ void __launch_bounds__(256, 3)
convT_scatter_mfma_k(GConv g, const float4* __restrict__ bf, int KC, int jpad) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const long M = (long)g.N * g.SD * g.SH * g.SW;
  const long m = (long)blockIdx.x * 128 + wave * 32 + li;   // this lane's source voxel
  const int nt0 = blockIdx.y * NTG;
  const bool mok = m < M;

  f32x16 acc[NTG];
#pragma unroll
  for (int t = 0; t < NTG; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

  const float* xrow = g.src + (mok ? m : 0) * g.sld + lh * 4;
  const float4* wl = bf + (long)lh * jpad + nt0 * 32 + li;
  for (int kc0 = 0; kc0 < KC; kc0 += KCB) {
    float4 xv[KCB];
#pragma unroll
    for (int i = 0; i < KCB; ++i)   // CK % 4 == 0: a quad is inside the voxel or not at all
      xv[i] = (mok && (kc0 + i) * 8 + lh * 4 < g.CK) ? *reinterpret_cast<const float4*>(xrow + (kc0 + i) * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < KCB; ++i) {
      if (kc0 + i < KC) {  // wave-uniform
        const float4* wk = wl + (long)(kc0 + i) * 2 * jpad;
        float4 w4[NTG];
#pragma unroll
        for (int t = 0; t < NTG; ++t) w4[t] = wk[t * 32];
#pragma unroll
        for (int t = 0; t < NTG; ++t) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[t].x, xv[i].x, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[t].y, xv[i].y, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[t].z, xv[i].z, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[t].w, xv[i].w, acc[t], 0, 0, 0);
        }
      }
    }
  }
  if (!mok) return;

  // D[row = (tap, cn)][col = source voxel] -> dst[(n, d*sd+a, h*sh+b, w*sw+c)][cn .. cn+3]
  unsigned base;
  {
    const unsigned r = (unsigned)m;
    const unsigned t1 = r / (unsigned)g.SW, t2 = t1 / (unsigned)g.SH;
    const unsigned w_ = r - t1 * (unsigned)g.SW, h_ = t1 - t2 * (unsigned)g.SH;
    const unsigned n_ = t2 / (unsigned)g.SD, d_ = t2 - n_ * (unsigned)g.SD;
    base = (((n_ * g.DD + d_ * g.sd) * g.DH + h_ * g.sh) * g.DW + w_ * g.sw) * g.dld;
  }
  const int khw = g.kh * g.kw;
#pragma unroll
  for (int t = 0; t < NTG; ++t) {
    float4 old[4];
    unsigned off[4];
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = (nt0 + t) * 32 + 8 * q + 4 * lh;  // the quad's first row; CN % 4 == 0: one tap per quad
      const int tap = row / g.CN, cn = row - tap * g.CN;
      const int ta = tap / khw, tb = (tap - ta * khw) / g.kw, tc = tap - ta * khw - tb * g.kw;
      ok[q] = row < jpad && tap < g.kd * khw;
      off[q] = base + ((unsigned)(ta * g.DH + tb) * g.DW + tc) * g.dld + cn;
      old[q] = (g.accumulate && ok[q]) ? *reinterpret_cast<const float4*>(g.dst + off[q]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (ok[q]) {
        const int row = (nt0 + t) * 32 + 8 * q + 4 * lh;
        const int cn = row % g.CN;
        const float4 bv = g.bias ? make_float4(g.bias[cn], g.bias[cn + 1], g.bias[cn + 2], g.bias[cn + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 v;
        v.x = acc[t][4 * q + 0] + bv.x + old[q].x;
        v.y = acc[t][4 * q + 1] + bv.y + old[q].y;
        v.z = acc[t][4 * q + 2] + bv.z + old[q].z;
        v.w = acc[t][4 * q + 3] + bv.w + old[q].w;
        *reinterpret_cast<float4*>(g.dst + off[q]) = v;
      }
    }
  }
}


// ---- LDS-staged form ------------------------------------------------------------------------------------------------------
// The kernel above reads and writes in MFMA-fragment shape: a lane owns a voxel, so a wavefront's 16-byte accesses land as
// 32-byte runs on 32 different lines (x loads at a voxel stride, stores at two fine voxels' stride), and a wavefront's load,
// MFMA and store phases follow each other.  Measured at 32 -> 16 @ 64^3 -> 128^3: 0.051 ms without the stores, 0.091 ms
// without the loads, 0.116 ms together (335 MB: 2.9 TB/s).  Here a workgroup stages its 64 source voxels through LDS with
// whole-line loads (16 B per lane, consecutive lanes consecutive addresses), takes the MFMA operand from LDS as
// ds_read_b128 (voxel pitch = an odd number of 16-byte slots: conflict-free), and every wavefront owns whole 32-row tiles
// (for 16 output channels: the two W-adjacent fine voxels of one (kd, kh)) whose results go through a private LDS patch and
// leave as 1 KB contiguous stores -- the accumulate reads and the bias ride on the same coalesced pass.
// STATS (round 5, msk_convT3d_fwd_ex): the BatchNorm statistics of the stored values ride along -- every lane keeps (n, mean, M2)
// of the 4 channels of its store quad over the 8 voxels it stores per tile (shifted by its first value), the 8 lane groups of a
// wavefront are merged with shuffles (Chan), the (tile, quad) records of the workgroup meet in LDS and thread c merges the taps of
// channel c in tap order: one record per workgroup and channel for msk_bn_stats_merge.  The separate pass over the up-convolution's
// output (268 MB at 16ch@128^3, 47 us) is gone.
struct ScWF { float n, mean, m2; };
__device__ __forceinline__ ScWF sc_merge(ScWF a, ScWF b) {
  if (b.n == 0.f) return a;
  if (a.n == 0.f) return b;
  ScWF r;
  r.n = a.n + b.n;
  const float d = b.mean - a.mean, f = b.n / r.n;
  r.mean = a.mean + d * f;
  r.m2 = a.m2 + b.m2 + d * d * a.n * f;
  return r;
}
template <int TPW, bool STATS = false>  // 32-row tiles of (tap, cn) per wavefront
__global__ void __launch_bounds__(256, TPW == 1 ? 4 : 3)
convT_scatter_lds_k(GConv g, const float4* __restrict__ bf, int KC, int jpad, float* __restrict__ stat_partial /*[gridDim.x][CN][3]*/) {
  extern __shared__ float4 smem4[];
  float* smem = reinterpret_cast<float*>(smem4);
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int XP = KC * 8 + 4;                       // floats per staged voxel: (XP / 4) odd
  constexpr int OP = 36;                           // floats per voxel of a wavefront's output patch
  float* Xs = smem;                                // [64][XP]
  float* Os = smem + 64 * XP + wave * (32 * OP);   // [32][OP] per wavefront
  unsigned* vbase = reinterpret_cast<unsigned*>(smem + 64 * XP + 4 * 32 * OP);  // [64] destination element offset of a source voxel
  ScWF* rec = reinterpret_cast<ScWF*>(smem + 64 * XP + 4 * 32 * OP + 64);       // STATS: [4 * TPW tiles][8 quads][4] records
  const long M = (long)g.N * g.SD * g.SH * g.SW;
  const long m0 = (long)blockIdx.x * 64;

  // stage x: 64 voxels x KC*8 channels (zero padded), 16 B per thread per pass
  {
    const int P4 = KC * 2;
    const float* src = g.src + m0 * g.sld;
    for (int idx = tid; idx < 64 * P4; idx += 256) {
      const int v = idx / P4, p = idx - v * P4;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m0 + v < M && p * 4 < g.CK) val = *reinterpret_cast<const float4*>(src + (long)v * g.sld + p * 4);
      *reinterpret_cast<float4*>(Xs + v * XP + p * 4) = val;
    }
    if (tid < 64) {
      const long m = m0 + tid;
      unsigned b = 0xFFFFFFFFu;
      if (m < M) {
        const unsigned r = (unsigned)m;
        const unsigned t1 = r / (unsigned)g.SW, t2 = t1 / (unsigned)g.SH;
        const unsigned w_ = r - t1 * (unsigned)g.SW, h_ = t1 - t2 * (unsigned)g.SH;
        const unsigned n_ = t2 / (unsigned)g.SD, d_ = t2 - n_ * (unsigned)g.SD;
        b = (((n_ * g.DD + d_ * g.sd) * g.DH + h_ * g.sh) * g.DW + w_ * g.sw) * g.dld;
      }
      vbase[tid] = b;
    }
  }
  __syncthreads();

  const int ntiles = jpad >> 5;
  const int khw = g.kh * g.kw, rows = g.kd * khw * g.CN;
#pragma unroll 1
  for (int t = 0; t < TPW; ++t) {
    const int tile = wave * TPW + t;
    if (tile >= ntiles) break;   // wave-uniform; no barrier below
    f32x16 acc[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[h][j] = 0.f;
    const float4* wl = bf + (long)lh * jpad + tile * 32 + li;
    const float* x0 = Xs + li * XP + lh * 4;
    for (int kc0 = 0; kc0 < KC; kc0 += 4) {
      float4 w4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) w4[i] = kc0 + i < KC ? wl[(long)(kc0 + i) * 2 * jpad] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (kc0 + i < KC) {  // wave-uniform
          const float4 xa = *reinterpret_cast<const float4*>(x0 + (kc0 + i) * 8);
          const float4 xb = *reinterpret_cast<const float4*>(x0 + 32 * XP + (kc0 + i) * 8);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[i].x, xa.x, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[i].x, xb.x, acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[i].y, xa.y, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[i].y, xb.y, acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[i].z, xa.z, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[i].z, xb.z, acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[i].w, xa.w, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[i].w, xb.w, acc[1], 0, 0, 0);
        }
      }
    }
    // this lane's quad of the tile in the store pass: rows tile*32 + 4*rq .. +3 = one tap, 4 consecutive output channels
    const int rq = lane & 7, row = tile * 32 + rq * 4;
    const bool rok = row < rows;
    const int tap = rok ? row / g.CN : 0, cn = rok ? row - tap * g.CN : 0;
    const int ta = tap / khw, tb = (tap - ta * khw) / g.kw, tc = tap - ta * khw - tb * g.kw;
    const unsigned toff = ((unsigned)(ta * g.DH + tb) * g.DW + tc) * g.dld + cn;
    const float4 bv = (g.bias && rok) ? *reinterpret_cast<const float4*>(g.bias + cn) : make_float4(0.f, 0.f, 0.f, 0.f);
    float sK[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, sn = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // D[row][col = voxel li]: lane holds rows 8q + 4lh + {0..3}
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(Os + li * OP + 8 * q + 4 * lh) = make_float4(acc[h][4 * q], acc[h][4 * q + 1], acc[h][4 * q + 2], acc[h][4 * q + 3]);
      // the patch is private to the wavefront: LDS operations of one wavefront complete in order
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int v = i * 8 + (lane >> 3);
        const unsigned vb = vbase[h * 32 + v];
        float4 o = *reinterpret_cast<const float4*>(Os + v * OP + rq * 4);
        if (rok && vb != 0xFFFFFFFFu) {
          float* dp = g.dst + (size_t)vb + toff;
          o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
          if (g.accumulate) {
            const float4 old = *reinterpret_cast<const float4*>(dp);
            o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
          }
          *reinterpret_cast<float4*>(dp) = o;
          if constexpr (STATS) {
            const float ov[4] = {o.x, o.y, o.z, o.w};
            if (sn == 0.f) {
#pragma unroll
              for (int j = 0; j < 4; ++j) sK[j] = ov[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float d = ov[j] - sK[j];
              s1[j] += d;
              s2[j] = fmaf(d, d, s2[j]);
            }
            sn += 1.f;
          }
        }
      }
    }
    if constexpr (STATS) {
      // the lane's quad over its <= 8 voxels -> the wavefront's 32 voxels (lane groups lane >> 3, fixed order of the merges)
      ScWF w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        w[j].n = sn;
        w[j].mean = sn > 0.f ? sK[j] + s1[j] / sn : 0.f;
        w[j].m2 = sn > 0.f ? fmaxf(s2[j] - s1[j] * s1[j] / sn, 0.f) : 0.f;
      }
#pragma unroll
      for (int off = 8; off < 64; off <<= 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ScWF o2;
          o2.n = __shfl_xor(w[j].n, off, 64);
          o2.mean = __shfl_xor(w[j].mean, off, 64);
          o2.m2 = __shfl_xor(w[j].m2, off, 64);
          // both partners must add in the same order: the lane with the lower group index first
          w[j] = (lane & off) ? sc_merge(o2, w[j]) : sc_merge(w[j], o2);
        }
      }
      if (lane < 8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) rec[(tile * 8 + lane) * 4 + j] = w[j];
      }
    }
  }
  if constexpr (STATS) {
    __syncthreads();
    if (tid < g.CN) {
      const int taps = g.kd * khw;
      ScWF a = {0.f, 0.f, 0.f};
      for (int tap = 0; tap < taps; ++tap) {
        const int row = tap * g.CN + tid;
        a = sc_merge(a, rec[((row >> 5) * 8 + ((row & 31) >> 2)) * 4 + (row & 3)]);
      }
      float* p = stat_partial + ((long)blockIdx.x * g.CN + tid) * 3;
      p[0] = a.n;
      p[1] = a.mean;
      p[2] = a.m2;
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
  // 16-byte accesses: channel counts, voxel strides and pointers aligned to 4 floats
  if (g.CK % 4 || g.CN % 4 || g.sld % 4 || g.dld % 4 || ((uintptr_t)g.src) % 16 || ((uintptr_t)g.dst) % 16)
    return 0;
  const long M = (long)g.N * g.SD * g.SH * g.SW;
  const unsigned long dst_elems = (unsigned long)g.N * g.DD * g.DH * g.DW * g.dld;
  if (M >= (1L << 31) || dst_elems >= (1UL << 32)) return 0;  // 32-bit row arithmetic in the epilogue
  // deep levels (<= 16^3 sources) have too few 128-voxel tiles to fill 256 CUs: the parity-class kernel, which
  // also parallelises over the classes, measured faster there (0.08 vs 0.14 ms at 256->128 @ 8^3)
  if (M < 16384 && ctx->conv_impl != 7) return 0;
  const int KC = (g.CK + 7) / 8;
  const int jpad = ((taps * g.CN + 31) / 32) * 32;
  const float4* bf = (const float4*)msk_pack_scatter_get(ctx, w_canon, A, B, taps, swap, g.CK, g.CN, KC, jpad);   // cached image (round 5)
  if (!bf && ctx->small_pack_cache) return -1;
  if (!bf) {
    float4* bfw = (float4*)msk_workspace2(ctx, (size_t)KC * 2 * jpad * sizeof(float4));
    if (!bfw) return -1;
    bf = bfw;
    msk_launch_scope ls(ctx, "pack_weights_scatter");
    long blocks = ((long)KC * 2 * jpad + 255) / 256;
    if (blocks > 4L * ctx->num_cu) blocks = 4L * ctx->num_cu;
    hipLaunchKernelGGL(pack_scatter_weights_k, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, w_canon, A, B, taps, swap,
                       g.CK, g.CN, KC, jpad, bfw);
    MSK_LAUNCH_CHECK(ctx);
  }
  const char* tag = "convT_scatter_mfma";
  if (ctx->prof && ctx->prof_shapes) {
    char buf[160];
    snprintf(buf, sizeof(buf), "convT_scatter_mfma[ck=%d,cn=%d,k=%dx%dx%d,dst=%dx%dx%dx%d]", g.CK, g.CN, g.kd, g.kh, g.kw, g.N,
             g.DD, g.DH, g.DW);
    tag = msk_intern_tag(ctx, buf);
  }
  std::optional<msk_launch_scope> ls;   // (ended by hand in front of the statistics merge: profile brackets do not nest)
  ls.emplace(ctx, tag);
  const int ntiles = jpad / 32;
  // LDS-staged form: up to 16 row tiles (4 per wavefront), the staged voxels + patches within 64 KB; option ks_legacy bit 1 = fragment-shaped kernel (A/B)
  const int tpw0 = (ntiles + 3) / 4, tpw_t = tpw0 <= 1 ? 1 : (tpw0 == 2 ? 2 : 4);
  const size_t lds = ((size_t)64 * (KC * 8 + 4) + 4 * 32 * 36 + 64 + 4 * tpw_t * 32 * 3) * sizeof(float);
  if (ntiles <= 16 && lds <= 65536 && (!g.bias || ((uintptr_t)g.bias) % 16 == 0) && !(ctx->ks_legacy & 2)) {
    const dim3 grid((unsigned)((M + 63) / 64));
    // BatchNorm statistics of the output in the store pass (msk_convT3d_fwd_ex; option "ks_stats" 0 = the separate pass)
    const bool stats = g.stats != nullptr && !g.accumulate && !g.stats_ps && g.CN <= 256 && ctx->ks_stats;
    float* sp = nullptr;
    if (stats) {
      sp = (float*)msk_workspace(ctx, (size_t)grid.x * g.CN * 3 * sizeof(float));
      if (!sp) return -1;
    }
#define SC_LDS(T_) \
    do { \
      if (stats) hipLaunchKernelGGL((convT_scatter_lds_k<T_, true>), grid, dim3(256), lds, ctx->stream, g, (const float4*)bf, KC, jpad, sp); \
      else hipLaunchKernelGGL((convT_scatter_lds_k<T_, false>), grid, dim3(256), lds, ctx->stream, g, (const float4*)bf, KC, jpad, sp); \
    } while (0)
    if (tpw_t == 1) SC_LDS(1); else if (tpw_t == 2) SC_LDS(2); else SC_LDS(4);
#undef SC_LDS
    MSK_LAUNCH_CHECK(ctx);
    ls.reset();
    if (stats) {
      if (msk_bn_stats_merge(ctx, sp, (int)grid.x, g.CN, g.stats, g.fin) != 0) return -1;
      ctx->stats_fused = true;
    }
    return 1;
  }
  const int ntg = ntiles % 4 == 0 ? 4 : (ntiles % 2 == 0 ? 2 : 1);
  dim3 grid((unsigned)((M + 127) / 128), ntiles / ntg);
  switch (ntg) {
    case 4: hipLaunchKernelGGL((convT_scatter_mfma_k<4>), grid, dim3(256), 0, ctx->stream, g, (const float4*)bf, KC, jpad); break;
    case 2: hipLaunchKernelGGL((convT_scatter_mfma_k<2>), grid, dim3(256), 0, ctx->stream, g, (const float4*)bf, KC, jpad); break;
    default: hipLaunchKernelGGL((convT_scatter_mfma_k<1>), grid, dim3(256), 0, ctx->stream, g, (const float4*)bf, KC, jpad); break;
  }
  MSK_LAUNCH_CHECK(ctx);
  return 1;
}
