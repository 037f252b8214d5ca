// 'Same' 5^3 convolution with a TINY output-channel count (5*CN <= 16: out_tr.conv1, 32 -> ncls <= 3, vnet.py:165)
// on the fp32 matrix pipe with the kd taps FOLDED INTO THE MFMA COLUMNS, marching along D.
//
// Padding CN = 3 to a 16-wide MFMA tile wastes 13 of 16 columns; the channel-quad VALU kernel this replaces
// (conv_halo_valu2_k, 1.7 ms for 32 -> 3 @ 2x128^3) staged ONE channel quad of a 3-D halo tile at a time, i.e. it
// fetched 16 of the 128 bytes of a voxel per pass, eight passes per tile: round-1 counters showed 19x the algorithmic
// traffic on the fabric.  Here
//   * the N dimension of v_mfma_f32_16x16x4_f32 enumerates (co, kd) pairs (15 of 16 columns for CN = 3): one INPUT
//     plane d' contributes to the five output planes d' + 2 - kd at once,
//         Z[w][(co, kd)] = sum_{kh, kw, ci} x[d', h + kh - 2, w + kw - 2, ci] * W[kd][kh][kw][ci][co],
//     K = 25 (kh, kw) taps x channel quads, M = 16 consecutive output positions along W (no halo rows in M);
//   * a workgroup owns an (8 x 16) column of (h, w) and walks along D.  The partial sums ride in the accumulator
//     registers: before plane d' + 1 is accumulated every column moves one kd to the right inside its 16-lane row
//     (DPP row_shr:1), so column (co, kd) always holds the partial sum of output plane d' + 2 - kd; what leaves
//     kd = 4 is a finished output plane.  No 3-D halo: each input plane is staged ONCE per column, whole 128-byte
//     voxels (all channels), halo only in (h, w): 12 x 20 / (8 x 16) = 1.9x, served by L2 between neighbouring columns.
//   * LDS holds one input plane (12 x 20 voxels x 32 channels, bank-spread layout below: conflict-free ds_read_b32 for
//     the A operand); the next plane is prefetched into registers during the 200 MFMAs of
//     the current one.  B operands (weights, 51 KB packed per lane) stream from L1/L2 as 16-byte loads.
#include "msk_conv.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct FNArgs {
  const float* src;
  int sld;
  float* dst;
  int dld;
  int N, D, H, W, CN;
  const float4* wb;  // [tap2d = kh*5 + kw][quad group][lane][4]: B operand of quad 4*group + e for lane (k = lane/16, col = lane%16)
  const float* bias;
  const float* prelu;
  int accumulate;
  int tiles_h, tiles_w, segs, seg_len, nblk;
  unsigned src_bytes;
};

__device__ __forceinline__ int xcd_remap_fn(int bid, int nb) {
  const int q = nb >> 3, r = nb & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// wb[((tap2d*QG + g)*64 + lane)*4 + e] = W[kd][kh][kw][ci = 4*(4g + e) + lane/16][co], (co, kd) = divmod(lane % 16, 5)
__global__ void __launch_bounds__(256)
pack_foldn_weights_k(const float* __restrict__ w, int A, int B, int swap, int flip, int CK, int CN, int QG,
                     float* __restrict__ out) {
  const int total = 25 * QG * 64 * 4;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int e = idx & 3, lane = (idx >> 2) & 63, g = (idx >> 8) % QG, tap2 = (idx >> 8) / QG;
    const int j = lane & 15, k = lane >> 4;
    const int ci = 4 * (4 * g + e) + k;
    float v = 0.f;
    if (j < 5 * CN && ci < CK) {
      const int co = j / 5, kd = j % 5;
      int tap = kd * 25 + tap2;
      if (flip) tap = 124 - tap;
      const int ia = swap ? co : ci, ib = swap ? ci : co;
      v = w[((long)ia * B + ib) * 125 + tap];
    }
    out[idx] = v;
  }
}

template <int Q>  // channel quads (CK = 4 Q, Q % 4 == 0)
__global__ void __launch_bounds__(256, 3)  // one wavefront per SIMD and workgroup: three workgroups per CU hide the staging
conv_foldn_k(FNArgs a) {
  constexpr int TH = 8, TW = 16, HH = TH + 4, HW = TW + 4, NV = HH * HW;  // 12 x 20 = 240 voxels per plane
  constexpr int NLD = (NV * Q + 255) / 256;                               // 16-byte loads per thread and plane
  constexpr int QG = Q / 4;
  constexpr unsigned kOOB = 0xFFFFFFF0u;
  // plane in LDS as dwords [quad: QS][channel pair of the quad: HS][voxel][2]: ds_read_b32 is serviced in two groups of 32
  // lanes over 32 banks (MI355X_MICROARCH.md, LDS): the half-wave (k = 0, 1 | k = 2, 3) of an A read covers 32 consecutive
  // dwords; ds_write_b64 in four groups of 16 lanes = 8 quads x 2 voxels: QS = 4 (mod 32) puts quad q on banks 4q .. 4q+3.
  // (The first layout, [quad][voxel][4], measured 59 % of the LDS cycles in bank conflicts: a half-wave touched only
  // dwords 0, 1 (mod 4).)
  constexpr int HS = 2 * NV, QS = 2 * HS + 4;
  static_assert(QS % 32 == 4, "bank spreading");
  __shared__ __attribute__((aligned(16))) float lds[Q * QS + 64];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  int t = xcd_remap_fn(blockIdx.x, a.nblk);
  const int twi = t % a.tiles_w;
  t /= a.tiles_w;
  const int thi = t % a.tiles_h;
  t /= a.tiles_h;
  const int seg = t % a.segs;
  const int n = t / a.segs;
  const int h0 = thi * TH, w0 = twi * TW;
  const int d_begin = seg * a.seg_len;
  const int d_end = min(a.D, d_begin + a.seg_len);

  // staging map: 8 consecutive lanes (Q = 8) fetch the quads of one voxel = its whole 128-byte line
  const __amdgpu_buffer_rsrc_t sres = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
  unsigned st_off[NLD];
  int st_lds[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int e = j * 256 + tid;
    const int v = e / Q, q = e % Q;
    const int hh = v / HW, ww = v % HW;
    const int gh = h0 - 2 + hh, gw = w0 - 2 + ww;
    const bool ok = v < NV;
    const bool inb = ok && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
    st_off[j] = inb ? (unsigned)(((((long)n * a.D * a.H + gh) * a.W + gw) * a.sld + 4 * q) * 4) : kOOB;
    st_lds[j] = ok ? q * QS + v * 2 : -1;
  }
  const long plane_bytes = (long)a.H * a.W * a.sld * 4;

  float4 pre[NLD];
  auto fetch = [&](int dp) {
    const bool live = dp >= 0 && dp < a.D;  // uniform
    const unsigned soff = live ? (unsigned)(dp * plane_bytes) : 0u;
#pragma unroll
    for (int j = 0; j < NLD; ++j)
      pre[j] = live ? __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(sres, (int)st_off[j], (int)soff, 0))
                    : make_float4(0.f, 0.f, 0.f, 0.f);
  };

  const float* ldsf = lds;
  const int abase = (lk >> 1) * HS + (2 * wave * HW + li) * 2 + (lk & 1);  // this lane's A element of tap (0, 0), quad 0, first row of the wave
  const float4* wbl = a.wb + lane;

  const int col_co = li / 5, col_kd = li - col_co * 5;
  const bool col_live = li < 5 * a.CN;
  const bool col_first = col_kd == 0 || !col_live;
  f32x4 acc[2];  // the wave's two rows h0 + 2*wave + {0, 1}: two independent MFMA chains sharing every B operand
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) acc[rr] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int steps = (d_end - d_begin) + 4;
  fetch(d_begin - 2);
#pragma unroll 1
  for (int s = 0; s < steps; ++s) {
    const int dp = d_begin - 2 + s;
    const bool live = dp >= 0 && dp < a.D;
    __syncthreads();  // the previous plane's reads are done
    if (live) {
#pragma unroll
      for (int j = 0; j < NLD; ++j)
        if (st_lds[j] >= 0) {
          *reinterpret_cast<float2*>(lds + st_lds[j]) = make_float2(pre[j].x, pre[j].y);
          *reinterpret_cast<float2*>(lds + st_lds[j] + HS) = make_float2(pre[j].z, pre[j].w);
        }
    }
    __syncthreads();
    if (s + 1 < steps) fetch(dp + 1);

    // every partial sum moves one kd to the right (it now belongs to the same output plane seen from plane dp)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float cur = acc[rr][r];  // (a bit_cast applied straight to the vector element reads element 0: clang lvalue bug)
        const int sh = __builtin_amdgcn_update_dpp(0, __float_as_int(cur), 0x111 /* row_shr:1 */, 0xF, 0xF, true);
        acc[rr][r] = col_first ? 0.f : __int_as_float(sh);
      }
    if (live) {
      // A and B operands one (kh, kw) step ahead of their MFMAs: the waits then cover loads issued 16 MFMAs earlier (the
      // compiler's own schedule read LDS just in time).  The packed weights carry one padding tap and the LDS array a
      // padding row for the step behind the last.
      float4 bn[QG];
      float an[2][Q];
#pragma unroll
      for (int g = 0; g < QG; ++g) bn[g] = wbl[g * 64];
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        an[0][q] = ldsf[abase + q * QS];
        an[1][q] = ldsf[abase + q * QS + HW * 2];
      }
#pragma unroll 1
      for (int kh = 0; kh < 5; ++kh) {  // rolled: a fully unrolled plane (400 MFMAs) made the scheduler hoist loads into spills
        const float* arow = ldsf + abase + kh * HW * 2;
        const float4* brow = wbl + kh * 5 * QG * 64;
#pragma unroll
        for (int kw = 0; kw < 5; ++kw) {
          float4 b[QG];
          float av[2][Q];
          const int nxt = kw < 4 ? (kw + 1) * 2 : HW * 2;  // (kh, kw + 1) or (kh + 1, 0)
#pragma unroll
          for (int g = 0; g < QG; ++g) {
            b[g] = bn[g];
            bn[g] = brow[((kw + 1) * QG + g) * 64];
          }
#pragma unroll
          for (int q = 0; q < Q; ++q) {
            av[0][q] = an[0][q];
            av[1][q] = an[1][q];
            an[0][q] = arow[q * QS + nxt];
            an[1][q] = arow[q * QS + nxt + HW * 2];
          }
#pragma unroll
          for (int g = 0; g < QG; ++g) {
            const float bq[4] = {b[g].x, b[g].y, b[g].z, b[g].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][4 * g + e], bq[e], acc[0], 0, 0, 0);
              acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][4 * g + e], bq[e], acc[1], 0, 0, 0);
            }
          }
        }
      }
    }
    // column (co, 4) now holds output plane dp - 2 complete
    const int d = dp - 2;
    if (d >= d_begin && d < d_end && col_live && col_kd == 4) {
      const float bv = a.bias ? a.bias[col_co] : 0.f;
      const float sl = a.prelu ? a.prelu[col_co] : 1.f;
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int gh = h0 + 2 * wave + rr;
        if (gh < a.H) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int gw = w0 + 4 * lk + r;
            if (gw < a.W) {
              float* o = a.dst + ((((long)n * a.D + d) * a.H + gh) * a.W + gw) * a.dld + col_co;
              float v = acc[rr][r] + bv;
              if (a.accumulate) v += *o;
              if (v < 0.f) v *= sl;
              *o = v;
            }
          }
        }
      }
    }
  }
}

}  // namespace

int msk_gconv_halo_foldn(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap, bool* act_fused) {
  if (!(g.kd == 5 && g.kh == 5 && g.kw == 5)) return 0;
  if (!(g.sd == 1 && g.sh == 1 && g.sw == 1 && g.pd == 2 && g.ph == 2 && g.pw == 2)) return 0;
  if (!(g.SD == g.DD && g.SH == g.DH && g.SW == g.DW)) return 0;
  if (!(g.CK == 32 && g.CN >= 1 && 5 * g.CN <= 16)) return 0;
  if (g.DW < 12 || g.DD < 4) return 0;  // narrow slabs keep the VALU kernels
  if (!((g.sld % 4 == 0) && (((uintptr_t)g.src) % 16 == 0))) return 0;
  constexpr int Q = 8, QG = Q / 4;
  const size_t sb = (size_t)g.N * g.SD * g.SH * g.SW * g.sld * sizeof(float);
  if (sb >= 0xFFFFFFF0ull) return 0;
  float* wb = (float*)msk_workspace2(ctx, (size_t)26 * QG * 64 * 4 * sizeof(float));  // + one padding tap (prefetched, never used)
  if (!wb) return -1;
  {
    msk_launch_scope ls(ctx, "pack_weights_foldn");
    hipLaunchKernelGGL(pack_foldn_weights_k, dim3(25 * QG), dim3(256), 0, ctx->stream, w_canon, A, B, swap, g.transposed ? 1 : 0,
                       g.CK, g.CN, QG, wb);
    MSK_LAUNCH_CHECK(ctx);
  }
  FNArgs a{};
  a.src = g.src; a.sld = g.sld; a.dst = g.dst; a.dld = g.dld;
  a.N = g.N; a.D = g.DD; a.H = g.DH; a.W = g.DW; a.CN = g.CN;
  a.wb = (const float4*)wb; a.bias = g.bias; a.prelu = g.prelu; a.accumulate = g.accumulate;
  a.tiles_h = msk_cdiv(a.H, 8); a.tiles_w = msk_cdiv(a.W, 16);
  const long cols = (long)a.N * a.tiles_h * a.tiles_w;
  // D segments: enough workgroups for two per CU, each segment re-walks 4 planes
  const long per_cu = ctx->foldn_wgs > 0 ? ctx->foldn_wgs : 2;  // tuning: option "foldn_wgs"
  int segs = (int)((per_cu * ctx->num_cu + cols - 1) / cols);
  if (segs > a.D / 8) segs = a.D / 8;
  if (segs < 1) segs = 1;
  a.seg_len = msk_cdiv(a.D, segs);
  a.segs = msk_cdiv(a.D, a.seg_len);
  const long nblk = cols * a.segs;
  if (nblk > 0x7fffffff) return 0;
  a.nblk = (int)nblk;
  a.src_bytes = (unsigned)sb;
  const char* tag = "conv_foldn";
  if (ctx->prof && ctx->prof_shapes) {
    char buf[160];
    snprintf(buf, sizeof(buf), "conv_foldn[ck=%d,cn=%d,n=%d,dhw=%dx%dx%d,acc=%d]", g.CK, g.CN, g.N, g.DD, g.DH, g.DW, g.accumulate);
    tag = msk_intern_tag(ctx, buf);
  }
  msk_launch_scope ls(ctx, tag);
  hipLaunchKernelGGL((conv_foldn_k<Q>), dim3((unsigned)nblk), dim3(256), 0, ctx->stream, a);
  MSK_LAUNCH_CHECK(ctx);
  *act_fused = true;
  return 1;
}
