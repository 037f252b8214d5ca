// Strided gather convolution with kernel == stride and no padding (every VNet down-convolution forward, vnet.py:67,98,
// and the data gradient of every up-convolution, :133): each source voxel feeds exactly one destination voxel, the layer
// is a streaming GEMM  dst[m][cn] = sum_{tap, k} src[m*s + tap][k] W[tap][k][cn]  at 13-20 flop/B -- HBM-bound.
//
// The general gather kernel (gconv_gather_mfma_k) ran a load -> MFMA chain per tap with two to four loads in flight per
// wavefront: 1.4-1.8 TB/s at the 128^3 <-> 64^3 level.  Here
//   * K = (tap, 8-channel chunk) is flattened and walked in batches of KB steps with TWO batches of operands in
//     registers: the loads of batch i+1 (x quads and weight quads) are issued before the MFMAs of batch i;
//   * D is produced transposed (weights as the MFMA A operand: rows = output channels, columns = the wavefront's 32
//     destination voxels): a lane owns one voxel -- one address decode -- and four consecutive accumulator rows are four
//     consecutive output channels: 16-byte stores / accumulate loads.
#include "msk_conv.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr unsigned kOOBk = 0xFFFFFFF0u;

template <int NR, int KB>  // NR N tiles (32 output channels each) per workgroup, KB K steps per operand batch
__global__ void __launch_bounds__(256, NR == 1 ? 4 : (NR == 2 ? 4 : 2))
gconv_ks_fwd_k(GConv g, const float4* __restrict__ wm, int KC, int npad, unsigned src_bytes) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const long M = (long)g.N * g.DD * g.DH * g.DW;
  const long m = ((long)blockIdx.x * 4 + wave) * 32 + li;
  if (((long)blockIdx.x * 4 + wave) * 32 >= M) return;
  const bool mok = m < M;
  unsigned ow, oh, od, n;
  {
    const unsigned r = (unsigned)(mok ? m : 0);
    const unsigned t1 = r / (unsigned)g.DW, t2 = t1 / (unsigned)g.DH;
    ow = r - t1 * (unsigned)g.DW;
    oh = t1 - t2 * (unsigned)g.DH;
    n = t2 / (unsigned)g.DD;
    od = t2 - n * (unsigned)g.DD;
  }
  // byte offset of the lane's first source voxel (+ its half of an 8-channel chunk); out of range for idle lanes
  const unsigned xbase = mok ? ((((n * g.SD + od * g.sd) * g.SH + oh * g.sh) * g.SW + ow * g.sw) * (unsigned)g.sld + lh * 4u) * 4u : kOOBk;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g.src, 0, src_bytes, 0x00020000);
  const int nt0 = blockIdx.y * NR;
  const int T = g.kd * g.kh * g.kw * KC;
  const float4* wl = wm + (long)lh * npad + nt0 * 32 + li;

  f32x16 acc[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[r][j] = 0.f;

  float4 xa[KB], xb[KB], wa[KB][NR], wb[KB][NR];
  // running (tap, chunk) of the next K step to load: scalar counters instead of divisions per step
  int lt = 0, lkc = 0, lta = 0, ltb = 0, ltc = 0;
  auto load = [&](float4 (&xv)[KB], float4 (&wv)[KB][NR]) {
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const bool tin = lt < T;  // uniform
      const unsigned soff = (unsigned)((((lta * g.SH + ltb) * g.SW + ltc) * g.sld + lkc * 8) * 4);
      const bool live = tin && lkc * 8 + lh * 4 < g.CK;
      xv[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(live ? xbase : kOOBk), (int)(tin ? soff : 0u), 0));
#pragma unroll
      for (int r = 0; r < NR; ++r) wv[u][r] = tin ? wl[(long)lt * 2 * npad + r * 32] : make_float4(0.f, 0.f, 0.f, 0.f);
      ++lt;
      if (++lkc == KC) {
        lkc = 0;
        if (++ltc == g.kw) {
          ltc = 0;
          if (++ltb == g.kh) {
            ltb = 0;
            ++lta;
          }
        }
      }
    }
  };
  auto compute = [&](const float4 (&xv)[KB], const float4 (&wv)[KB][NR]) {
#pragma unroll
    for (int u = 0; u < KB; ++u)
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[u][r].x, xv[u].x, acc[r], 0, 0, 0);
        acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[u][r].y, xv[u].y, acc[r], 0, 0, 0);
        acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[u][r].z, xv[u].z, acc[r], 0, 0, 0);
        acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[u][r].w, xv[u].w, acc[r], 0, 0, 0);
      }
  };
  load(xa, wa);
  for (int t0 = 0; t0 < T; t0 += 2 * KB) {
    load(xb, wb);
    compute(xa, wa);
    load(xa, wa);
    if (t0 + KB < T) compute(xb, wb);
  }
  if (!mok) return;

  // D[row = cn][col = dst voxel]: 16-byte stores of 4 consecutive output channels
  float* orow = g.dst + m * g.dld;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    float4 old[4];
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cn = (nt0 + r) * 32 + 8 * q + 4 * lh;
      ok[q] = cn < g.CN;  // CN % 4 == 0
      old[q] = (g.accumulate && ok[q]) ? *reinterpret_cast<const float4*>(orow + cn) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (ok[q]) {
        const int cn = (nt0 + r) * 32 + 8 * q + 4 * lh;
        const float4 bv = g.bias ? make_float4(g.bias[cn], g.bias[cn + 1], g.bias[cn + 2], g.bias[cn + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 v;
        v.x = acc[r][4 * q + 0] + bv.x + old[q].x;
        v.y = acc[r][4 * q + 1] + bv.y + old[q].y;
        v.z = acc[r][4 * q + 2] + bv.z + old[q].z;
        v.w = acc[r][4 * q + 3] + bv.w + old[q].w;
        *reinterpret_cast<float4*>(orow + cn) = v;
      }
    }
  }
}

// Round 4 -- the TRANSPOSED gather of the anisotropic MRI levels (up-convolution forward, vnet.py:133, and down-convolution
// data gradient, :98, with kernel (2, 2, 4) / stride (2, 2, 1) or (2, 2, 2) / (2, 2, 1),
// vnet_mri_spine_seg_512_512_12_15k.yml:9-10): kernel == stride along D and H, stride 1 along W, no padding:
//   dst[n, d*sd + a, h*sh + b, W][cn] = bias + sum_{c: 0 <= W - c < SW} sum_k src[n, d, h, W - c][k] * Wt[(a, b, c)][k][cn]
// i.e. per (a, b) parity class a 1-D convolution along W.  Same streaming structure as gconv_ks_fwd_k (D transposed, two
// operand batches in flight, 16-byte stores); a wavefront's 32 destination voxels share their parity class, so the weight
// fragments stay wave-uniform.  The general parity-class kernel took 1.21 / 0.75 ms for the two 512 x 512 x 12 problems.
// PAIR (CN <= 16, kh == sh == 2): the 32 rows of the matrix instruction carry the output channels of BOTH h-parity classes of
// one d-parity -- rows 0-15 class (ra, 0), rows 16-31 class (ra, 1); they read the same source voxels -- instead of 16 channels
// and 16 rows of zero padding: half the matrix instructions and half the source reads (64 -> 16 channels at 512 x 512 x 12:
// the kernel is bound by the fp32 matrix pipe).
template <int NR, int KB, bool PAIR = false>
__global__ void __launch_bounds__(256, NR == 1 ? 4 : (NR == 2 ? 4 : 2))
gconv_kst_k(GConv g, const float4* __restrict__ wm, int KC, int npad, unsigned src_bytes, int waves_per_class) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const long gw = (long)blockIdx.x * 4 + wave;
  const int cls = (int)(gw / waves_per_class);
  if (cls >= (PAIR ? g.sd : g.sd * g.sh)) return;
  const int ra = PAIR ? cls : cls / g.sh, rb = PAIR ? 0 : cls - ra * g.sh;
  const long per_class = (long)g.N * g.SD * g.SH * g.DW;
  const long q = (gw - (long)cls * waves_per_class) * 32 + li;
  const bool mok = q < per_class;
  unsigned W, h, d, n;
  {
    const unsigned r = (unsigned)(mok ? q : 0);
    const unsigned t1 = r / (unsigned)g.DW, t2 = t1 / (unsigned)g.SH;
    W = r - t1 * (unsigned)g.DW;
    h = t1 - t2 * (unsigned)g.SH;
    n = t2 / (unsigned)g.SD;
    d = t2 - n * (unsigned)g.SD;
  }
  const unsigned xrow = (((n * g.SD + d) * g.SH + h) * g.SW) * (unsigned)g.sld + lh * 4u;   // element index of (n, d, h, 0)[lh half]
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g.src, 0, src_bytes, 0x00020000);
  const int nt0 = blockIdx.y * NR;
  const int T = g.kw * KC;
  const float4* wl = PAIR ? wm + ((long)(ra * g.kh + (li >> 4)) * g.kw * KC) * 2 * npad + (long)lh * npad + (li & 15)
                          : wm + ((long)(ra * g.kh + rb) * g.kw * KC) * 2 * npad + (long)lh * npad + nt0 * 32 + li;

  f32x16 acc[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[r][j] = 0.f;

  float4 xa[KB], xb[KB], wa[KB][NR], wb[KB][NR];
  int lt = 0, lkc = 0, ltc = 0;
  auto load = [&](float4 (&xv)[KB], float4 (&wv)[KB][NR]) {
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const bool tin = lt < T;  // uniform
      const int sw_ = (int)W - ltc;   // source position along W of tap ltc
      const bool live = tin && mok && sw_ >= 0 && sw_ < g.SW && lkc * 8 + lh * 4 < g.CK;
      const unsigned voff = live ? (xrow + (unsigned)sw_ * (unsigned)g.sld) * 4u : kOOBk;
      xv[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)(tin ? lkc * 32 : 0), 0));
#pragma unroll
      for (int r = 0; r < NR; ++r) wv[u][r] = tin ? wl[(long)lt * 2 * npad + r * 32] : make_float4(0.f, 0.f, 0.f, 0.f);
      ++lt;
      if (++lkc == KC) {
        lkc = 0;
        ++ltc;
      }
    }
  };
  auto compute = [&](const float4 (&xv)[KB], const float4 (&wv)[KB][NR]) {
#pragma unroll
    for (int u = 0; u < KB; ++u)
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[u][r].x, xv[u].x, acc[r], 0, 0, 0);
        acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[u][r].y, xv[u].y, acc[r], 0, 0, 0);
        acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[u][r].z, xv[u].z, acc[r], 0, 0, 0);
        acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[u][r].w, xv[u].w, acc[r], 0, 0, 0);
      }
  };
  load(xa, wa);
  for (int t0 = 0; t0 < T; t0 += 2 * KB) {
    load(xb, wb);
    compute(xa, wa);
    load(xa, wa);
    if (t0 + KB < T) compute(xb, wb);
  }
  if (!mok) return;

  const long dvox = (((long)n * g.DD + d * g.sd + ra) * g.DH + h * g.sh + rb) * g.DW + W;
  float* orow0 = g.dst + dvox * g.dld;
  const long pair_step = (long)g.DW * g.dld;   // PAIR: the voxel of class (ra, 1) is one destination row further
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    float4 old[4];
    bool ok[4];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int cn = PAIR ? 8 * (qd & 1) + 4 * lh : (nt0 + r) * 32 + 8 * qd + 4 * lh;
      const float* orow = PAIR ? orow0 + (qd >> 1) * pair_step : orow0;
      ok[qd] = cn < g.CN;  // CN % 4 == 0
      old[qd] = (g.accumulate && ok[qd]) ? *reinterpret_cast<const float4*>(orow + cn) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      if (ok[qd]) {
        const int cn = PAIR ? 8 * (qd & 1) + 4 * lh : (nt0 + r) * 32 + 8 * qd + 4 * lh;
        float* orow = PAIR ? orow0 + (qd >> 1) * pair_step : orow0;
        const float4 bv = g.bias ? make_float4(g.bias[cn], g.bias[cn + 1], g.bias[cn + 2], g.bias[cn + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 v;
        v.x = acc[r][4 * qd + 0] + bv.x + old[qd].x;
        v.y = acc[r][4 * qd + 1] + bv.y + old[qd].y;
        v.z = acc[r][4 * qd + 2] + bv.z + old[qd].z;
        v.w = acc[r][4 * qd + 3] + bv.w + old[qd].w;
        *reinterpret_cast<float4*>(orow + cn) = v;
      }
    }
  }
}

}  // namespace

// transposed gather, kernel == stride along D and H, stride 1 along W (see gconv_kst_k); 1 handled, 0 not eligible, < 0 error
int msk_gconv_kst(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap) {
  if (!g.transposed) return 0;
  if (!(g.pd == 0 && g.ph == 0 && g.pw == 0 && g.kd == g.sd && g.kh == g.sh && g.sw == 1 && g.kw >= 2)) return 0;
  if (!(g.DD == g.SD * g.sd && g.DH == g.SH * g.sh && g.DW == g.SW + g.kw - 1)) return 0;
  if (g.CK % 4 || g.CN % 4 || g.sld % 4 || g.dld % 4 || ((uintptr_t)g.src) % 16 || ((uintptr_t)g.dst) % 16) return 0;
  const size_t sbytes = (size_t)g.N * g.SD * g.SH * g.SW * g.sld * sizeof(float);
  const long per_class = (long)g.N * g.SD * g.SH * g.DW;
  if (sbytes >= 0xFFFFFFF0ull || per_class >= (1L << 31) || (long)g.N * g.DD * g.DH * g.DW >= (1L << 31)) return 0;
  const int taps = g.kd * g.kh * g.kw;
  const int KC = (g.CK + 7) / 8;
  const int npad = ((g.CN + 31) / 32) * 32;
  const float* wm = msk_pack_weights_get(ctx, w_canon, A, B, taps, swap, 0, g.kd, g.kh, g.kw, 1, g.CK, g.CN, KC, npad);
  if (!wm) return -1;
  const int ntn = npad / 32;
  const long wpc = (per_class + 31) / 32;                 // wavefronts per parity class
  const bool pair = g.CN <= 16 && g.kh == 2 && g.sh == 2 && ctx->kst_pair;
  const long blocks = (wpc * g.sd * (pair ? 1 : g.sh) + 3) / 4;
  int NR = ntn % 4 == 0 ? 4 : (ntn % 2 == 0 ? 2 : 1);
  while (NR > ctx->ks_nr_max) NR >>= 1;
  while (NR > 1 && blocks * (ntn / NR) < 2L * ctx->num_cu) NR >>= 1;
  const char* tag = "gconv_kst";
  if (ctx->prof && ctx->prof_shapes) {
    char buf[160];
    snprintf(buf, sizeof(buf), "gconv_kst[ck=%d,cn=%d,k=%dx%dx%d,dst=%dx%dx%dx%d,sld=%d,dld=%d,acc=%d,nr=%d]", g.CK, g.CN, g.kd, g.kh, g.kw, g.N, g.DD, g.DH, g.DW,
             g.sld, g.dld, g.accumulate, NR);
    tag = msk_intern_tag(ctx, buf);
  }
  msk_launch_scope ls(ctx, tag);
  dim3 grid((unsigned)blocks, ntn / NR);
  const float4* w4 = reinterpret_cast<const float4*>(wm);
  if (pair) {
    hipLaunchKernelGGL((gconv_kst_k<1, 4, true>), grid, dim3(256), 0, ctx->stream, g, w4, KC, npad, (unsigned)sbytes, (int)wpc);
    MSK_LAUNCH_CHECK(ctx);
    return 1;
  }
  switch (NR) {
    case 4: hipLaunchKernelGGL((gconv_kst_k<4, 2>), grid, dim3(256), 0, ctx->stream, g, w4, KC, npad, (unsigned)sbytes, (int)wpc); break;
    case 2: hipLaunchKernelGGL((gconv_kst_k<2, 2>), grid, dim3(256), 0, ctx->stream, g, w4, KC, npad, (unsigned)sbytes, (int)wpc); break;
    default: hipLaunchKernelGGL((gconv_kst_k<1, 4>), grid, dim3(256), 0, ctx->stream, g, w4, KC, npad, (unsigned)sbytes, (int)wpc); break;
  }
  MSK_LAUNCH_CHECK(ctx);
  return 1;
}

int msk_gconv_ks_fwd(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap) {
  if (g.transposed) return 0;
  // unpadded windows that lie inside the source.  kernel == stride is the VNet case (every source voxel read once); round 4: any
  // stride -- the anisotropic MRI levels (kernel (2, 2, 4) / stride (2, 2, 1) and (2, 2, 2) / (2, 2, 1),
  // vnet_mri_spine_seg_512_512_12_15k.yml:9-10) overlap along W only, the kernel's addressing never assumed otherwise
  if (!(g.pd == 0 && g.ph == 0 && g.pw == 0)) return 0;
  if ((g.DD - 1) * g.sd + g.kd > g.SD || (g.DH - 1) * g.sh + g.kh > g.SH || (g.DW - 1) * g.sw + g.kw > g.SW) return 0;
  const int taps = g.kd * g.kh * g.kw;
  if (taps < 2 || (g.kd == 5 && g.kh == 5 && g.kw == 5)) return 0;
  if (g.CK % 4 || g.CN % 4 || g.sld % 4 || g.dld % 4 || ((uintptr_t)g.src) % 16 || ((uintptr_t)g.dst) % 16) return 0;
  const size_t sbytes = (size_t)g.N * g.SD * g.SH * g.SW * g.sld * sizeof(float);
  const long M = (long)g.N * g.DD * g.DH * g.DW;
  if (sbytes >= 0xFFFFFFF0ull || M >= (1L << 31)) return 0;
  const int KC = (g.CK + 7) / 8;
  const int npad = ((g.CN + 31) / 32) * 32;
  const float* wm = msk_pack_weights_get(ctx, w_canon, A, B, taps, swap, 0, g.kd, g.kh, g.kw, 1, g.CK, g.CN, KC, npad);
  if (!wm) return -1;
  const int ntn = npad / 32;
  const long mtiles = (M + 127) / 128;
  // N tiles per workgroup: as many as divide ntn (x is then read once), fewer when the grid would not fill the GPU
  int NR = ntn % 4 == 0 ? 4 : (ntn % 2 == 0 ? 2 : 1);
  while (NR > ctx->ks_nr_max) NR >>= 1;
  while (NR > 1 && mtiles * (ntn / NR) < 2L * ctx->num_cu) NR >>= 1;
  const char* tag = "gconv_ks_fwd";
  if (ctx->prof && ctx->prof_shapes) {
    char buf[160];
    snprintf(buf, sizeof(buf), "gconv_ks_fwd[ck=%d,cn=%d,k=%dx%dx%d,dst=%dx%dx%dx%d,sld=%d,dld=%d,acc=%d,nr=%d]", g.CK, g.CN, g.kd, g.kh, g.kw, g.N, g.DD, g.DH,
             g.DW, g.sld, g.dld, g.accumulate, NR);
    tag = msk_intern_tag(ctx, buf);
  }
  msk_launch_scope ls(ctx, tag);
  dim3 grid((unsigned)mtiles, ntn / NR);
  const float4* w4 = reinterpret_cast<const float4*>(wm);
  switch (NR) {
    case 4: hipLaunchKernelGGL((gconv_ks_fwd_k<4, 2>), grid, dim3(256), 0, ctx->stream, g, w4, KC, npad, (unsigned)sbytes); break;
    case 2: hipLaunchKernelGGL((gconv_ks_fwd_k<2, 2>), grid, dim3(256), 0, ctx->stream, g, w4, KC, npad, (unsigned)sbytes); break;
    default: hipLaunchKernelGGL((gconv_ks_fwd_k<1, 4>), grid, dim3(256), 0, ctx->stream, g, w4, KC, npad, (unsigned)sbytes); break;
  }
  MSK_LAUNCH_CHECK(ctx);
  return 1;
}
