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
#ifdef KS_PROBE_NOMFMA   // knock-out probe (timing only, wrong results): the fp32 matrix instructions of this file become register moves
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a_, b_, c_, x_, y_, z_) ks_probe_keep((a_), (b_), (c_))
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a_, b_, c_, x_, y_, z_) ks_probe_keep((a_), (b_), (c_))
template <typename T> __device__ __forceinline__ T ks_probe_keep(float a, float b, T c) { asm volatile("" ::"v"(a), "v"(b)); c[0] += a * 1e-30f + b * 1e-30f; return c; }
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr unsigned kOOBk = 0xFFFFFFF0u;

// FUSE (round 5, msk_convT3d_bwd_bnact): the source is the gradient dy behind a BatchNorm + PReLU that nobody has written --
// the data gradient of an up-convolution (vnet.py:133-150) evaluates  dy = scale (du - s1 - xhat s2),  du = dout prelu'(scale y +
// shift),  from (y, dout) in the registers that feed the matrix instructions, with the arithmetic of affine_act_bwd_apply_cs_k.
// The pass that used to write dy on the compute stream (0.20 ms at 16ch@128^3) moves to the weight-gradient stream, where
// the weight gradient is its only other reader.  CK <= 16 (two 8-channel chunks): a lane's coefficients live in registers.
struct KsBnBwd {
  const float* y;      // convolution output (pre-BatchNorm), voxel stride yld
  const float* dout;   // gradient w.r.t. the unit's output, voxel stride dld
  int yld, dld;
  unsigned y_bytes, d_bytes;
  const float *scale, *shift, *alpha, *mean, *invstd, *sums;   // [CK] each, sums [2 CK]
  float invM;
};

template <int NR, int KB, bool FUSE = false>  // NR N tiles (32 output channels each) per workgroup, KB K steps per operand batch
__global__ void __launch_bounds__(256, FUSE ? 2 : (NR == 1 ? 4 : (NR == 2 ? 4 : 2)))
gconv_ks_fwd_k(GConv g, const float4* __restrict__ wm, int KC, int npad, unsigned src_bytes, KsBnBwd bn) {
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
  const unsigned vox0 = ((n * g.SD + od * g.sd) * g.SH + oh * g.sh) * g.SW + ow * g.sw;   // the lane's first source voxel
  const unsigned xbase = mok ? (vox0 * (unsigned)g.sld + lh * 4u) * 4u : kOOBk;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(FUSE ? bn.y : g.src), 0, FUSE ? bn.y_bytes : src_bytes, 0x00020000);
  // FUSE: y through rs (stride yld), dout through rd (stride dld); coefficients of the lane's channels kc * 8 + lh * 4 + j
  const unsigned ybase = mok ? (vox0 * (unsigned)bn.yld + lh * 4u) * 4u : kOOBk;
  const unsigned dbase = mok ? (vox0 * (unsigned)bn.dld + lh * 4u) * 4u : kOOBk;
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(FUSE ? bn.dout : g.src), 0, FUSE ? bn.d_bytes : src_bytes, 0x00020000);
  float c_sc[2][4], c_sf[2][4], c_al[2][4], c_mu[2][4], c_is[2][4], c_s1[2][4], c_s2[2][4];
  if constexpr (FUSE) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = k * 8 + lh * 4 + j;
        const bool ok = c < g.CK;
        c_sc[k][j] = ok ? bn.scale[c] : 0.f;
        c_sf[k][j] = ok ? bn.shift[c] : 0.f;
        c_al[k][j] = (ok && bn.alpha) ? bn.alpha[c] : 1.f;
        c_mu[k][j] = ok ? bn.mean[c] : 0.f;
        c_is[k][j] = ok ? bn.invstd[c] : 0.f;
        c_s1[k][j] = ok ? bn.sums[c] * bn.invM : 0.f;
        c_s2[k][j] = ok ? bn.sums[g.CK + c] * bn.invM : 0.f;
      }
  }
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
      if constexpr (FUSE) {
        const unsigned tvox = (unsigned)((lta * g.SH + ltb) * g.SW + ltc);
        const float4 yq = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(live ? ybase : kOOBk), (int)(tin ? (tvox * bn.yld + lkc * 8) * 4u : 0u), 0));
        const float4 dq = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rd, (int)(live ? dbase : kOOBk), (int)(tin ? (tvox * bn.dld + lkc * 8) * 4u : 0u), 0));
        const float yv[4] = {yq.x, yq.y, yq.z, yq.w}, dv[4] = {dq.x, dq.y, dq.z, dq.w};
        const float on = live ? 1.f : 0.f;   // (a multiply, not a select: the loads stay unconditional and batched)
        const int k = lkc & 1;               // KC <= 2; the unrolled batch alternates the chunk
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float sc = k ? c_sc[1][j] : c_sc[0][j], sf = k ? c_sf[1][j] : c_sf[0][j], al = k ? c_al[1][j] : c_al[0][j];
          const float mu = k ? c_mu[1][j] : c_mu[0][j], is = k ? c_is[1][j] : c_is[0][j];
          const float s1 = k ? c_s1[1][j] : c_s1[0][j], s2 = k ? c_s2[1][j] : c_s2[0][j];
          float d = dv[j];
          if (bn.alpha) {
            const float uu = fmaf(yv[j], sc, sf);
            if (!(uu > 0.f)) d *= al;
          }
          const float xh = (yv[j] - mu) * is;
          o[j] = on * (sc * (d - s1 - xh * s2));
        }
        xv[u] = make_float4(o[0], o[1], o[2], o[3]);
      } else {
        xv[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(live ? xbase : kOOBk), (int)(tin ? soff : 0u), 0));
      }
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

// Round 5 -- the 2 x 2 x 2 / stride 2 problems with <= 16 source channels (16ch@128^3: the down-convolution of down_tr32,
// vnet.py:98, and the data gradient of up_tr32's up-convolution, :133) through an LDS tile with WHOLE-LINE loads.  Above, a lane
// fetches 16 bytes of ITS source voxel per tap: 32 lines per load instruction, each 128-byte line consumed 32 bytes at a time
// over four instructions -- 2.0 TB/s (164 / 150 us for 335 / 403 MB).  Here a wavefront owns a run of 32 destination voxels
// along W and stages, per kd plane, the two source rows (kh = 0, 1) of that run -- 64 consecutive voxels each, one contiguous
// 4 KiB stretch of a dense 16-channel tensor -- with 16-byte loads of consecutive lanes (1 KiB per instruction).  The matrix
// operands then come from LDS: voxel PAIRS at a pitch of 2 CK floats + 16 bytes, so that the 16 lanes of a ds_read_b128 group
// (lane stride = one pair) fall on 16 distinct 16-byte bank groups.  The region is private to the wavefront: no barriers; the
// second plane's global loads are in flight while the first plane's matrix instructions run.
// FUSE: the source is dy behind a BatchNorm + PReLU evaluated from (y, dout) when the tile is staged (msk_convT3d_bwd_bnact):
// a lane stages the same channel quad in every load (64 % (CK / 4) == 0), so its 28 coefficients are registers.
#ifndef KS_DBG
#define KS_DBG 0   // tools/ab_build.sh ... -DKS_DBG=n: 1 no matrix instructions, 2 no weight loads, 4 no stores, 8 no source loads (what bounds the kernel)
#endif
template <int NR, int CK, bool FUSE>
__global__ void __launch_bounds__(256, 3)
gconv_ks_lds_k(GConv g, const float4* __restrict__ wm, int npad, KsBnBwd bn, int runs) {
  constexpr int KC = CK / 8, QV = CK / 4;        // 8-channel chunks, 16-byte quads per voxel
  constexpr int PP = 2 * CK * 4 + 16;            // pair pitch (bytes)
  constexpr int ROWB = 32 * PP;                  // one source row of a run: 32 voxel pairs
  constexpr int NL = 64 * QV / 64;               // 16-byte loads per lane and row (= QV)
  __shared__ __attribute__((aligned(16))) char lds[4 * 2 * ROWB];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int rid = blockIdx.x * 4 + wave;
  if (rid >= runs) return;                       // wave-uniform; no block-wide barrier below
  char* reg = lds + wave * 2 * ROWB;
  const int rpr = g.DW / 32;                     // runs per destination row
  int t = rid / rpr;
  const int ow0 = (rid - t * rpr) * 32;
  const int oh = t % g.DH;
  t /= g.DH;
  const int od = t % g.DD, n = t / g.DD;
  const long m = (((long)n * g.DD + od) * g.DH + oh) * g.DW + ow0 + li;   // the lane's destination voxel
  const int nt0 = blockIdx.y * NR;
  const float4* wl = wm + (long)lh * npad + nt0 * 32 + li;

  // staging: chunk q = lane + 64 i  ->  voxel q / QV of the run's 64 source voxels, quad q % QV = lane % QV
  const int jq = lane % QV;
  float c_sc[4], c_sf[4], c_al[4], c_mu[4], c_is[4], c_s1[4], c_s2[4];
  if constexpr (FUSE) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = jq * 4 + j;
      c_sc[j] = bn.scale[c];
      c_sf[j] = bn.shift[c];
      c_al[j] = bn.alpha ? bn.alpha[c] : 1.f;
      c_mu[j] = bn.mean[c];
      c_is[j] = bn.invstd[c];
      c_s1[j] = bn.sums[c] * bn.invM;
      c_s2[j] = bn.sums[CK + c] * bn.invM;
    }
  }
  const int sld = FUSE ? bn.yld : g.sld;
  const float* sp = FUSE ? bn.y : g.src;
  float4 xa[2][NL], xd[2][NL];
  auto issue = [&](int a) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const long v0 = (((long)n * g.SD + 2 * od + a) * g.SH + 2 * oh + b) * g.SW + 2 * ow0;   // first source voxel of the row
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int q = lane + 64 * i, v = q / QV;
        xa[b][i] = (KS_DBG & 8) ? make_float4((float)v, 1.f, 2.f, 3.f) : *reinterpret_cast<const float4*>(sp + (v0 + v) * sld + jq * 4);
        if constexpr (FUSE) xd[b][i] = *reinterpret_cast<const float4*>(bn.dout + (v0 + v) * bn.dld + jq * 4);
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int q = lane + 64 * i, v = q / QV;
        float4 o = xa[b][i];
        if constexpr (FUSE) {
          const float yv[4] = {o.x, o.y, o.z, o.w}, dv[4] = {xd[b][i].x, xd[b][i].y, xd[b][i].z, xd[b][i].w};
          float r[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float d = dv[j];
            if (bn.alpha) {
              const float uu = fmaf(yv[j], c_sc[j], c_sf[j]);
              if (!(uu > 0.f)) d *= c_al[j];
            }
            const float xh = (yv[j] - c_mu[j]) * c_is[j];
            r[j] = c_sc[j] * (d - c_s1[j] - xh * c_s2[j]);
          }
          o = make_float4(r[0], r[1], r[2], r[3]);
        }
        *reinterpret_cast<float4*>(reg + b * ROWB + (v >> 1) * PP + (v & 1) * (CK * 4) + jq * 16) = o;
      }
  };

  f32x16 acc[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[r][j] = 0.f;

  issue(0);
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    commit();                       // (waits for the plane's loads; the previous plane's LDS reads have been consumed)
    if (a == 0) issue(1);           // in flight under the matrix instructions below
    // (LDS operations of one wavefront execute in order: its reads below see the tile, the next commit() follows its reads)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const int tap = (a * 2 + b) * 2 + c;
          const float4 xq = *reinterpret_cast<const float4*>(reg + b * ROWB + li * PP + c * (CK * 4) + kc * 32 + lh * 16);
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const float4 wq = (KS_DBG & 2) ? make_float4((float)(tap + kc + r), 0.5f, 0.25f, 2.f) : wl[(long)(tap * KC + kc) * 2 * npad + r * 32];
            if (KS_DBG & 1) {
              acc[r][(tap + kc) & 15] += wq.x * xq.x + wq.y * xq.y + wq.z * xq.z + wq.w * xq.w;
              continue;
            }
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq.x, xq.x, acc[r], 0, 0, 0);
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq.y, xq.y, acc[r], 0, 0, 0);
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq.z, xq.z, acc[r], 0, 0, 0);
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq.w, xq.w, acc[r], 0, 0, 0);
          }
        }
  }

  // D[row = cn][col = dst voxel]: 16-byte stores of 4 consecutive output channels (as gconv_ks_fwd_k)
  float* orow = g.dst + m * g.dld;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    float4 old[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cn = (nt0 + r) * 32 + 8 * q + 4 * lh;
      old[q] = g.accumulate ? *reinterpret_cast<const float4*>(orow + cn) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cn = (nt0 + r) * 32 + 8 * q + 4 * lh;
      const float4 bv = g.bias ? make_float4(g.bias[cn], g.bias[cn + 1], g.bias[cn + 2], g.bias[cn + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 v;
      v.x = acc[r][4 * q + 0] + bv.x + old[q].x;
      v.y = acc[r][4 * q + 1] + bv.y + old[q].y;
      v.z = acc[r][4 * q + 2] + bv.z + old[q].z;
      v.w = acc[r][4 * q + 3] + bv.w + old[q].w;
      if (!(KS_DBG & 4) || v.x == 12345.678f) *reinterpret_cast<float4*>(orow + cn) = v;
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

static int gconv_ks_fwd_impl(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap, const KsBnBwd* bnp, bool dry = false);
int msk_gconv_ks_fwd(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap) {
  return gconv_ks_fwd_impl(ctx, g, w_canon, A, B, swap, nullptr);
}
// the same convolution with the source evaluated on the fly from (y, dout) behind a BatchNorm + PReLU (KsBnBwd; g.src / g.sld are
// ignored): 1 handled, 0 not eligible (CK > 16, strides, alignment), < 0 error
int msk_gconv_ks_fwd_bnbwd(msk_ctx* ctx, const GConv& g0, const float* w_canon, int A, int B, int swap, const float* y, int yld,
                           const float* dout, int dld, const float* scale, const float* shift, const float* alpha, const float* mean,
                           const float* invstd, const float* sums, double M_total, bool dry) {
  if (g0.CK > 16 || yld % 4 || dld % 4 || ((uintptr_t)y) % 16 || ((uintptr_t)dout) % 16) return 0;
  const size_t vox = (size_t)g0.N * g0.SD * g0.SH * g0.SW;
  if (vox * yld * sizeof(float) >= 0xFFFFFFF0ull || vox * dld * sizeof(float) >= 0xFFFFFFF0ull) return 0;
  KsBnBwd bn{};
  bn.y = y; bn.dout = dout; bn.yld = yld; bn.dld = dld;
  bn.y_bytes = (unsigned)(vox * yld * sizeof(float)); bn.d_bytes = (unsigned)(vox * dld * sizeof(float));
  bn.scale = scale; bn.shift = shift; bn.alpha = alpha; bn.mean = mean; bn.invstd = invstd; bn.sums = sums;
  bn.invM = (float)(1.0 / M_total);
  GConv g = g0;
  g.src = y; g.sld = yld;
  return gconv_ks_fwd_impl(ctx, g, w_canon, A, B, swap, &bn, dry);
}
static int gconv_ks_fwd_impl(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap, const KsBnBwd* bnp, bool dry) {
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
  if (dry) return 1;   // every eligibility test passed; nothing launched
  const int KC = (g.CK + 7) / 8;
  const int npad = ((g.CN + 31) / 32) * 32;
  const float* wm = msk_pack_weights_get(ctx, w_canon, A, B, taps, swap, 0, g.kd, g.kh, g.kw, 1, g.CK, g.CN, KC, npad);
  if (!wm) return -1;
  const int ntn = npad / 32;
  if (ctx->ks_lds && g.kd == 2 && g.kh == 2 && g.kw == 2 && g.sd == 2 && g.sh == 2 && g.sw == 2 && g.SD == 2 * g.DD && g.SH == 2 * g.DH &&
      g.SW == 2 * g.DW && g.DW % 32 == 0 && (g.CK == 8 || g.CK == 16) && g.CN % 32 == 0 && (!g.bias || ((uintptr_t)g.bias) % 16 == 0)) {
    // LDS-staged form (gconv_ks_lds_k): whole-line loads of the source rows
    const int runs = (int)(M / 32);
    const int NRl = ntn % 2 == 0 ? 2 : 1;
    const char* tag = bnp ? "gconv_ks_lds_bnbwd" : "gconv_ks_lds";
    if (ctx->prof && ctx->prof_shapes) {
      char buf[160];
      snprintf(buf, sizeof(buf), "%s[ck=%d,cn=%d,dst=%dx%dx%dx%d,sld=%d,dld=%d,acc=%d,nr=%d]", tag, g.CK, g.CN, g.N, g.DD, g.DH, g.DW, g.sld, g.dld,
               g.accumulate, NRl);
      tag = msk_intern_tag(ctx, buf);
    }
    msk_launch_scope ls(ctx, tag);
    const dim3 grid((unsigned)((runs + 3) / 4), ntn / NRl);
    const float4* w4 = reinterpret_cast<const float4*>(wm);
    const KsBnBwd bnv = bnp ? *bnp : KsBnBwd{};
#define KS_LDS(NR_, CK_) \
    do { \
      if (bnp) hipLaunchKernelGGL((gconv_ks_lds_k<NR_, CK_, true>), grid, dim3(256), 0, ctx->stream, g, w4, npad, bnv, runs); \
      else hipLaunchKernelGGL((gconv_ks_lds_k<NR_, CK_, false>), grid, dim3(256), 0, ctx->stream, g, w4, npad, bnv, runs); \
    } while (0)
    if (g.CK == 16) { if (NRl == 2) KS_LDS(2, 16); else KS_LDS(1, 16); }
    else { if (NRl == 2) KS_LDS(2, 8); else KS_LDS(1, 8); }
#undef KS_LDS
    MSK_LAUNCH_CHECK(ctx);
    return 1;
  }
  const long mtiles = (M + 127) / 128;
  // N tiles per workgroup: as many as divide ntn (x is then read once), fewer when the grid would not fill the GPU
  int NR = ntn % 4 == 0 ? 4 : (ntn % 2 == 0 ? 2 : 1);
  while (NR > ctx->ks_nr_max) NR >>= 1;
  while (NR > 1 && mtiles * (ntn / NR) < 2L * ctx->num_cu) NR >>= 1;
  if (bnp && NR > 2) NR = 2;   // the fused-source form carries 56 coefficient registers: two N tiles (226 registers, no spills)
  const char* tag = bnp ? "gconv_ks_fwd_bnbwd" : "gconv_ks_fwd";
  if (ctx->prof && ctx->prof_shapes) {
    char buf[160];
    snprintf(buf, sizeof(buf), "%s[ck=%d,cn=%d,k=%dx%dx%d,dst=%dx%dx%dx%d,sld=%d,dld=%d,acc=%d,nr=%d]", tag, g.CK, g.CN, g.kd, g.kh, g.kw, g.N, g.DD, g.DH,
             g.DW, g.sld, g.dld, g.accumulate, NR);
    tag = msk_intern_tag(ctx, buf);
  }
  msk_launch_scope ls(ctx, tag);
  dim3 grid((unsigned)mtiles, ntn / NR);
  const float4* w4 = reinterpret_cast<const float4*>(wm);
  switch (NR) {
    case 4:
      hipLaunchKernelGGL((gconv_ks_fwd_k<4, 2>), grid, dim3(256), 0, ctx->stream, g, w4, KC, npad, (unsigned)sbytes, KsBnBwd{});
      break;
    case 2:
      if (bnp) hipLaunchKernelGGL((gconv_ks_fwd_k<2, 2, true>), grid, dim3(256), 0, ctx->stream, g, w4, KC, npad, (unsigned)sbytes, *bnp);
      else hipLaunchKernelGGL((gconv_ks_fwd_k<2, 2>), grid, dim3(256), 0, ctx->stream, g, w4, KC, npad, (unsigned)sbytes, KsBnBwd{});
      break;
    default:
      if (bnp) hipLaunchKernelGGL((gconv_ks_fwd_k<1, 2, true>), grid, dim3(256), 0, ctx->stream, g, w4, KC, npad, (unsigned)sbytes, *bnp);
      else hipLaunchKernelGGL((gconv_ks_fwd_k<1, 4>), grid, dim3(256), 0, ctx->stream, g, w4, KC, npad, (unsigned)sbytes, KsBnBwd{});
      break;
  }
  MSK_LAUNCH_CHECK(ctx);
  return 1;
}
