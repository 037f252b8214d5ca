// Weight gradient of a 'same' 5^3 convolution with a TINY output-channel count (out_tr.conv1, 32 -> ncls <= 4,
// vnet.py:165):   dW[cb][ca][tap] = sum_u x[u][ca] * dy[u - tap + P][cb].
// The generic folded kernel (wgrad_fold_mfma_k<false>) gathers dy straight from global memory with one dword per lane
// and (tap, cb) pair: PMC MFMA pipe 41 % busy, 1.65 ms for 2 x 128^3.  Here, like wgrad_c1_mfma_k, the small-channel
// tensor lives in LDS: the dy halo of a 4 x 8 x 32 voxel tile as CB planes (CB x 13.8 KB), MFMA rows = (tap, cb) pairs
// (v_mfma_f32_16x16x4_f32: 16 pairs x 16 input channels x 4 voxels), the A operand is one ds_read_b32 at
// (voxel - tap offset) of the pair's plane, the B operand one coalesced dword of x (16 channels of 4 consecutive voxels).
// The four wavefronts of a workgroup own disjoint row tiles and all walk the whole voxel tile (x re-reads hit L1);
// workgroups are persistent and write one partial slab each.
#include "msk_conv.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOOBc = 0xFFFFFFF0u;

__device__ __forceinline__ float cbs_load(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0));
}

// RTW = row tiles per wavefront, CT = column tiles (CA / 16)
template <int KS, int RTW, int CT>
__global__ void __launch_bounds__(256)
wgrad_cbs_mfma_k(WGrad g, int ntiles, int tiles_d, int tiles_h, int tiles_w, float* __restrict__ partial, unsigned a_bytes,
                 unsigned b_bytes) {
  constexpr int TD = 4, TH = 8, TW = 32, P = KS / 2;
  constexpr int HD = TD + 2 * P, HH = TH + 2 * P, HW = TW + 2 * P;
  constexpr int NV = HD * HH * HW, NVP = NV + 1;  // odd plane pitch
  constexpr int TAPS = KS * KS * KS;
  extern __shared__ float ys[];                    // [CB][NVP]

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 15, kq = lane >> 4;
  const int D = g.BD, H = g.BH, W = g.BW, CB = g.CB;
  const int Q = TAPS * CB;                         // (tap, cb) pairs, pair q = tap * CB + cb
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, b_bytes, 0x00020000);

  // this lane's pairs: LDS offset relative to the halo index of the CENTRE (u + P in every axis):
  //   plane(cb) * NVP - ((kd - P) * HH + (kh - P)) * HW - (kw - P)
  int po[RTW];
#pragma unroll
  for (int t = 0; t < RTW; ++t) {
    const int q = (wave * RTW + t) * 16 + r;
    if (q < Q) {
      const int tap = q / CB, cb = q - tap * CB;
      const int kd = tap / (KS * KS), kh = (tap / KS) % KS, kw = tap % KS;
      po[t] = cb * NVP - (((kd - P) * HH + (kh - P)) * HW + (kw - P));
    } else {
      po[t] = 0;                                   // padding pair: reads something valid, its rows are never stored
    }
  }

  f32x4 acc[RTW][CT];
#pragma unroll
  for (int t = 0; t < RTW; ++t)
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int t_ = tile;
    const int twi = t_ % tiles_w;
    t_ /= tiles_w;
    const int thi = t_ % tiles_h;
    t_ /= tiles_h;
    const int tdi = t_ % tiles_d;
    const int n = t_ / tiles_d;
    const int d0 = tdi * TD, h0 = thi * TH, w0 = twi * TW;
    __syncthreads();
    for (int hv = tid; hv < NV; hv += 256) {
      const int hd = hv / (HH * HW), rem = hv % (HH * HW), hh = rem / HW, hw = rem % HW;
      const int gd = d0 - P + hd, gh = h0 - P + hh, gw = w0 - P + hw;
      const bool in = (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
      const unsigned off = in ? (unsigned)((((n * D + gd) * H + gh) * W + gw) * g.bld) * 4u : kOOBc;
      for (int cb = 0; cb < CB; ++cb) ys[cb * NVP + hv] = cbs_load(rb, off == kOOBc ? kOOBc : off + 4u * cb);
    }
    __syncthreads();
    constexpr int UB = 8;
#pragma unroll 1
    for (int s0 = 0; s0 < TD * TH * (TW / 4); s0 += UB) {
      float xv[UB][CT];
      int ctr[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int s = s0 + u;
        const int dz = s / (TH * (TW / 4)), rem = s % (TH * (TW / 4)), h = rem / (TW / 4), w = (rem % (TW / 4)) * 4 + kq;
        const int gd = d0 + dz, gh = h0 + h, gw = w0 + w;
        const bool vok = gd < D && gh < H && gw < W;
        ctr[u] = ((dz + P) * HH + h + P) * HW + w + P;  // halo index of the centre of voxel u
        const unsigned xo = (unsigned)((((n * D + gd) * H + gh) * W + gw) * g.ald) * 4u;
#pragma unroll
        for (int c = 0; c < CT; ++c)
          xv[u][c] = cbs_load(ra, (vok && c * 16 + r < g.CA) ? xo + (unsigned)(c * 16 + r) * 4u : kOOBc);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u)
#pragma unroll
        for (int t = 0; t < RTW; ++t) {
          const float a = ys[ctr[u] + po[t]];
#pragma unroll
          for (int c = 0; c < CT; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv[u][c], acc[t][c], 0, 0, 0);
        }
    }
  }

  // D[row = 4*(lane >> 4) + j][col = lane & 15]: row = pair within the row tile, col = ca within the column tile
#pragma unroll
  for (int t = 0; t < RTW; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = (wave * RTW + t) * 16 + 4 * kq + j;
      if (q < Q) {
        const int tap = q / CB, cb = q - tap * CB;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const int ca = c * 16 + r;
          if (ca < g.CA) partial[(((long)blockIdx.x * TAPS + tap) * g.CA + ca) * CB + cb] = acc[t][c][j];
        }
      }
    }
}

}  // namespace

// returns 1 when handled, 0 when not eligible, < 0 on error
int msk_wgrad_cbs(msk_ctx* ctx, const WGrad& g) {
  if (!(g.CB >= 1 && g.CB <= 4 && g.CA >= 8 && g.CA <= 32)) return 0;
  if (!(g.kd == 5 && g.kh == 5 && g.kw == 5 && g.sd == 1 && g.sh == 1 && g.sw == 1 && g.pd == 2 && g.ph == 2 && g.pw == 2))
    return 0;
  if (!(g.AD == g.BD && g.AH == g.BH && g.AW == g.BW)) return 0;
  const long M = (long)g.N * g.BD * g.BH * g.BW;
  const size_t abytes = (size_t)M * g.ald * sizeof(float), bbytes = (size_t)M * g.bld * sizeof(float);
  if (M >= (1L << 30) || abytes >= 0xFFFFFFF0ull || bbytes >= 0xFFFFFFF0ull) return 0;
  const int tiles_d = (g.BD + 3) / 4, tiles_h = (g.BH + 7) / 8, tiles_w = (g.BW + 31) / 32;
  const long ntiles = (long)g.N * tiles_d * tiles_h * tiles_w;
  if (ntiles > 0x7fffffff) return 0;
  const int taps = 125;
  const int rtw = ((taps * g.CB + 15) / 16 + 3) / 4;  // row tiles per wavefront: 2 / 4 / 6 / 8 for CB = 1 .. 4
  const int ct = (g.CA + 15) / 16;
  const size_t lds = (size_t)g.CB * (8 * 12 * 36 + 1) * sizeof(float);  // CB planes of the 8 x 12 x 36 halo (+ the zero slot)
  long splits = 2L * ctx->num_cu;
  if (splits > ntiles) splits = ntiles;
  const size_t per = (size_t)taps * g.CA * g.CB * sizeof(float);
  float* partial = (float*)msk_workspace(ctx, (size_t)splits * per);
  if (!partial) return -1;
  {
    const char* tag = "wgrad_cbs_mfma";
    if (ctx->prof && ctx->prof_shapes) {
      char buf[160];
      snprintf(buf, sizeof(buf), "wgrad_cbs_mfma[ca=%d,cb=%d,M=%ld,splits=%ld]", g.CA, g.CB, M, splits);
      tag = msk_intern_tag(ctx, buf);
    }
    msk_launch_scope ls(ctx, tag);
    const dim3 grid((unsigned)splits), block(256);
#define MSK_CBS(RTW_, CT_)                                                                                              \
  hipLaunchKernelGGL((wgrad_cbs_mfma_k<5, RTW_, CT_>), grid, block, lds, ctx->stream, g, (int)ntiles, tiles_d, tiles_h, \
                     tiles_w, partial, (unsigned)abytes, (unsigned)bbytes)
    if (ct == 1) {
      switch (rtw) {
        case 2: MSK_CBS(2, 1); break;
        case 4: MSK_CBS(4, 1); break;
        case 6: MSK_CBS(6, 1); break;
        default: MSK_CBS(8, 1); break;
      }
    } else {
      switch (rtw) {
        case 2: MSK_CBS(2, 2); break;
        case 4: MSK_CBS(4, 2); break;
        case 6: MSK_CBS(6, 2); break;
        default: MSK_CBS(8, 2); break;
      }
    }
#undef MSK_CBS
    MSK_LAUNCH_CHECK(ctx);
  }
  const int rc = msk_wgrad_reduce(ctx, partial, (int)splits, taps, g.CA, g.CB, g.dw, g.accumulate);
  return rc == 0 ? 1 : rc;
}
