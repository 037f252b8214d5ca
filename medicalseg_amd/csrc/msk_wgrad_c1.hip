// Weight gradient of a 'same' 5^3 convolution with ONE input channel (in_tr.conv1, 1 -> 16, vnet.py:67):
//     dW[cb][0][tap] = sum_v x[v + tap] * dy[v][cb]
// i.e. 125 x 16 dot products over all voxels.  It is the LAST weight gradient of a backward pass, so nothing hides it:
// the optimizer waits for it.  The generic folded kernel (wgrad_fold_mfma_k<true>) gathers x straight from global
// memory, one dword per lane and tap (0.63 ms + 0.03 ms reduce for 2 x 128^3; the tensors are 17 MB and 268 MB).
// Here the 1-channel x halo of a 4 x 8 x 32 voxel tile sits in LDS (13.8 KB), the MFMA rows are TAPS
// (v_mfma_f32_16x16x4_f32: 16 taps x 16 output channels x 4 voxels; 8 row tiles cover 125 taps), the A operand is
// one ds_read_b32 at (voxel + tap offset from an LDS table), the B operand one coalesced dword of dy (16 channels of 4
// consecutive voxels = 256 B per load).  A workgroup walks tiles round-robin and writes ONE partial slab.
#include "msk_conv.h"
#include "msk_wbf.h"   // WbfBnBwd

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOOB1 = 0xFFFFFFF0u;

__device__ __forceinline__ float c1_load(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0));
}

// FUSE: dy is not read -- it is evaluated from (y, dout) and the per-channel BatchNorm-backward coefficients, the arithmetic of
// affine_act_bwd_apply (msk_conv3d_bwd_bnact for the one-input-channel class): the in_tr unit's backward loses a pass that read
// two and wrote one full-resolution tensor, and the step's last weight gradient starts one kernel earlier.
template <int KS, bool FUSE>
__global__ void __launch_bounds__(256)
wgrad_c1_mfma_k(WGrad g, int ntiles, int tiles_d, int tiles_h, int tiles_w, float* __restrict__ partial, unsigned a_bytes,
                unsigned b_bytes, WbfBnBwd yf, unsigned d_bytes) {
  constexpr int TD = 4, TH = 8, TW = 32, P = KS / 2;
  constexpr int HD = TD + 2 * P, HH = TH + 2 * P, HW = TW + 2 * P;
  constexpr int NV = HD * HH * HW;
  constexpr int TAPS = KS * KS * KS, RT = (TAPS + 15) / 16;  // 8 row tiles of 16 taps
  __shared__ float xs[NV + 1];                                // xs[NV] = 0: the address of the padding taps
  __shared__ int toff[RT * 16];
  __shared__ float red[4][RT * 16 * 16];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 15, kq = lane >> 4;
  const int D = g.BD, H = g.BH, W = g.BW;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(FUSE ? yf.y : g.B), 0, b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(FUSE ? yf.dout : g.B), 0, FUSE ? d_bytes : b_bytes, 0x00020000);
  // this lane's output channel r: coefficients of dy = scale * (du - sums[c]/M - xhat * sums[C + c]/M)
  float c_sc = 0.f, c_sf = 0.f, c_al = 1.f, c_mu = 0.f, c_is = 0.f, c_s1 = 0.f, c_s2 = 0.f;
  if (FUSE && r < g.CB) {
    c_sc = yf.scale[r]; c_sf = yf.shift[r]; c_al = yf.alpha ? yf.alpha[r] : 1.f; c_mu = yf.mean[r]; c_is = yf.invstd[r];
    c_s1 = yf.sums[r] * yf.invM; c_s2 = yf.sums[g.CB + r] * yf.invM;
  }
  const bool use_res = FUSE && yf.res_is_input != 0;

  for (int i = tid; i < RT * 16; i += 256)
    toff[i] = i < TAPS ? ((i / (KS * KS)) * HH + (i / KS) % KS) * HW + i % KS : NV;
  if (tid == 0) xs[NV] = 0.f;

  __syncthreads();
  int to[RT];  // this lane's tap offsets (halo index space), one per row tile
#pragma unroll
  for (int t = 0; t < RT; ++t) to[t] = toff[t * 16 + r];

  f32x4 acc[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int t_ = tile;
    const int twi = t_ % tiles_w;
    t_ /= tiles_w;
    const int thi = t_ % tiles_h;
    t_ /= tiles_h;
    const int tdi = t_ % tiles_d;
    const int n = t_ / tiles_d;
    const int d0 = tdi * TD, h0 = thi * TH, w0 = twi * TW;
    __syncthreads();  // the previous tile's readers are done (and toff / xs[NV] are visible on the first trip)
    {
      // all of a thread's halo loads are issued before the first LDS store (tools/isa_scan.py: one load -> store per trip was
      // ~14 dependent round trips in front of every tile)
      constexpr int NLD = (NV + 255) / 256;
      float hx[NLD];
#pragma unroll
      for (int q = 0; q < NLD; ++q) {
        const int hv = tid + 256 * q;
        const int hd = hv / (HH * HW), rem = hv % (HH * HW), hh = rem / HW, hw = rem % HW;
        const int gd = d0 - P + hd, gh = h0 - P + hh, gw = w0 - P + hw;
        const bool in = hv < NV && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
        hx[q] = c1_load(ra, in ? (unsigned)((((n * D + gd) * H + gh) * W + gw) * g.ald) * 4u : kOOB1);
      }
#pragma unroll
      for (int q = 0; q < NLD; ++q)
        if (tid + 256 * q < NV) xs[tid + 256 * q] = hx[q];
    }
    __syncthreads();
    const int gd = d0 + wave;  // this wavefront's plane
    if (gd < D) {
      // batches of UB K-steps: all dy loads of a batch are in flight before its MFMAs (one load -> 8 MFMAs per step
      // left the loop waiting on global latency)
      constexpr int UB = 8;
#pragma unroll 1
      for (int s0 = 0; s0 < TH * (TW / 4); s0 += UB) {
        float bv[UB];
        int bs[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int s = s0 + u;
          const int h = s / (TW / 4), w = (s % (TW / 4)) * 4 + kq;
          const int gh = h0 + h, gw = w0 + w;
          const bool vok = gh < H && gw < W;
          // x of a voxel outside the volume only ever meets dy = 0 (vok false -> b = 0): its halo index needs no guard
          bs[u] = (wave * HH + h) * HW + w;
          if (!FUSE) {
            bv[u] = c1_load(rb, (vok && r < g.CB) ? (unsigned)(((((n * D + gd) * H + gh) * W + gw) * g.bld) + r) * 4u : kOOB1);
          } else {
            const bool live = vok && r < g.CB;
            const unsigned vx = (unsigned)(((n * D + gd) * H + gh) * W + gw);
            const float yv = c1_load(rb, live ? (vx * (unsigned)yf.yld + r) * 4u : kOOB1);
            float d = c1_load(rd, live ? (vx * (unsigned)yf.dld + r) * 4u : kOOB1);
            // branch-free (c_al = 1 without a PReLU, a select for the residual): a runtime branch here kept the batch's
            // loads from being issued together (0.23 -> 0.41 ms)
            const float xc = xs[bs[u] + (P * HH + P) * HW + P];
            const float uu = fmaf(yv, c_sc, c_sf) + (use_res ? xc : 0.f);   // + x at this voxel (the halo's centre tap)
            d *= (uu > 0.f) ? 1.f : c_al;
            const float xh = (yv - c_mu) * c_is;
            // dead lanes loaded zeros (out-of-range offsets): a multiply instead of a select keeps the compiler from sinking the
            // loads into a branch (which made every K-step wait for its own loads: 0.23 -> 0.41 ms)
            bv[u] = (live ? 1.f : 0.f) * (c_sc * (d - c_s1 - xh * c_s2));
          }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
          for (int t = 0; t < RT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xs[bs[u] + to[t]], bv[u], acc[t], 0, 0, 0);
      }
    }
  }

  // D[row = 4*(lane >> 4) + j][col = lane & 15]: row = tap within the row tile, col = cb.  Sum the four planes' waves.
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[wave][(t * 16 + 4 * kq + j) * 16 + r] = acc[t][j];
  __syncthreads();
  for (int i = tid; i < RT * 16 * 16; i += 256) {
    const int tap = i >> 4, cb = i & 15;
    if (tap < TAPS && cb < g.CB)
      partial[((long)blockIdx.x * TAPS + tap) * g.CB + cb] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
  }
}

}  // namespace

// returns 1 when handled, 0 when not eligible, < 0 on error
// round 4: also 3^3 (the first convolution of the builder-defined UNet3D, 1 -> 32), and more than 16 output channels as blocks of
// 16 (one launch per block: dy is still read once in total, the one-channel x once per block)
int msk_wgrad_c1(msk_ctx* ctx, const WGrad& g) {
  const bool k5 = g.kd == 5 && g.kh == 5 && g.kw == 5 && g.pd == 2 && g.ph == 2 && g.pw == 2;
  const bool k3 = g.kd == 3 && g.kh == 3 && g.kw == 3 && g.pd == 1 && g.ph == 1 && g.pw == 1;
  const WbfBnBwd* yf = g.yfuse;   // msk_conv3d_bwd_bnact: dy evaluated in the kernel (B is not read)
  if (!(g.CA == 1 && g.CB >= 1 && g.CB <= (yf ? 16 : 64))) return 0;
  if (!(k5 || (k3 && !yf)) || !(g.sd == 1 && g.sh == 1 && g.sw == 1)) return 0;
  if (!(g.AD == g.BD && g.AH == g.BH && g.AW == g.BW)) return 0;
  const long M = (long)g.N * g.BD * g.BH * g.BW;
  const size_t abytes = (size_t)M * g.ald * sizeof(float), bbytes = (size_t)M * (yf ? yf->yld : g.bld) * sizeof(float);
  const size_t dbytes = yf ? (size_t)M * yf->dld * sizeof(float) : 0;
  if (M >= (1L << 30) || abytes >= 0xFFFFFFF0ull || bbytes >= 0xFFFFFFF0ull || dbytes >= 0xFFFFFFF0ull) return 0;
  const int tiles_d = (g.BD + 3) / 4, tiles_h = (g.BH + 7) / 8, tiles_w = (g.BW + 31) / 32;
  const long ntiles = (long)g.N * tiles_d * tiles_h * tiles_w;
  if (ntiles > 0x7fffffff) return 0;
  long splits = (long)ctx->wgrad_c1_wpc * ctx->num_cu;  // persistent workgroups (LDS: 13.8 KB halo + 32 KB reduction buffer -> 3 per CU; option "wgrad_c1_wpc", default 2)
  if (splits > ntiles) splits = ntiles;
  const int taps = k5 ? 125 : 27;
  for (int cb0 = 0; cb0 < g.CB; cb0 += 16) {
    WGrad gb = g;
    gb.CB = g.CB - cb0 < 16 ? g.CB - cb0 : 16;
    if (gb.B) gb.B = g.B + cb0;
    gb.dw = g.dw + (size_t)cb0 * taps;
    const size_t per = (size_t)taps * gb.CB * sizeof(float);
    float* partial = (float*)msk_workspace(ctx, (size_t)splits * per);
    if (!partial) return -1;
    {
      const char* tag = "wgrad_c1_mfma";
      if (ctx->prof && ctx->prof_shapes) {
        char buf[160];
        snprintf(buf, sizeof(buf), "wgrad_c1_mfma[cb=%d,k=%d,M=%ld,splits=%ld%s]", gb.CB, g.kd, M, splits, yf ? ",bn-fused" : "");
        tag = msk_intern_tag(ctx, buf);
      }
      msk_launch_scope ls(ctx, tag);
      const unsigned bb = (unsigned)(bbytes - (size_t)cb0 * sizeof(float));
      if (yf)
        hipLaunchKernelGGL((wgrad_c1_mfma_k<5, true>), dim3((unsigned)splits), dim3(256), 0, ctx->stream, gb, (int)ntiles, tiles_d, tiles_h,
                           tiles_w, partial, (unsigned)abytes, bb, *yf, (unsigned)dbytes);
      else if (k5)
        hipLaunchKernelGGL((wgrad_c1_mfma_k<5, false>), dim3((unsigned)splits), dim3(256), 0, ctx->stream, gb, (int)ntiles, tiles_d, tiles_h,
                           tiles_w, partial, (unsigned)abytes, bb, WbfBnBwd{}, 0u);
      else
        hipLaunchKernelGGL((wgrad_c1_mfma_k<3, false>), dim3((unsigned)splits), dim3(256), 0, ctx->stream, gb, (int)ntiles, tiles_d, tiles_h,
                           tiles_w, partial, (unsigned)abytes, bb, WbfBnBwd{}, 0u);
      MSK_LAUNCH_CHECK(ctx);
    }
    const int rc = msk_wgrad_reduce(ctx, partial, (int)splits, taps, 1, gb.CB, gb.dw, g.accumulate);
    if (rc != 0) return rc;
  }
  return 1;
}
