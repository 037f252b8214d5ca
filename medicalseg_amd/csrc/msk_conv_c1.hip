// 'Same' 5^3 convolution with ONE input channel (in_tr.conv1, 1 -> 16, vnet.py:67) on the fp32 matrix pipe:
//     y[v][co] = bias[co] + sum_tap x[v + tap] * W[co][tap]
// The GEMM has K = 125 taps (padded to 128), N = CN <= 16 output channels, M = voxels.  The one-voxel-per-thread VALU
// kernel (conv_halo_valu_k) took 0.39 ms for 2 x 128^3 -- 16.8 GFLOP are 0.11 ms of fp32 MFMA, the 268 MB of output
// 0.06 ms of HBM.  Here (mirror of wgrad_c1_mfma_k):
//   * the 1-channel x halo of a 4 x 8 x 32 voxel tile sits in LDS (13.8 KB);
//   * v_mfma_f32_16x16x4_f32 with D TRANSPOSED: A = weights (rows = output channels, the lane's 32 values -- all 125 taps --
//     stay in registers for the whole kernel), B = x gathered from LDS at (voxel + tap offset); the lane's 32 tap offsets
//     are registers too, so a K step is one ds_read_b32 + one MFMA;
//   * a lane ends up with 4 consecutive output channels of one voxel: a wavefront's store covers 16 voxels x 64 B = 1 KiB
//     of contiguous output.
#include "msk_conv.h"
#include "msk_wbf.h"   // msk_bn_stats_merge

#ifndef C1_PIPE
#define C1_PIPE 1
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOOBc = 0xFFFFFFF0u;

struct C1Args {
  const float* src;
  int sld;
  float* dst;
  int dld;
  int N, D, H, W, CN;
  const float* w;  // canonical [CN][1][125]
  const float* bias;
  int flip;
  int tiles_d, tiles_h, tiles_w, ntiles;
  unsigned src_bytes;
  float* stat_partial;  // non-null: BatchNorm records [gridDim.x][CN][3] = (n, mean, M2) of the stored values (msk_conv3d_fwd_ex)
};

struct C1Rec {
  float n, mean, m2;
};
__device__ __forceinline__ C1Rec c1rec_merge(C1Rec a, C1Rec b) {
  if (b.n == 0.f) return a;
  if (a.n == 0.f) return b;
  C1Rec r;
  r.n = a.n + b.n;
  const float d = b.mean - a.mean, f = b.n / r.n;
  r.mean = a.mean + d * f;
  r.m2 = a.m2 + b.m2 + d * d * a.n * f;
  return r;
}

__global__ void __launch_bounds__(256)
conv_c1_mfma_k(C1Args a) {
  constexpr int KS = 5, TD = 4, TH = 8, TW = 32, P = 2;
  constexpr int HD = TD + 2 * P, HH = TH + 2 * P, HW = TW + 2 * P;
  constexpr int NV = HD * HH * HW;  // 8 x 12 x 36 = 3456 floats
  constexpr int TAPS = 125, KSTEPS = 32;
  __shared__ float xs[NV];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);

  // K step s consumes taps 4s .. 4s+3; this lane supplies tap 4s + lk of both operands
  float wreg[KSTEPS];
  int toff[KSTEPS];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    const int tap = 4 * s + lk;
    const bool live = tap < TAPS;
    const int st = a.flip ? TAPS - 1 - tap : tap;
    wreg[s] = (live && li < a.CN) ? a.w[(long)li * TAPS + st] : 0.f;
    // the three padding taps (zero weight) re-read tap 0's voxel: a finite operand that belongs to this output anyway
    toff[s] = live ? ((tap / (KS * KS)) * HH + (tap / KS) % KS) * HW + tap % KS : 0;
  }
  const int cq = 4 * lk;  // this lane's output-channel quad
  float bq[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bq[j] = (a.bias && cq + j < a.CN) ? a.bias[cq + j] : 0.f;

  // BatchNorm statistics of the stored values (in_tr: conv -> BatchNorm, vnet.py:70,74): shifted sums per lane over ALL the
  // voxels it stores (the workgroup is persistent), merged once at the end -- saves a 268 MB read of y per step
  float sk[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, cnt = 0.f;

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    int t_ = tile;
    const int twi = t_ % a.tiles_w;
    t_ /= a.tiles_w;
    const int thi = t_ % a.tiles_h;
    t_ /= a.tiles_h;
    const int tdi = t_ % a.tiles_d;
    const int n = t_ / a.tiles_d;
    const int d0 = tdi * TD, h0 = thi * TH, w0 = twi * TW;
    __syncthreads();  // the previous tile's readers are done
    {
      // all of a thread's halo loads are issued before the first LDS store (one load -> store per trip left ~14 dependent
      // round trips in front of every tile: tools/isa_scan.py)
      constexpr int NLD = (NV + 255) / 256;
      float hx[NLD];
#pragma unroll
      for (int q = 0; q < NLD; ++q) {
        const int hv = tid + 256 * q;
        const int hd = hv / (HH * HW), rem = hv % (HH * HW), hh = rem / HW, hw = rem % HW;
        const int gd = d0 - P + hd, gh = h0 - P + hh, gw = w0 - P + hw;
        const bool in = hv < NV && (unsigned)gd < (unsigned)a.D && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W;
        hx[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rs, (int)(in ? (unsigned)((((n * a.D + gd) * a.H + gh) * a.W + gw) * a.sld) * 4u : kOOBc), 0, 0));
      }
#pragma unroll
      for (int q = 0; q < NLD; ++q)
        if (tid + 256 * q < NV) xs[tid + 256 * q] = hx[q];
    }
    __syncthreads();
    // a wavefront owns plane d0 + wave: 8 rows x 2 column tiles of 16 voxels
    const int gd = d0 + wave;
    if (gd < a.D) {
#pragma unroll 1
      for (int rt = 0; rt < TH * 2; rt += 2) {  // two independent accumulator chains per trip
        const int h = rt >> 1;
        const int gh = h0 + h;
        if (gh >= a.H) break;
        // two accumulators per column tile (even / odd K steps): four independent MFMA chains and shorter fp32 sums
        f32x4 acc0 = {bq[0], bq[1], bq[2], bq[3]}, acc1 = acc0;
        f32x4 odd0 = {0.f, 0.f, 0.f, 0.f}, odd1 = odd0;
        const int b0 = (wave * HH + h) * HW + li, b1 = b0 + 16;
#if C1_PIPE
        // the operands of the NEXT pair of K steps are requested before this pair's MFMAs (the compiler's own order was
        // ds_read -> wait -> two MFMAs, one LDS round trip exposed per K step with two wavefronts per SIMD)
        float xa0 = xs[b0 + toff[0]], xb0 = xs[b1 + toff[0]], xa1 = xs[b0 + toff[1]], xb1 = xs[b1 + toff[1]];
#pragma unroll
        for (int s = 0; s < KSTEPS; s += 2) {
          float na0 = 0.f, nb0 = 0.f, na1 = 0.f, nb1 = 0.f;
          if (s + 2 < KSTEPS) {
            na0 = xs[b0 + toff[s + 2]]; nb0 = xs[b1 + toff[s + 2]];
            na1 = xs[b0 + toff[s + 3]]; nb1 = xs[b1 + toff[s + 3]];
          }
          __builtin_amdgcn_sched_barrier(0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s], xa0, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s], xb0, acc1, 0, 0, 0);
          odd0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s + 1], xa1, odd0, 0, 0, 0);
          odd1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s + 1], xb1, odd1, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          xa0 = na0; xb0 = nb0; xa1 = na1; xb1 = nb1;
        }
#else
#pragma unroll
        for (int s = 0; s < KSTEPS; s += 2) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s], xs[b0 + toff[s]], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s], xs[b1 + toff[s]], acc1, 0, 0, 0);
          odd0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s + 1], xs[b0 + toff[s + 1]], odd0, 0, 0, 0);
          odd1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s + 1], xs[b1 + toff[s + 1]], odd1, 0, 0, 0);
        }
#endif
        acc0 += odd0;
        acc1 += odd1;
        // D[row = co = 4*lk + j][col = voxel li]
        if (cq < a.CN) {
          const long rowv = (((long)n * a.D + gd) * a.H + gh) * a.W;
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const int gw = w0 + c * 16 + li;
            if (gw < a.W) {
              float* o = a.dst + (rowv + gw) * a.dld + cq;
              const f32x4 v = c == 0 ? acc0 : acc1;
              if (a.stat_partial) {
                if (cnt == 0.f) {
#pragma unroll
                  for (int j = 0; j < 4; ++j) sk[j] = v[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  const float dl = v[j] - sk[j];
                  s1[j] += dl;
                  s2[j] = fmaf(dl, dl, s2[j]);
                }
                cnt += 1.f;
              }
              if (cq + 3 < a.CN) {
                *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
              } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                  if (cq + j < a.CN) o[j] = v[j];
              }
            }
          }
        }
      }
    }
  }
  if (a.stat_partial) {
    __shared__ float shr[4][16][3];
    C1Rec r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r[j] = C1Rec{0.f, 0.f, 0.f};
      if (cnt > 0.f) {
        r[j].n = cnt;
        r[j].mean = sk[j] + s1[j] / cnt;
        r[j].m2 = fmaxf(s2[j] - s1[j] * s1[j] / cnt, 0.f);
      }
      // the 16 lanes li of a quad group hold the same four channels: fixed butterfly order
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        C1Rec q;
        q.n = __shfl_xor(r[j].n, o, 64);
        q.mean = __shfl_xor(r[j].mean, o, 64);
        q.m2 = __shfl_xor(r[j].m2, o, 64);
        r[j] = (li & o) ? c1rec_merge(q, r[j]) : c1rec_merge(r[j], q);   // both partners merge (lower, upper)
      }
    }
    __syncthreads();
    if (li == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        shr[wave][cq + j][0] = r[j].n; shr[wave][cq + j][1] = r[j].mean; shr[wave][cq + j][2] = r[j].m2;
      }
    }
    __syncthreads();
    if (tid < a.CN) {
      C1Rec t = {shr[0][tid][0], shr[0][tid][1], shr[0][tid][2]};
#pragma unroll
      for (int w = 1; w < 4; ++w) t = c1rec_merge(t, C1Rec{shr[w][tid][0], shr[w][tid][1], shr[w][tid][2]});
      float* o = a.stat_partial + ((long)blockIdx.x * a.CN + tid) * 3;
      o[0] = t.n; o[1] = t.mean; o[2] = t.m2;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Round 4: the same convolution class on the 16-bit matrix pipe -- v_mfma_f32_16x16x32_f16, K = 32 TAPS per instruction
// (125 taps: 4 steps; 27 taps: 1) instead of 4, operands as fp16 pieces of the power-of-two-scaled values (NP = 2: hi + lo,
// three products per fp32 product, the product default "conv_split" 2; NP = 1: one fp16 value, UNet3D precision="fp16").
// The one-channel halo tile sits in LDS as fp16 pieces (converted once per tile); a lane gathers the 8 taps of its K group
// for its voxel (ds_read_u16 x 8 per piece), the weights' fragments of all steps stay in registers for the whole kernel.
// fp32 form: 32 MFMAs of 32 cycles per 16 voxels x 16 channels; here 12 of ~17 cycles (5^3), 3 (3^3, fp16 x 2).
// KS = 3 with up to 32 output channels (NT = 2) is the first convolution of the builder-defined UNet3D.
struct C1HArgs {
  C1Args c;
  const float* x_amax;   // amax array of the source tensor (power-of-two scale), or null: unscaled
};
typedef _Float16 c1_f16x8 __attribute__((ext_vector_type(8)));

template <int KS, int NT, int NP>
__global__ void __launch_bounds__(256)
conv_c1_h2_k(C1HArgs ha) {
  const C1Args& a = ha.c;
  constexpr int TD = 4, TH = 8, TW = 32, P = KS / 2;
  constexpr int HD = TD + 2 * P, HH = TH + 2 * P, HW = TW + 2 * P;
  constexpr int NV = HD * HH * HW;
  constexpr int TAPS = KS * KS * KS, KSTEPS = (TAPS + 31) / 32;
  __shared__ _Float16 xh[NV + 8];
  __shared__ _Float16 xl[NP == 2 ? NV + 8 : 8];
  __shared__ float red[4];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);

  // power-of-two scales: the source tensor's from its amax array, the weights' from their maximum (<= 32 x 125 values: every
  // workgroup takes it itself)
  float wm = 0.f;
  for (int i = tid; i < a.CN * TAPS; i += 256) wm = fmaxf(wm, fabsf(a.w[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wm = fmaxf(wm, __shfl_xor(wm, o, 64));
  if (lane == 0) red[wave] = wm;
  __syncthreads();
  wm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sx = wbf_scale_of(ha.x_amax), sw = wbf_scale_from(wm);
  const float osc = 1.f / (sx * sw);

  // A fragments (weights): lane (row li = output channel of the tile, K group lk) holds taps 32 s + 8 lk .. + 7
  uint4 wh[KSTEPS][NT], wl[KSTEPS][NT];
  int toff[KSTEPS][8];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int tap = 32 * s + 8 * lk + e;
      // the padding taps (zero weight) re-read tap 0's voxel: a finite operand
      toff[s][e] = tap < TAPS ? ((tap / (KS * KS)) * HH + (tap / KS) % KS) * HW + tap % KS : 0;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      unsigned hv[4], lv[4];
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        float v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int tap = 32 * s + 8 * lk + e + u, co = nt * 16 + li;
          const int st = a.flip ? TAPS - 1 - tap : tap;
          v[u] = (tap < TAPS && co < a.CN) ? a.w[(long)co * TAPS + st] * sw : 0.f;
        }
        wbf_split2h_pair(v[0], v[1], hv[e >> 1], lv[e >> 1]);
      }
      wh[s][nt] = make_uint4(hv[0], hv[1], hv[2], hv[3]);
      wl[s][nt] = make_uint4(lv[0], lv[1], lv[2], lv[3]);
    }
  }
  const int cq = 4 * lk;  // this lane's output-channel quad inside a 16-channel tile
  float bq[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) bq[nt][j] = (a.bias && nt * 16 + cq + j < a.CN) ? a.bias[nt * 16 + cq + j] : 0.f;

  // BatchNorm statistics of the stored values: shifted sums per lane over all the voxels it stores (persistent workgroup)
  float sk[NT][4], s1[NT][4], s2[NT][4], cnt = 0.f;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) sk[nt][j] = s1[nt][j] = s2[nt][j] = 0.f;

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    int t_ = tile;
    const int twi = t_ % a.tiles_w;
    t_ /= a.tiles_w;
    const int thi = t_ % a.tiles_h;
    t_ /= a.tiles_h;
    const int tdi = t_ % a.tiles_d;
    const int n = t_ / a.tiles_d;
    const int d0 = tdi * TD, h0 = thi * TH, w0 = twi * TW;
    __syncthreads();  // the previous tile's readers are done
    {
      constexpr int NLD = (NV + 255) / 256;
      float hx[NLD];
#pragma unroll
      for (int q = 0; q < NLD; ++q) {
        const int hv = tid + 256 * q;
        const int hd = hv / (HH * HW), rem = hv % (HH * HW), hh = rem / HW, hw = rem % HW;
        const int gd = d0 - P + hd, gh = h0 - P + hh, gw = w0 - P + hw;
        const bool in = hv < NV && (unsigned)gd < (unsigned)a.D && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W;
        hx[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rs, (int)(in ? (unsigned)((((n * a.D + gd) * a.H + gh) * a.W + gw) * a.sld) * 4u : kOOBc), 0, 0));
      }
#pragma unroll
      for (int q = 0; q < NLD; ++q)
        if (tid + 256 * q < NV) {
          const float v = hx[q] * sx;
          const _Float16 h = (_Float16)v;
          xh[tid + 256 * q] = h;
          if (NP == 2) xl[tid + 256 * q] = (_Float16)(v - (float)h);
        }
    }
    __syncthreads();
    const int gd = d0 + wave;
    if (gd < a.D) {
#pragma unroll 1
      for (int h = 0; h < TH; ++h) {
        const int gh = h0 + h;
        if (gh >= a.H) break;
        f32x4 acc[2][NT];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[c][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int b0 = (wave * HH + h) * HW + li;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            c1_f16x8 bh, bl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              bh[e] = xh[b0 + c * 16 + toff[s][e]];
              if (NP == 2) bl[e] = xl[b0 + c * 16 + toff[s][e]];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              if (NP == 2) {
                acc[c][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(c1_f16x8, wl[s][nt]), bh, acc[c][nt], 0, 0, 0);
                acc[c][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(c1_f16x8, wh[s][nt]), bl, acc[c][nt], 0, 0, 0);
              }
              acc[c][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(c1_f16x8, wh[s][nt]), bh, acc[c][nt], 0, 0, 0);
            }
          }
        }
        // D[row = co = 16 nt + 4 lk + j][col = voxel li]
        const long rowv = (((long)n * a.D + gd) * a.H + gh) * a.W;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int gw = w0 + c * 16 + li;
          if (gw < a.W) {
            float* o = a.dst + (rowv + gw) * a.dld;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const int co = nt * 16 + cq;
              if (co >= a.CN) continue;
              float v[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = fmaf(acc[c][nt][j], osc, bq[nt][j]);
              if (a.stat_partial) {
                if (cnt == 0.f) {
#pragma unroll
                  for (int j = 0; j < 4; ++j) sk[nt][j] = v[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  const float dl = v[j] - sk[nt][j];
                  s1[nt][j] += dl;
                  s2[nt][j] = fmaf(dl, dl, s2[nt][j]);
                }
              }
              if (co + 3 < a.CN) {
                *reinterpret_cast<float4*>(o + co) = make_float4(v[0], v[1], v[2], v[3]);
              } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                  if (co + j < a.CN) o[co + j] = v[j];
              }
            }
            if (a.stat_partial) cnt += 1.f;
          }
        }
      }
    }
  }
  if (a.stat_partial) {
    __shared__ float shr[4][16 * NT][3];
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      C1Rec r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r[j] = C1Rec{0.f, 0.f, 0.f};
        if (cnt > 0.f) {
          r[j].n = cnt;
          r[j].mean = sk[nt][j] + s1[nt][j] / cnt;
          r[j].m2 = fmaxf(s2[nt][j] - s1[nt][j] * s1[nt][j] / cnt, 0.f);
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          C1Rec q;
          q.n = __shfl_xor(r[j].n, o, 64);
          q.mean = __shfl_xor(r[j].mean, o, 64);
          q.m2 = __shfl_xor(r[j].m2, o, 64);
          r[j] = (li & o) ? c1rec_merge(q, r[j]) : c1rec_merge(r[j], q);
        }
      }
      if (li == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          shr[wave][nt * 16 + cq + j][0] = r[j].n; shr[wave][nt * 16 + cq + j][1] = r[j].mean; shr[wave][nt * 16 + cq + j][2] = r[j].m2;
        }
      }
    }
    __syncthreads();
    if (tid < a.CN) {
      C1Rec t = {shr[0][tid][0], shr[0][tid][1], shr[0][tid][2]};
#pragma unroll
      for (int w = 1; w < 4; ++w) t = c1rec_merge(t, C1Rec{shr[w][tid][0], shr[w][tid][1], shr[w][tid][2]});
      float* o = a.stat_partial + ((long)blockIdx.x * a.CN + tid) * 3;
      o[0] = t.n; o[1] = t.mean; o[2] = t.m2;
    }
  }
}

}  // namespace

// One input channel, 5^3 (<= 16 output channels: in_tr.conv1) or 3^3 (<= 32: UNet3D's first convolution) 'same' convolution on
// the 16-bit matrix pipe (conv_c1_h2_k).  1 handled, 0 not eligible, < 0 error.  conv_impl 23 / 27 = A/B: the fp32 kernels.
int msk_gconv_c1_h2(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap) {
  (void)A; (void)B; (void)swap;
  const bool k5 = g.kd == 5 && g.kh == 5 && g.kw == 5 && g.pd == 2 && g.ph == 2 && g.pw == 2;
  const bool k3 = g.kd == 3 && g.kh == 3 && g.kw == 3 && g.pd == 1 && g.ph == 1 && g.pw == 1;
  if (!(k5 || k3) || !(g.sd == 1 && g.sh == 1 && g.sw == 1)) return 0;
  if (!(g.SD == g.DD && g.SH == g.DH && g.SW == g.DW)) return 0;
  if (g.CK != 1 || g.CN < 1 || g.CN > (k5 ? 16 : 32) || g.accumulate || g.prelu) return 0;
  // 5^3 (in_tr.conv1): measured 0.162 against 0.191 ms for the fp32-MFMA kernel -- the 16-bit form is bound by its LDS gathers -- and a
  // different rounding realisation of the FIRST layer moves the ill-conditioned 16^3 trajectory tests (tools/ab_c1_16cube.py):
  // not worth 0.03 ms, so the product keeps the fp32 kernel there (option c1_h2 2 = use it for 5^3 as well)
  if (g.DW < 16 || ctx->conv_impl == 27 || !ctx->c1_h2 || (k5 && ctx->c1_h2 < 2)) return 0;
  const int NP = (k3 && ctx->conv_fp16) ? 1 : (ctx->conv_split == 2 ? 2 : 0);
  if (NP == 0) return 0;   // exact operands requested: the fp32 kernels
  if (g.dld % 4 || (((uintptr_t)g.dst) & 15)) return 0;
  const size_t sb = (size_t)g.N * g.SD * g.SH * g.SW * g.sld * sizeof(float);
  if (sb >= 0xFFFFFFF0ull) return 0;
  C1HArgs ha{};
  C1Args& a = ha.c;
  a.src = g.src; a.sld = g.sld; a.dst = g.dst; a.dld = g.dld;
  a.N = g.N; a.D = g.DD; a.H = g.DH; a.W = g.DW; a.CN = g.CN;
  a.w = w_canon; a.bias = g.bias; a.flip = g.transposed ? 1 : 0;
  a.tiles_d = msk_cdiv(a.D, 4); a.tiles_h = msk_cdiv(a.H, 8); a.tiles_w = msk_cdiv(a.W, 32);
  const long ntiles = (long)a.N * a.tiles_d * a.tiles_h * a.tiles_w;
  if (ntiles > 0x7fffffff) return 0;
  a.ntiles = (int)ntiles;
  a.src_bytes = (unsigned)sb;
  ha.x_amax = g.in_amax ? g.in_amax : msk_absmax(ctx, g.src, g.sld, 1, (long)g.N * g.SD * g.SH * g.SW);
  if (!ha.x_amax) return -1;
  long blocks = 4L * ctx->num_cu;
  if (blocks > ntiles) blocks = ntiles;
  const int NT = g.CN > 16 ? 2 : 1;
  const char* tag = "conv_c1_h2";
  if (ctx->prof && ctx->prof_shapes) {
    char buf[160];
    snprintf(buf, sizeof(buf), "conv_c1_h2[cn=%d,n=%d,dhw=%dx%dx%d,k=%d,np=%d]", g.CN, g.N, g.DD, g.DH, g.DW, g.kd, NP);
    tag = msk_intern_tag(ctx, buf);
  }
  const bool want_stats = g.stats != nullptr && !g.stats_ps && g.CN == 16 * NT;
  if (want_stats) {
    a.stat_partial = (float*)msk_workspace(ctx, (size_t)blocks * g.CN * 3 * sizeof(float));
    if (!a.stat_partial) return -1;
  }
  {
    msk_launch_scope ls(ctx, tag);
    const dim3 grid((unsigned)blocks);
    if (k5) hipLaunchKernelGGL((conv_c1_h2_k<5, 1, 2>), grid, dim3(256), 0, ctx->stream, ha);
    else if (NP == 2 && NT == 1) hipLaunchKernelGGL((conv_c1_h2_k<3, 1, 2>), grid, dim3(256), 0, ctx->stream, ha);
    else if (NP == 2) hipLaunchKernelGGL((conv_c1_h2_k<3, 2, 2>), grid, dim3(256), 0, ctx->stream, ha);
    else if (NT == 1) hipLaunchKernelGGL((conv_c1_h2_k<3, 1, 1>), grid, dim3(256), 0, ctx->stream, ha);
    else hipLaunchKernelGGL((conv_c1_h2_k<3, 2, 1>), grid, dim3(256), 0, ctx->stream, ha);
    MSK_LAUNCH_CHECK(ctx);
  }
  if (want_stats) {
    if (msk_bn_stats_merge(ctx, a.stat_partial, (int)blocks, g.CN, g.stats, g.fin) != 0) return -1;
    ctx->stats_fused = true;
  }
  return 1;
}

namespace {
}  // namespace

// returns 1 when handled, 0 when not eligible, < 0 on error
int msk_gconv_c1_mfma(msk_ctx* ctx, const GConv& g, const float* w_canon, int A, int B, int swap) {
  if (!(g.kd == 5 && g.kh == 5 && g.kw == 5 && g.sd == 1 && g.sh == 1 && g.sw == 1 && g.pd == 2 && g.ph == 2 && g.pw == 2)) return 0;
  if (!(g.SD == g.DD && g.SH == g.DH && g.SW == g.DW)) return 0;
  if (!(g.CK == 1 && g.CN >= 1 && g.CN <= 16) || g.accumulate) return 0;
  if (g.DW < 16) return 0;                                   // narrow slabs keep the VALU kernel's tiles
  // canonical w[a][b][tap] with (k, n) = swap ? (b, a) : (a, b); CK = 1 -> both read w[n*125 + tap]
  (void)A; (void)B; (void)swap;
  if (g.dld % 4 || (((uintptr_t)g.dst) & 15)) return 0;
  const size_t sb = (size_t)g.N * g.SD * g.SH * g.SW * g.sld * sizeof(float);
  if (sb >= 0xFFFFFFF0ull) return 0;
  C1Args a{};
  a.src = g.src; a.sld = g.sld; a.dst = g.dst; a.dld = g.dld;
  a.N = g.N; a.D = g.DD; a.H = g.DH; a.W = g.DW; a.CN = g.CN;
  a.w = w_canon; a.bias = g.bias; a.flip = g.transposed ? 1 : 0;
  a.tiles_d = msk_cdiv(a.D, 4); a.tiles_h = msk_cdiv(a.H, 8); a.tiles_w = msk_cdiv(a.W, 32);
  const long ntiles = (long)a.N * a.tiles_d * a.tiles_h * a.tiles_w;
  if (ntiles > 0x7fffffff) return 0;
  a.ntiles = (int)ntiles;
  a.src_bytes = (unsigned)sb;
  long blocks = 4L * ctx->num_cu;  // persistent: the 64 weight / offset registers are loaded once per workgroup
  if (blocks > ntiles) blocks = ntiles;
  const char* tag = "conv_c1_mfma";
  if (ctx->prof && ctx->prof_shapes) {
    char buf[160];
    snprintf(buf, sizeof(buf), "conv_c1_mfma[cn=%d,n=%d,dhw=%dx%dx%d]", g.CN, g.N, g.DD, g.DH, g.DW);
    tag = msk_intern_tag(ctx, buf);
  }
  const bool want_stats = g.stats != nullptr && !g.stats_ps && !g.prelu && g.CN == 16;   // (all four channel quads live: the LDS record table is full)
  if (want_stats) {
    a.stat_partial = (float*)msk_workspace(ctx, (size_t)blocks * g.CN * 3 * sizeof(float));
    if (!a.stat_partial) return -1;
  }
  {
    msk_launch_scope ls(ctx, tag);
    hipLaunchKernelGGL(conv_c1_mfma_k, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
    MSK_LAUNCH_CHECK(ctx);
  }
  if (want_stats) {
    if (msk_bn_stats_merge(ctx, a.stat_partial, (int)blocks, g.CN, g.stats, g.fin) != 0) return -1;
    ctx->stats_fused = true;
  }
  return 1;
}
