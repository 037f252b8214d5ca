// Fused CrossEntropy + Dice loss (forward statistics, backward) and the fused
// SGD-momentum-L2 update, for gfx950.
//
// Thread mapping for the loss: CB = pow2ceil(C) adjacent lanes own one voxel
// (lane = class), so the softmax max/sum are wavefront shuffles over CB lanes and the
// per-class Dice sums are per-thread scalars.  Logits are read once per pass.
#include "msk_common.h"

namespace {

constexpr int kThreads = 256;

inline int pow2ceil(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

template <typename F>
__device__ __forceinline__ float group_reduce(float v, int CB, F op) {
  for (int o = CB >> 1; o > 0; o >>= 1) v = op(v, __shfl_xor(v, o, 64));
  return v;
}

// mode 0: class-weight statistics (sum of softmax mass per class)
// mode 1: loss statistics {I, S, T, ce_num, ce_den} per class
template <int MODE>
__global__ void __launch_bounds__(kThreads)
loss_stats_k(const float* __restrict__ z, int ld, const int32_t* __restrict__ labels,
             const float* __restrict__ weights, int ignore_index, long voxels, int C, int CB,
             float* __restrict__ partial /*[nb][NQ][CB]*/, int dice_softmax) {
  constexpr int NQ = MODE == 0 ? 1 : 5;
  __shared__ float sh[NQ][kThreads];
  const int t = threadIdx.x;
  const int c = t % CB, vl = t / CB, VPB = kThreads / CB;
  const int nb = gridDim.x;
  const long per = (voxels + nb - 1) / nb;
  const long v0 = (long)blockIdx.x * per;
  long v1 = v0 + per;
  if (v1 > voxels) v1 = voxels;
  float acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = 0.f;
  const float wc = (MODE == 1 && c < C) ? weights[c] : 0.f;
  // all CB lanes of a voxel group iterate together (trip count depends on vl only)
  for (long v = v0 + vl; v < v1; v += VPB) {
    const bool live = c < C;
    const float zc = live ? z[v * ld + c] : 0.f;
    const float zs = live ? zc + 1e-8f : -INFINITY;  // cross_entropy_loss.py:79 logit + EPS
    const float m = group_reduce(zs, CB, [](float a, float b) { return fmaxf(a, b); });
    const float e = live ? expf(zs - m) : 0.f;
    const float se = group_reduce(e, CB, [](float a, float b) { return a + b; });
    if constexpr (MODE == 0) {
      acc[0] += e / se;
    } else {
      const int y = labels[v];
      float s;
      if (dice_softmax) {  // dice_loss.py:42-43 nn.Softmax(axis=1) of the raw logits (no EPS: that is the CE term's)
        const float m2 = group_reduce(live ? zc : -INFINITY, CB, [](float a, float b) { return fmaxf(a, b); });
        const float e2 = live ? expf(zc - m2) : 0.f;
        s = e2 / group_reduce(e2, CB, [](float a, float b) { return a + b; });
      } else {
        s = 1.f / (1.f + expf(-zc));
      }
      const bool hit = live && (y == c);
      if (live) {
        acc[1] = fmaf(s, s, acc[1]);
        if (hit) {
          acc[0] += s;
          acc[2] += 1.f;
          if (y != ignore_index) {
            const float logp = zs - m - logf(se);
            acc[3] = fmaf(wc, -logp, acc[3]);
            acc[4] += wc;
          }
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) sh[q][t] = acc[q];
  __syncthreads();
  for (int s = VPB >> 1; s > 0; s >>= 1) {
    if (vl < s) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) sh[q][t] += sh[q][t + s * CB];
    }
    __syncthreads();
  }
  if (vl == 0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) partial[((long)blockIdx.x * NQ + q) * CB + c] = sh[q][t];
  }
}

__global__ void class_weights_final_k(const float* __restrict__ partial, int nb, int C, int CB, double voxels,
                                      float* __restrict__ weights) {
  int c = threadIdx.x;
  if (c >= C) return;
  double s = 0;
  for (int b = 0; b < nb; ++b) s += partial[(long)b * CB + c];
  weights[c] = (float)((voxels - s) / s);  // sum(1-p)/sum(p)
}

__global__ void loss_final_k(const float* __restrict__ partial, int nb, int C, int CB, float* __restrict__ out,
                             double* __restrict__ stats /*[3C+2]*/, const float* __restrict__ dice_weight) {
  // 16 wavefronts: the (class, quantity) pairs are dealt out to the wavefronts, whose 64 lanes stride over the block
  // partials and are combined with a fixed-order shuffle tree (one wavefront for everything took 0.075 ms at C = 3 with
  // 2048 partials; thread c walking all partials of class c alone: 0.4 ms)
  __shared__ double q[64][5];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  for (int p = wave; p < C * 5; p += nwaves) {
    const int c = p / 5, k = p - c * 5;
    double s = 0.0;
    for (int b = lane; b < nb; b += 64) s += partial[((long)b * 5 + k) * CB + c];
    s = msk_wave_sum_d(s);
    if (lane == 0) q[c][k] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sp = 0, sn = 0, sd = 0;
    for (int c = 0; c < C; ++c) {
      stats[c] = q[c][0];
      stats[C + c] = q[c][1];
      stats[2 * C + c] = q[c][2];
      const double den = q[c][1] + q[c][2];
      const double wI = dice_weight ? (double)dice_weight[c] * q[c][0] : q[c][0];  // dice_loss.py:68-69
      const double per = 2.0 * wI / (den > 1e-6 ? den : 1e-6);  // dice_loss.py:74 clip(min=1e-6)
      out[2 + c] = (float)per;
      sp += per;
      sn += q[c][3];
      sd += q[c][4];
    }
    stats[3 * C] = sn;
    stats[3 * C + 1] = sd;
    out[0] = (float)(sd != 0 ? sn / sd : 0.0);
    out[1] = (float)(1.0 - sp / C);
  }
}

__global__ void __launch_bounds__(kThreads)
loss_bwd_k(const float* __restrict__ z, int ld, const int32_t* __restrict__ labels,
           const float* __restrict__ weights, int ignore_index, const double* __restrict__ stats,
           float coef_ce, float coef_dice, float* __restrict__ dz, int lddz, long voxels, int C, int CB,
           int dice_softmax, const float* __restrict__ dice_weight) {
  const int t = threadIdx.x;
  const int c = t % CB, vl = t / CB, VPB = kThreads / CB;
  const bool live = c < C;
  // per-class dice constants
  float a_t = 0.f, a_s = 0.f;
  if (live) {
    double I = stats[c], den = stats[C + c] + stats[2 * C + c];
    double dc = den > 1e-6 ? den : 1e-6;
    const double wd_ = dice_weight ? (double)dice_weight[c] : 1.0;  // per = 2 w I / den
    a_t = (float)(wd_ * 2.0 / dc);                                   // d per / d s  (t term)
    a_s = (float)(den > 1e-6 ? wd_ * 4.0 * I / (dc * dc) : 0.0);     // clip passes gradient only above min
  }
  const double cden = stats[3 * C + 1];
  const float inv_den = cden != 0 ? (float)(1.0 / cden) : 0.f;
  const float kd = -coef_dice / (float)C;
  const long nvg = (voxels + VPB - 1) / VPB;  // voxel groups
  for (long g = blockIdx.x; g < nvg; g += gridDim.x) {
    const long v = g * VPB + vl;
    const bool ok = v < voxels;
    const float zc = (live && ok) ? z[v * ld + c] : 0.f;
    const float zs = (live && ok) ? zc + 1e-8f : -INFINITY;
    const float m = group_reduce(zs, CB, [](float a, float b) { return fmaxf(a, b); });
    const float e = (live && ok) ? expf(zs - m) : 0.f;
    const float se = group_reduce(e, CB, [](float a, float b) { return a + b; });
    const int y = ok ? labels[v] : -1;
    const float tt = (y == c) ? 1.f : 0.f;
    float g_d;
    if (dice_softmax) {
      // p = softmax(z): dL/dz_k = p_k (g_k - sum_j g_j p_j) with g = dL/dp
      const float m2 = group_reduce((live && ok) ? zc : -INFINITY, CB, [](float a, float b) { return fmaxf(a, b); });
      const float e2 = (live && ok) ? expf(zc - m2) : 0.f;
      const float se2 = group_reduce(e2, CB, [](float a, float b) { return a + b; });
      const float pd = ok ? e2 / se2 : 0.f;
      const float gp = (live && ok) ? kd * (a_t * tt - a_s * pd) : 0.f;
      const float dot = group_reduce(gp * pd, CB, [](float a, float b) { return a + b; });
      g_d = pd * (gp - dot);
    } else {
      const float s = 1.f / (1.f + expf(-zc));
      g_d = kd * (a_t * tt - a_s * s) * s * (1.f - s);
    }
    if (live && ok) {
      const float p = e / se;
      float g_ce = 0.f;
      if (y != ignore_index && y >= 0 && y < C) g_ce = (p - tt) * weights[y] * inv_den;
      dz[v * lddz + c] = coef_ce * g_ce + g_d;
    }
  }
}


// ---- thread-per-voxel forms for C > 4 (20-class MRI model: the lane-per-class kernels above spend two 32-lane shuffle
// reductions per voxel and leave 12 of 32 lanes idle: 0.45 + 0.49 ms per output at 512 x 512 x 12, four outputs with deep
// supervision).  A thread owns a voxel: its C logits sit in registers (float4 loads), softmax / sigmoid run over them in
// place, the per-class sums are per-thread registers reduced once per block.
// flat form of msk_tile_load for records that are not whole quads (3 classes: 12 bytes): the tile's floats are copied as they
// lie (16-byte accesses, scalar tail), a thread's record starts at float tid * C -- an odd stride for C = 3: conflict-free
__device__ __forceinline__ void tile_load_flat(const float* __restrict__ src, int nfloats, float* __restrict__ tile) {
  const int nq = nfloats >> 2;
  for (int i = threadIdx.x; i < nq; i += blockDim.x) reinterpret_cast<float4*>(tile)[i] = reinterpret_cast<const float4*>(src)[i];
  for (int i = 4 * nq + threadIdx.x; i < nfloats; i += blockDim.x) tile[i] = src[i];
}
__device__ __forceinline__ void tile_store_flat(float* __restrict__ dst, int nfloats, const float* __restrict__ tile) {
  const int nq = nfloats >> 2;
  for (int i = threadIdx.x; i < nq; i += blockDim.x) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(tile)[i];
  for (int i = 4 * nq + threadIdx.x; i < nfloats; i += blockDim.x) dst[i] = tile[i];
}

// ST (round 4): dense logits (ld == C, C % 4 == 0) travel through an LDS tile of 256 voxel records with whole-line accesses
// (msk_tile_load) -- a lane reading its own 80-byte record straight from HBM held these passes at 1.0 / 1.9 TB/s.
// ST = 2: the flat form for C <= 4 (tile_load_flat): the 3-class head of the lung model ran the lane-per-class kernels above
// (four lanes per voxel, one idle, 192 bytes per load instruction: 60 + 67 us per step at 2 x 128^3).
template <int CM, int ST>
__global__ void __launch_bounds__(kThreads)
loss_stats_tpv_k(const float* __restrict__ z, int ld, const int32_t* __restrict__ labels, const float* __restrict__ weights,
                 int ignore_index, long voxels, int C, int CB, float* __restrict__ partial /*[nb][5][CB]*/, int dice_softmax) {
  constexpr int P = (CM / 4) | 1;
  __shared__ float4 tile[ST == 1 ? kThreads * P : (ST == 2 ? kThreads * CM / 4 : 1)];
  float aI[CM], aS[CM], aT[CM], cen = 0.f, ced = 0.f;
#pragma unroll
  for (int c = 0; c < CM; ++c) aI[c] = aS[c] = aT[c] = 0.f;
  const bool v4 = (C % 4 == 0) && (ld % 4 == 0) && ((((uintptr_t)z) & 15) == 0);
  const long ntile = (voxels + kThreads - 1) / kThreads;
  for (long it = blockIdx.x; it < ntile; it += gridDim.x) {
    const long v = it * kThreads + threadIdx.x;
    if (ST) {
      const long v0 = it * kThreads;
      const int nv = (int)(voxels - v0 < kThreads ? voxels - v0 : kThreads);
      __syncthreads();   // the previous tile's records have been taken
      if (ST == 2) tile_load_flat(z + v0 * C, nv * C, reinterpret_cast<float*>(tile));
      else msk_tile_load(z + v0 * C, nv * (C >> 2), C >> 2, P, tile);
      __syncthreads();
    }
    if (v >= voxels) continue;   // (after the barriers: every thread of the block reaches them)
    float zz[CM];
    const float* zp = z + v * ld;
    if (ST == 2) {
#pragma unroll
      for (int c = 0; c < CM; ++c) zz[c] = c < C ? reinterpret_cast<const float*>(tile)[threadIdx.x * C + c] : 0.f;
    } else if (ST) {
#pragma unroll
      for (int c = 0; c < CM; c += 4) {
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C) q = tile[threadIdx.x * P + (c >> 2)];
        zz[c] = q.x; zz[c + 1] = q.y; zz[c + 2] = q.z; zz[c + 3] = q.w;
      }
    } else if (v4) {
#pragma unroll
      for (int c = 0; c < CM; c += 4) {
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C) q = *reinterpret_cast<const float4*>(zp + c);
        zz[c] = q.x; zz[c + 1] = q.y; zz[c + 2] = q.z; zz[c + 3] = q.w;
      }
    } else {
#pragma unroll
      for (int c = 0; c < CM; ++c) zz[c] = c < C ? zp[c] : 0.f;
    }
    const int y = labels[v];
    float m = -INFINITY, m2 = -INFINITY;
#pragma unroll
    for (int c = 0; c < CM; ++c)
      if (c < C) {
        m = fmaxf(m, zz[c] + 1e-8f);  // cross_entropy_loss.py:79 logit + EPS
        m2 = fmaxf(m2, zz[c]);
      }
    float se = 0.f, se2 = 0.f, zy = 0.f;
#pragma unroll
    for (int c = 0; c < CM; ++c)
      if (c < C) {
        se += expf(zz[c] + 1e-8f - m);
        if (dice_softmax) se2 += expf(zz[c] - m2);
        if (c == y) zy = zz[c] + 1e-8f;
      }
    const float ise2 = dice_softmax ? 1.f / se2 : 0.f;
#pragma unroll
    for (int c = 0; c < CM; ++c)
      if (c < C) {
        const float sgm = dice_softmax ? expf(zz[c] - m2) * ise2 : 1.f / (1.f + expf(-zz[c]));
        aS[c] = fmaf(sgm, sgm, aS[c]);
        if (c == y) {
          aI[c] += sgm;
          aT[c] += 1.f;
        }
      }
    if (y != ignore_index && y >= 0 && y < C) {
      const float wy = weights[y];
      cen = fmaf(wy, -(zy - m - logf(se)), cen);
      ced += wy;
    }
  }
  // block reduction: shuffle tree per value, then the four wavefronts' results through LDS (fixed order)
  __shared__ float sh[kThreads / 64][3 * CM + 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto wsum = [](float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
#pragma unroll
  for (int c = 0; c < CM; ++c) {
    const float a = wsum(aI[c]), b = wsum(aS[c]), t = wsum(aT[c]);
    if (lane == 0) {
      sh[wave][c] = a;
      sh[wave][CM + c] = b;
      sh[wave][2 * CM + c] = t;
    }
  }
  {
    const float a = wsum(cen), b = wsum(ced);
    if (lane == 0) {
      sh[wave][3 * CM] = a;
      sh[wave][3 * CM + 1] = b;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 5 * CB; i += blockDim.x) {
    const int kq = i / CB, c = i - kq * CB;
    float v = 0.f;
    if (c < C) {
      const int idx = kq < 3 ? kq * CM + c : (c == 0 ? 3 * CM + (kq - 3) : -1);  // the CE sums ride in class slot 0
      if (idx >= 0)
        for (int w = 0; w < kThreads / 64; ++w) v += sh[w][idx];
    }
    partial[((long)blockIdx.x * 5 + kq) * CB + c] = v;
  }
}

#ifndef LOSS_BWD_LB
#define LOSS_BWD_LB 3   // wavefronts per SIMD the register allocation aims for (A/B: -DLOSS_BWD_LB=2 = the compiler's own choice, 230-249 registers)
#endif
template <int CM, int ST>
__global__ void __launch_bounds__(kThreads, ST ? LOSS_BWD_LB : 2)
loss_bwd_tpv_k(const float* __restrict__ z, int ld, const int32_t* __restrict__ labels, const float* __restrict__ weights,
               int ignore_index, const double* __restrict__ stats, float coef_ce, float coef_dice, float* __restrict__ dz, int lddz,
               long voxels, int C, int dice_softmax, const float* __restrict__ dice_weight) {
  __shared__ float s_at[CM], s_as[CM], s_w[CM];
  if (threadIdx.x < CM) {
    const int c = threadIdx.x;
    float a_t = 0.f, a_s = 0.f;
    if (c < C) {
      const double I = stats[c], den = stats[C + c] + stats[2 * C + c];
      const double dc = den > 1e-6 ? den : 1e-6;
      const double wd_ = dice_weight ? (double)dice_weight[c] : 1.0;
      a_t = (float)(wd_ * 2.0 / dc);
      a_s = (float)(den > 1e-6 ? wd_ * 4.0 * I / (dc * dc) : 0.0);
    }
    s_at[c] = a_t;
    s_as[c] = a_s;
    s_w[c] = c < C ? weights[c] : 0.f;
  }
  __syncthreads();
  const double cden = stats[3 * C + 1];
  const float inv_den = cden != 0 ? (float)(1.0 / cden) : 0.f;
  const float kd = -coef_dice / (float)C;
  const bool v4 = (C % 4 == 0) && (ld % 4 == 0) && (lddz % 4 == 0) && (((((uintptr_t)z) | ((uintptr_t)dz)) & 15) == 0);
  constexpr int P = (CM / 4) | 1;
  __shared__ float4 tile[ST == 1 ? kThreads * P : (ST == 2 ? kThreads * CM / 4 : 1)];
  const long ntile = (voxels + kThreads - 1) / kThreads;
  for (long it = blockIdx.x; it < ntile; it += gridDim.x) {
    const long v = it * kThreads + threadIdx.x;
    const long v0 = it * kThreads;
    const int nv = (int)(voxels - v0 < kThreads ? voxels - v0 : kThreads);
    if (ST) {
      __syncthreads();   // the previous tile's store pass is done
      if (ST == 2) tile_load_flat(z + v0 * C, nv * C, reinterpret_cast<float*>(tile));
      else msk_tile_load(z + v0 * C, nv * (C >> 2), C >> 2, P, tile);
      __syncthreads();
    }
    const bool mine = v < voxels;
    float zz[CM], g[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) g[c] = 0.f;
    const float* zp = z + (mine ? v : 0) * ld;
    if (ST == 2) {
#pragma unroll
      for (int c = 0; c < CM; ++c) zz[c] = (c < C && mine) ? reinterpret_cast<const float*>(tile)[threadIdx.x * C + c] : 0.f;
    } else if (ST) {
#pragma unroll
      for (int c = 0; c < CM; c += 4) {
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C && mine) q = tile[threadIdx.x * P + (c >> 2)];
        zz[c] = q.x; zz[c + 1] = q.y; zz[c + 2] = q.z; zz[c + 3] = q.w;
      }
    } else if (!mine) {
      continue;
    } else if (v4) {
#pragma unroll
      for (int c = 0; c < CM; c += 4) {
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C) q = *reinterpret_cast<const float4*>(zp + c);
        zz[c] = q.x; zz[c + 1] = q.y; zz[c + 2] = q.z; zz[c + 3] = q.w;
      }
    } else {
#pragma unroll
      for (int c = 0; c < CM; ++c) zz[c] = c < C ? zp[c] : 0.f;
    }
    const int y = mine ? labels[v] : -1;
    float m = -INFINITY, m2 = -INFINITY;
#pragma unroll
    for (int c = 0; c < CM; ++c)
      if (c < C) {
        m = fmaxf(m, zz[c] + 1e-8f);
        m2 = fmaxf(m2, zz[c]);
      }
    float se = 0.f, se2 = 0.f;
#pragma unroll
    for (int c = 0; c < CM; ++c)
      if (c < C) {
        se += expf(zz[c] + 1e-8f - m);
        if (dice_softmax) se2 += expf(zz[c] - m2);
      }
    const bool ce_on = y != ignore_index && y >= 0 && y < C;
    const float wy = ce_on ? s_w[y] * inv_den : 0.f;
    const float ise = 1.f / se, ise2 = dice_softmax ? 1.f / se2 : 0.f;
    float dot = 0.f;
    float gp[CM];  // softmax-normalised dice: dL/dp per class (second sweep needs sum_j gp_j p_j)
#pragma unroll
    for (int c = 0; c < CM; ++c)
      if (c < C) {
        const float tt = (y == c) ? 1.f : 0.f;
        const float p = expf(zz[c] + 1e-8f - m) * ise;
        g[c] = coef_ce * (p - tt) * wy;
        if (dice_softmax) {
          const float pd = expf(zz[c] - m2) * ise2;
          gp[c] = kd * (s_at[c] * tt - s_as[c] * pd);
          dot = fmaf(gp[c], pd, dot);
          zz[c] = pd;  // kept for the second sweep
        } else {
          const float sg = 1.f / (1.f + expf(-zz[c]));
          g[c] += kd * (s_at[c] * tt - s_as[c] * sg) * sg * (1.f - sg);
        }
      }
    if (dice_softmax) {
#pragma unroll
      for (int c = 0; c < CM; ++c)
        if (c < C) g[c] += zz[c] * (gp[c] - dot);  // p = softmax(z): dL/dz_k = p_k (g_k - sum_j g_j p_j)
    }
    if (ST == 2) {
      __syncthreads();   // every thread has taken its record
#pragma unroll
      for (int c = 0; c < CM; ++c)
        if (c < C) reinterpret_cast<float*>(tile)[threadIdx.x * C + c] = g[c];
      __syncthreads();
      tile_store_flat(dz + v0 * C, nv * C, reinterpret_cast<const float*>(tile));
      continue;
    }
    if (ST) {
      __syncthreads();   // every thread has taken its record
#pragma unroll
      for (int c = 0; c < CM; c += 4)
        if (c < C) tile[threadIdx.x * P + (c >> 2)] = make_float4(g[c], g[c + 1], g[c + 2], g[c + 3]);
      __syncthreads();
      msk_tile_store(dz + v0 * C, nv * (C >> 2), C >> 2, P, tile, false);
      continue;
    }
    float* dp = dz + v * lddz;
    if (v4) {
#pragma unroll
      for (int c = 0; c < CM; c += 4)
        if (c < C) *reinterpret_cast<float4*>(dp + c) = make_float4(g[c], g[c + 1], g[c + 2], g[c + 3]);
    } else {
#pragma unroll
      for (int c = 0; c < CM; ++c)
        if (c < C) dp[c] = g[c];
    }
  }
}

__global__ void __launch_bounds__(kThreads)
sgd_momentum_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ vel, size_t n4,
               size_t n, float lr, float mu, float wd, float gs) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 vv = reinterpret_cast<float4*>(vel)[i];
    float t;
    t = fmaf(wd, pp.x, gg.x * gs); vv.x = fmaf(mu, vv.x, t); pp.x = fmaf(-lr, vv.x, pp.x);
    t = fmaf(wd, pp.y, gg.y * gs); vv.y = fmaf(mu, vv.y, t); pp.y = fmaf(-lr, vv.y, pp.y);
    t = fmaf(wd, pp.z, gg.z * gs); vv.z = fmaf(mu, vv.z, t); pp.z = fmaf(-lr, vv.z, pp.z);
    t = fmaf(wd, pp.w, gg.w * gs); vv.w = fmaf(mu, vv.w, t); pp.w = fmaf(-lr, vv.w, pp.w);
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(vel)[i] = vv;
  }
  // tail
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float t = fmaf(wd, p[i], g[i] * gs);
    float v = fmaf(mu, vel[i], t);
    vel[i] = v;
    p[i] = fmaf(-lr, v, p[i]);
  }
}

// paddle.optimizer.Adam (cvlibs/config.py:214-216; Paddle's adam kernel, which is not part of the reference tree):
//   g += wd * p (float weight_decay = L2Decay);  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;
//   p -= lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps sqrt(1 - b2^t))
__global__ void __launch_bounds__(kThreads)
adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m1, float* __restrict__ m2, size_t n,
       float lr_t, float b1, float b2, float eps_t, float wd, float gs) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float pp = p[i];
    const float gg = fmaf(wd, pp, g[i] * gs);
    const float a = fmaf(b1, m1[i], (1.f - b1) * gg);
    const float b = fmaf(b2, m2[i], (1.f - b2) * gg * gg);
    m1[i] = a;
    m2[i] = b;
    p[i] = pp - lr_t * (a / (sqrtf(b) + eps_t));
  }
}

inline int stat_blocks(long voxels, int VPB, int num_cu) {
  long want = (voxels + (long)VPB * 32 - 1) / ((long)VPB * 32);
  long cap = (long)num_cu * 8;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  return (int)want;
}

}  // namespace

extern "C" {

int msk_class_weights(msk_ctx* ctx, msk_tensor logits, float* weights) {
  const int C = logits.c;
  MSK_REQUIRE(ctx, C >= 1 && C <= 64, "num_classes must be in [1,64]");
  const int CB = pow2ceil(C);
  const long voxels = msk_voxels(logits);
  const int nb = stat_blocks(voxels, kThreads / CB, ctx->num_cu);
  float* partial = (float*)msk_workspace(ctx, (size_t)nb * CB * sizeof(float));
  if (!partial) return -1;
  {
    msk_launch_scope ls(ctx, "loss_class_weights");
    hipLaunchKernelGGL(loss_stats_k<0>, dim3(nb), dim3(kThreads), 0, ctx->stream, (const float*)logits.p,
                       logits.ld, (const int32_t*)nullptr, (const float*)nullptr, 0, voxels, C, CB, partial, 0);
    MSK_LAUNCH_CHECK(ctx);
  }
  {
    msk_launch_scope ls(ctx, "loss_class_weights_final");
    hipLaunchKernelGGL(class_weights_final_k, dim3(1), dim3(64), 0, ctx->stream, partial, nb, C, CB,
                       (double)voxels, weights);
    MSK_LAUNCH_CHECK(ctx);
  }
  return 0;
}

int msk_loss_fwd_ex(msk_ctx* ctx, msk_tensor logits, const int32_t* labels, const float* weights,
                    int ignore_index, int dice_softmax, const float* dice_weight, float* out, double* stats) {
  const int C = logits.c;
  MSK_REQUIRE(ctx, C >= 1 && C <= 64, "num_classes must be in [1,64]");
  const int CB = pow2ceil(C);
  const long voxels = msk_voxels(logits);
  int nb = stat_blocks(voxels, kThreads / CB, ctx->num_cu);
  const bool flat4 = (ctx->tile_staging & 4) && C >= 2 && C <= 4 && logits.ld == C && (((uintptr_t)logits.p) & 15) == 0;
  if ((C > 4 && C <= 32) || flat4) {
    // thread-per-voxel forms: a workgroup covers 256 voxels per round, three workgroups per CU are resident -- more records only
    // lengthen the one-workgroup merge (20 classes: 2048 records x 160 values took loss_final_k 107 us per output)
    const long tiles = (voxels + kThreads - 1) / kThreads;
    nb = (int)(tiles < 3L * ctx->num_cu ? tiles : 3L * ctx->num_cu);
  }
  float* partial = (float*)msk_workspace(ctx, (size_t)nb * 5 * CB * sizeof(float));
  if (!partial) return -1;
  {
    msk_launch_scope ls(ctx, "loss_fwd_stats");
    if (C > 4 && C <= 32) {  // thread-per-voxel form
      const bool st = (ctx->tile_staging & 2) && logits.ld == C && C % 4 == 0 && (((uintptr_t)logits.p) & 15) == 0;
#define LOSS_STATS_TPV(CM_, ST_)                                                                                               \
  hipLaunchKernelGGL((loss_stats_tpv_k<CM_, ST_>), dim3(nb), dim3(kThreads), 0, ctx->stream, (const float*)logits.p, logits.ld, \
                     labels, weights, ignore_index, voxels, C, CB, partial, dice_softmax)
      // register arrays sized to the class count in steps of a quad (round 4: 20 classes ran the 32-slot instantiation)
#define LOSS_STATS_CM(CM_) { if (st) LOSS_STATS_TPV(CM_, 1); else LOSS_STATS_TPV(CM_, 0); }
      if (C <= 8) LOSS_STATS_CM(8) else if (C <= 12) LOSS_STATS_CM(12) else if (C <= 16) LOSS_STATS_CM(16)
      else if (C <= 20) LOSS_STATS_CM(20) else if (C <= 24) LOSS_STATS_CM(24) else LOSS_STATS_CM(32)
#undef LOSS_STATS_CM
#undef LOSS_STATS_TPV
    } else if (flat4) {
      // dense 2..4-class logits (the lung model's head): thread per voxel, records through a flat LDS tile
      hipLaunchKernelGGL((loss_stats_tpv_k<4, 2>), dim3(nb), dim3(kThreads), 0, ctx->stream, (const float*)logits.p, logits.ld,
                         labels, weights, ignore_index, voxels, C, CB, partial, dice_softmax);
    } else {
      hipLaunchKernelGGL(loss_stats_k<1>, dim3(nb), dim3(kThreads), 0, ctx->stream, (const float*)logits.p,
                         logits.ld, labels, weights, ignore_index, voxels, C, CB, partial, dice_softmax);
    }
    MSK_LAUNCH_CHECK(ctx);
  }
  {
    msk_launch_scope ls(ctx, "loss_fwd_final");
    hipLaunchKernelGGL(loss_final_k, dim3(1), dim3(1024), 0, ctx->stream, partial, nb, C, CB, out, stats, dice_weight);
    MSK_LAUNCH_CHECK(ctx);
  }
  return 0;
}

int msk_loss_fwd(msk_ctx* ctx, msk_tensor logits, const int32_t* labels, const float* weights,
                 int ignore_index, float* out, double* stats) {
  return msk_loss_fwd_ex(ctx, logits, labels, weights, ignore_index, 0, nullptr, out, stats);
}

int msk_loss_bwd_ex(msk_ctx* ctx, msk_tensor logits, const int32_t* labels, const float* weights,
                    int ignore_index, int dice_softmax, const float* dice_weight, const double* stats, float coef_ce,
                    float coef_dice, msk_tensor dlogits) {
  const int C = logits.c;
  MSK_REQUIRE(ctx, C >= 1 && C <= 64, "num_classes must be in [1,64]");
  MSK_REQUIRE(ctx, dlogits.c == C && msk_voxels(dlogits) == msk_voxels(logits), "dlogits shape mismatch");
  const int CB = pow2ceil(C);
  const long voxels = msk_voxels(logits);
  const int VPB = kThreads / CB;
  long nvg = (voxels + VPB - 1) / VPB;
  long blocks = nvg < (long)ctx->num_cu * 16 ? nvg : (long)ctx->num_cu * 16;
  msk_launch_scope ls(ctx, "loss_bwd");
  if (C > 4 && C <= 32) {  // thread-per-voxel form
    long tb = (voxels + kThreads - 1) / kThreads;
    if (tb > (long)ctx->num_cu * 16) tb = (long)ctx->num_cu * 16;
    const bool st = (ctx->tile_staging & 2) && logits.ld == C && dlogits.ld == C && C % 4 == 0 &&
                    ((((uintptr_t)logits.p) | ((uintptr_t)dlogits.p)) & 15) == 0;
#define LOSS_BWD_TPV(CM_, ST_)                                                                                                  \
  hipLaunchKernelGGL((loss_bwd_tpv_k<CM_, ST_>), dim3((int)tb), dim3(kThreads), 0, ctx->stream, (const float*)logits.p, logits.ld, \
                     labels, weights, ignore_index, stats, coef_ce, coef_dice, (float*)dlogits.p, dlogits.ld, voxels, C,        \
                     dice_softmax, dice_weight)
#define LOSS_BWD_CM(CM_) { if (st) LOSS_BWD_TPV(CM_, 1); else LOSS_BWD_TPV(CM_, 0); }
    if (C <= 8) LOSS_BWD_CM(8) else if (C <= 12) LOSS_BWD_CM(12) else if (C <= 16) LOSS_BWD_CM(16)
    else if (C <= 20) LOSS_BWD_CM(20) else if (C <= 24) LOSS_BWD_CM(24) else LOSS_BWD_CM(32)
#undef LOSS_BWD_CM
#undef LOSS_BWD_TPV
  } else if ((ctx->tile_staging & 4) && C >= 2 && C <= 4 && logits.ld == C && dlogits.ld == C &&
             ((((uintptr_t)logits.p) | ((uintptr_t)dlogits.p)) & 15) == 0) {
    long tb = (voxels + kThreads - 1) / kThreads;
    if (tb > (long)ctx->num_cu * 16) tb = (long)ctx->num_cu * 16;
    hipLaunchKernelGGL((loss_bwd_tpv_k<4, 2>), dim3((int)tb), dim3(kThreads), 0, ctx->stream, (const float*)logits.p, logits.ld,
                       labels, weights, ignore_index, stats, coef_ce, coef_dice, (float*)dlogits.p, dlogits.ld, voxels, C,
                       dice_softmax, dice_weight);
  } else
  hipLaunchKernelGGL(loss_bwd_k, dim3((int)blocks), dim3(kThreads), 0, ctx->stream, (const float*)logits.p,
                     logits.ld, labels, weights, ignore_index, stats, coef_ce, coef_dice, (float*)dlogits.p,
                     dlogits.ld, voxels, C, CB, dice_softmax, dice_weight);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

int msk_loss_bwd(msk_ctx* ctx, msk_tensor logits, const int32_t* labels, const float* weights,
                 int ignore_index, const double* stats, float coef_ce, float coef_dice, msk_tensor dlogits) {
  return msk_loss_bwd_ex(ctx, logits, labels, weights, ignore_index, 0, nullptr, stats, coef_ce, coef_dice, dlogits);
}

static int sgd_launch(msk_ctx* ctx, float* param, const float* grad, float* velocity, size_t count, float lr, float momentum,
                      float weight_decay, float grad_scale) {
  if (count == 0) return 0;
  size_t n4 = count / 4;
  long blocks = (long)((n4 + kThreads - 1) / kThreads);
  long cap = (long)ctx->num_cu * 16;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  msk_launch_scope ls(ctx, "sgd_momentum");
  hipLaunchKernelGGL(sgd_momentum_k, dim3((int)blocks), dim3(kThreads), 0, ctx->stream, param, grad, velocity, n4,
                     count, lr, momentum, weight_decay, grad_scale);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

int msk_sgd_momentum(msk_ctx* ctx, float* param, const float* grad, float* velocity, size_t count, float lr,
                     float momentum, float weight_decay, float grad_scale) {
  if (count == 0) return 0;
  MSK_REQUIRE(ctx, ((uintptr_t)param % 16 == 0) && ((uintptr_t)grad % 16 == 0) && ((uintptr_t)velocity % 16 == 0),
              "arenas must be 16-byte aligned");
  // the "late" weight gradient (in_tr.conv1, the last of the backward pass: msk_conv.hip bwd_bnact_c1) may still be running at
  // the end of the side stream: everything else is complete behind ev_late -- update it and re-pack the weights now, the late
  // tensor after the join
  size_t lo = 0, hi = 0;
  if (ctx->late_valid && ctx->late_ptr >= grad && ctx->late_ptr + ctx->late_count <= grad + count &&
      ((ctx->late_ptr - grad) & 3) == 0) {
    lo = (size_t)(ctx->late_ptr - grad);
    hi = (lo + ctx->late_count + 3) & ~(size_t)3;
    if (hi > count) hi = count;
    MSK_CHECK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_late, 0));
    if (sgd_launch(ctx, param, grad, velocity, lo, lr, momentum, weight_decay, grad_scale) != 0) return -1;
    if (sgd_launch(ctx, param + hi, grad + hi, velocity + hi, count - hi, lr, momentum, weight_decay, grad_scale) != 0) return -1;
    msk_weights_changed_impl(ctx, param, count * sizeof(float));
    if (msk_wbf_prepack_impl(ctx) != 0) return -1;
    if (msk_join_side_impl(ctx) != 0) return -1;
    if (sgd_launch(ctx, param + lo, grad + lo, velocity + lo, hi - lo, lr, momentum, weight_decay, grad_scale) != 0) return -1;
    // the late slice changes AFTER the re-pack above: a packed row over it (none today -- the late tensor is in_tr.conv1, one
    // input channel, which never enters the pack cache) must not be left marked valid (advisor, round 4)
    msk_weights_changed_impl(ctx, param + lo, (hi - lo) * sizeof(float));
    return msk_wbf_prepack_range_impl(ctx, param + lo, (hi - lo) * sizeof(float));
  }
  if (msk_join_side_impl(ctx) != 0) return -1;  // weight gradients may still be running on the side stream
  if (sgd_launch(ctx, param, grad, velocity, count, lr, momentum, weight_decay, grad_scale) != 0) return -1;
  // the packed / transformed forms of the convolution weights inside [param, param + count) are stale now: rebuild the ones
  // in use in one go (two launches instead of one pack + one maximum per layer and direction inside the next step)
  msk_weights_changed_impl(ctx, param, count * sizeof(float));
  return msk_wbf_prepack_impl(ctx);
}

// Eager update of ONE slice of the arena (round 5; round-4 "measured but not built (i)"): everything that produces the
// gradients of [param, param + count) has been enqueued -- the weight gradients on the side stream, the BatchNorm / PReLU
// parameter gradients and the data gradients that READ these weights on the compute stream.  The update and the re-pack of
// the slice's convolution weights go to the END OF THE SIDE STREAM behind an event on the compute stream's current tail:
// nothing on the compute stream waits for them until msk_sgd_momentum_finish, so 0.165 (update) + 0.33 (pack) + 0.08 ms
// (maxima) leave the tail of the step's critical path.  Same kernel, same arithmetic per element as msk_sgd_momentum.
int msk_sgd_momentum_eager(msk_ctx* ctx, float* param, const float* grad, float* velocity, size_t count, float lr,
                           float momentum, float weight_decay, float grad_scale) {
  if (count == 0) return 0;
  MSK_REQUIRE(ctx, ((uintptr_t)param % 16 == 0) && ((uintptr_t)grad % 16 == 0) && ((uintptr_t)velocity % 16 == 0),
              "arena slices must be 16-byte aligned");
  // Behind a LATE weight gradient (in_tr.conv1's, the last kernel of the side stream: msk_conv.hip bwd_bnact_c1) the side stream
  // is the step's critical path and the compute stream is idle: everything else this slice needs is complete behind ev_late,
  // so the update and the re-pack run on the compute stream beside that kernel (0.09 ms off the tail of a VNet step); the
  // late tensor's own slice waits for msk_sgd_momentum_finish.  Same kernels, same arithmetic per element.
  if (ctx->late_valid && ctx->eager_tail_main && ctx->wgrad_async && ctx->side != nullptr) {
    const float* lp = ctx->late_ptr;
    const bool inside = lp >= grad && lp + ctx->late_count <= grad + count && ((lp - grad) & 3) == 0 && ctx->late_piece.count == 0;
    const bool disjoint = lp + ctx->late_count <= grad || lp >= grad + count;
    if (inside || disjoint) {
      size_t lo = count, hi = count;
      if (inside) {
        lo = (size_t)(lp - grad);
        hi = (lo + ctx->late_count + 3) & ~(size_t)3;
        if (hi > count) hi = count;
      }
      MSK_CHECK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_late, 0));
      if (sgd_launch(ctx, param, grad, velocity, lo, lr, momentum, weight_decay, grad_scale) != 0) return -1;
      if (sgd_launch(ctx, param + hi, grad + hi, velocity + hi, count - hi, lr, momentum, weight_decay, grad_scale) != 0) return -1;
      msk_weights_changed_impl(ctx, param, count * sizeof(float));
      if (msk_wbf_prepack_range_impl(ctx, param, count * sizeof(float)) != 0) return -1;
      if (inside) {
        ctx->late_piece.param = param + lo; ctx->late_piece.grad = grad + lo; ctx->late_piece.velocity = velocity + lo;
        ctx->late_piece.count = hi - lo;
        ctx->late_piece.lr = lr; ctx->late_piece.momentum = momentum; ctx->late_piece.weight_decay = weight_decay;
        ctx->late_piece.grad_scale = grad_scale;
      }
      return 0;
    }
  }
  {
    msk_side_scope side(ctx);   // (no side stream: the same launches on the calling stream, in order)
    if (sgd_launch(ctx, param, grad, velocity, count, lr, momentum, weight_decay, grad_scale) != 0) return -1;
    msk_weights_changed_impl(ctx, param, count * sizeof(float));
    if (msk_wbf_prepack_range_impl(ctx, param, count * sizeof(float)) != 0) return -1;
  }
  return 0;
}

// End of an eagerly updated step: the compute stream waits for the side stream (updates, re-packs, a late weight gradient),
// packed rows that are still stale are rebuilt and the pack cache's use epoch advances as in msk_sgd_momentum.
int msk_sgd_momentum_finish(msk_ctx* ctx) {
  if (msk_join_side_impl(ctx) != 0) return -1;
  if (ctx->late_piece.count != 0) {   // the late tensor's slice of an update that ran beside its weight gradient (above)
    auto lp = ctx->late_piece;
    ctx->late_piece.count = 0;
    if (sgd_launch(ctx, lp.param, lp.grad, lp.velocity, lp.count, lp.lr, lp.momentum, lp.weight_decay, lp.grad_scale) != 0) return -1;
    msk_weights_changed_impl(ctx, lp.param, lp.count * sizeof(float));
  }
  return msk_wbf_prepack_impl(ctx);
}

int msk_adam(msk_ctx* ctx, float* param, const float* grad, float* moment1, float* moment2, size_t count, float lr,
             float beta1, float beta2, float epsilon, double beta1_pow, double beta2_pow, float weight_decay,
             float grad_scale) {
  if (count == 0) return 0;
  if (msk_join_side_impl(ctx) != 0) return -1;  // weight gradients may still be running on the side stream
  MSK_REQUIRE(ctx, beta1_pow < 1.0 && beta2_pow < 1.0 && beta1_pow >= 0.0 && beta2_pow >= 0.0, "beta powers must be in [0, 1)");
  const double c2 = sqrt(1.0 - beta2_pow);
  const float lr_t = (float)((double)lr * c2 / (1.0 - beta1_pow));
  const float eps_t = (float)((double)epsilon * c2);
  long blocks = (long)((count + kThreads - 1) / kThreads);
  const long cap = (long)ctx->num_cu * 16;
  if (blocks > cap) blocks = cap;
  {
  msk_launch_scope ls(ctx, "adam");
  hipLaunchKernelGGL(adam_k, dim3((int)blocks), dim3(kThreads), 0, ctx->stream, param, grad, moment1, moment2, count, lr_t,
                     beta1, beta2, eps_t, weight_decay, grad_scale);
  MSK_LAUNCH_CHECK(ctx);
  }
  msk_weights_changed_impl(ctx, param, count * sizeof(float));
  return msk_wbf_prepack_impl(ctx);
}

}  // extern "C"
