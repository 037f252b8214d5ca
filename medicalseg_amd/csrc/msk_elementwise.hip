// HBM-bound per-voxel kernels of the VNet path on gfx950: BatchNorm statistics,
// fused affine(+residual)(+PReLU) forward/backward, channel-slice copies
// (concat / Dropout3D), layout changes at the boundary, argmax.
//
// Layout: NDHWC fp32 with an explicit voxel stride `ld` (channel slices of a wider
// buffer are first-class).  Threads map (channel fastest) so that a wavefront
// reads 256 contiguous bytes (or 1 KiB with the float4 variants).
#include "msk_common.h"
#include "msk_wbf.h"

namespace {

constexpr int kThreads = 256;

inline int pow2ceil(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// Channel-block geometry for per-channel reductions: CB channels per block row
// (power of two <= 256), VPB = 256/CB voxel lanes.
struct ChanGeom {
  int CB, VPB, cblocks;
};
inline ChanGeom chan_geom(int C) {
  ChanGeom g;
  g.CB = pow2ceil(C) < kThreads ? pow2ceil(C) : kThreads;
  g.VPB = kThreads / g.CB;
  g.cblocks = (C + g.CB - 1) / g.CB;
  return g;
}

// Register budget of the big HBM-bound passes (A/B: tools/ab_build.sh ... -DEW_LB=n): they share the chip with the weight-gradient
// stream, whose three resident wavefronts per SIMD leave 104 of 512 vector registers
#ifndef EW_LB
#define EW_LB 0
#endif
#if EW_LB > 0
#define EW_BOUNDS __launch_bounds__(kThreads, EW_LB)
#else
#define EW_BOUNDS __launch_bounds__(kThreads)
#endif
int g_ew_cap = 4, g_reduce_cap = 4;   // blocks per CU of the elementwise / reduction kernels (options "ew_cap", "reduce_cap"); measured: 4 resident blocks per CU with grid-stride loops beat 32 queued ones by 0.25 ms per step; reduction kernels (round 4, after reduce_vpl 8; A/B on two boxes, two repetitions each, tools/ab_opt_sweep.sh): 8 / 6 / 5 / 4 / 3 / 2 per CU = 19.31 / -- / 19.34 / 19.24 / 19.26 / 19.40 ms and 18.87 / 18.89 / -- / 18.67 / -- / -- ms -- half the partial records for the merges, same serialized time
int g_reduce_vpl = 8;   // option "reduce_vpl": voxels per lane the reduction kernels aim for before they add workgroups (round 4: 64 left the deep levels with 4-64 workgroups of 32-64 dependent iterations: 40-110 us per pass for tensors of 2-30 MB; A/B 64 / 32 / 16 / 8 / 4: bn_prelu_join bucket 3.97 / 3.70 / 3.61 / 3.59 / 3.63 ms)
int g_reduce_vpl_site[4] = {0, 0, 0, 0};   // debug option "reduce_vpl_site" (site * 1000 + voxels per lane): 0 statistics, 1 BatchNorm backward sums, 2 joins, 3 channel sums
inline int reduce_blocks(long voxels, int VPB, int num_cu, int site) {
  const int vpl = g_reduce_vpl_site[site] > 0 ? g_reduce_vpl_site[site] : g_reduce_vpl;
  long want = (voxels + (long)VPB * vpl - 1) / ((long)VPB * vpl);  // >= vpl voxels per lane
  long cap = (long)num_cu * g_reduce_cap;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  return (int)want;
}

// ---------------------------------------------------------------------------
// BatchNorm statistics: shifted sums per thread, Chan merge across threads/blocks
// ---------------------------------------------------------------------------
struct WF {
  float n, mean, m2;
};
__device__ __forceinline__ WF wf_merge(WF a, WF b) {
  if (b.n == 0.f) return a;
  if (a.n == 0.f) return b;
  WF r;
  r.n = a.n + b.n;
  float d = b.mean - a.mean;
  float f = b.n / r.n;
  r.mean = a.mean + d * f;
  r.m2 = a.m2 + b.m2 + d * d * a.n * f;
  return r;
}

__global__ void __launch_bounds__(kThreads)
bn_stats_partial(const float* __restrict__ x, long voxels, int C, int ld, int CB, int VPB,
                 float* __restrict__ partial /*[cblocks][nb][CB][3]*/) {
  __shared__ WF sh[kThreads];
  const int t = threadIdx.x;
  const int cl = t % CB, vl = t / CB;
  const int c = blockIdx.y * CB + cl;
  const int nb = gridDim.x;
  const long per = (voxels + nb - 1) / nb;
  const long v0 = (long)blockIdx.x * per;
  long v1 = v0 + per;
  if (v1 > voxels) v1 = voxels;
  WF w = {0.f, 0.f, 0.f};
  if (c < C && v0 + vl < v1) {
    const float K = x[(v0 + vl) * ld + c];
    float s1 = 0.f, s2 = 0.f, n = 0.f;
    for (long v = v0 + vl; v < v1; v += VPB) {
      float d = x[v * ld + c] - K;
      s1 += d;
      s2 = fmaf(d, d, s2);
      n += 1.f;
    }
    w.n = n;
    w.mean = K + s1 / n;
    w.m2 = fmaxf(s2 - s1 * s1 / n, 0.f);
  }
  sh[t] = w;
  __syncthreads();
  for (int s = VPB >> 1; s > 0; s >>= 1) {
    if (vl < s) sh[t] = wf_merge(sh[t], sh[t + s * CB]);
    __syncthreads();
  }
  if (vl == 0) {
    float* p = partial + (((long)blockIdx.y * nb + blockIdx.x) * CB + cl) * 3;
    p[0] = sh[t].n;
    p[1] = sh[t].mean;
    p[2] = sh[t].m2;
  }
}

// float4 variant: one channel quad per thread, two voxels in flight (the scalar kernel reached 3.2 TB/s)
__global__ void __launch_bounds__(kThreads)
bn_stats_partial_v4(const float* __restrict__ x, long voxels, int C, int ld, int QCB, int VL,
                    float* __restrict__ partial /*[nb][4*QCB][3]*/) {
  __shared__ WF sh[4][kThreads];
  const int t = threadIdx.x;
  const int cq = t % QCB, vl = t / QCB;
  const int c = cq * 4;
  const int nb = gridDim.x;
  const long per = (voxels + nb - 1) / nb;
  const long v0 = (long)blockIdx.x * per;
  long v1 = v0 + per;
  if (v1 > voxels) v1 = voxels;
  WF w[4] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  if (c < C && v0 + vl < v1) {
    const float4 K = *reinterpret_cast<const float4*>(x + (v0 + vl) * ld + c);
    const float k[4] = {K.x, K.y, K.z, K.w};
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    float n = 0.f;
    auto acc = [&](const float4 q) {
      const float xv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = xv[j] - k[j];
        s1[j] += d;
        s2[j] = fmaf(d, d, s2[j]);
      }
      n += 1.f;
    };
    long v = v0 + vl;
    for (; v + VL < v1; v += 2 * VL) {
      const float4 a = *reinterpret_cast<const float4*>(x + v * ld + c);
      const float4 b = *reinterpret_cast<const float4*>(x + (v + VL) * ld + c);
      acc(a);
      acc(b);
    }
    if (v < v1) acc(*reinterpret_cast<const float4*>(x + v * ld + c));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w[j].n = n;
      w[j].mean = k[j] + s1[j] / n;
      w[j].m2 = fmaxf(s2[j] - s1[j] * s1[j] / n, 0.f);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) sh[j][t] = w[j];
  __syncthreads();
  for (int s = VL >> 1; s > 0; s >>= 1) {
    if (vl < s) {
#pragma unroll
      for (int j = 0; j < 4; ++j) sh[j][t] = wf_merge(sh[j][t], sh[j][t + s * QCB]);
    }
    __syncthreads();
  }
  if (vl == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* p = partial + (((long)blockIdx.x) * 4 * QCB + c + j) * 3;
      p[0] = sh[j][t].n;
      p[1] = sh[j][t].mean;
      p[2] = sh[j][t].m2;
    }
  }
}

// One block per channel: 64 lanes stride over the nb block partials (Chan merge in double),
// then a fixed-order tree over the 64 lane results -> deterministic and ~nb/64 serial steps.
// fin.scale != NULL: the per-channel finalisation of msk_bn_finalize(world = 1) runs here as well (one launch less per
// BatchNorm layer): the same arithmetic on the same float-rounded (mean, M2) record, so the results are bitwise those of
// the two-kernel form.
constexpr int kMergeThreads = 256, kMergeThreadsMax = 1024;
// block size by record count: 256 threads up to 4096 records; 1024 beyond (the 8192 records of up_tr32's up-convolution were a
// chain of 32 dependent (load, divide, update) steps per thread: 25 us for 16 channels)
static inline int merge_threads(int nb) { return nb > 4096 ? kMergeThreadsMax : kMergeThreads; }
__global__ void __launch_bounds__(kMergeThreadsMax)
bn_stats_merge(const float* __restrict__ partial, int nb, int C, int CB, float* __restrict__ stats /*[2C]*/, msk_bn_fin fin) {
  // a thread's chain of dependent (divide, update) steps is nb / blockDim long; the records of eight steps are requested
  // together (round 5: one load latency per eight steps instead of one per step -- 17.8 -> 12.6 us for the 4096 tile records of a
  // 128^3 layer, 25.5 -> 18.2 us for the 8192 of up_tr32's up-convolution with 1024 threads; what remains is the fp64 divide chain)
  __shared__ double sn[kMergeThreadsMax], sm[kMergeThreadsMax], s2[kMergeThreadsMax];
  const int c = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
  const int cb = c / CB, cl = c % CB;
  double n = 0, mean = 0, m2 = 0;
  constexpr int PF = 8;
  for (int b0 = t; b0 < nb; b0 += PF * nt) {
    float r[PF][3];
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const int b = b0 + k * nt;
      if (b < nb) {
        const float* p = partial + (((long)cb * nb + b) * CB + cl) * 3;
        r[k][0] = p[0]; r[k][1] = p[1]; r[k][2] = p[2];
      } else {
        r[k][0] = 0.f; r[k][1] = 0.f; r[k][2] = 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const double bn = r[k][0];
      if (bn == 0) continue;
      const double d = (double)r[k][1] - mean, tot = n + bn, f = bn / tot;
      m2 = m2 + (double)r[k][2] + d * d * n * f;
      mean = mean + d * f;
      n = tot;
    }
  }
  sn[t] = n; sm[t] = mean; s2[t] = m2;
  __syncthreads();
  for (int s = nt / 2; s > 0; s >>= 1) {
    if (t < s && sn[t + s] > 0) {
      const double bn = sn[t + s], d = sm[t + s] - sm[t], tot = sn[t] + bn, f = bn / tot;
      s2[t] = s2[t] + s2[t + s] + d * d * sn[t] * f;
      sm[t] = sm[t] + d * f;
      sn[t] = tot;
    }
    __syncthreads();
  }
  if (t == 0) {
    const float fm = (float)sm[0], fm2 = (float)s2[0];
    stats[c] = fm;
    stats[C + c] = fm2;
    if (fin.scale) {
      const double mu = (double)fm, var = (double)fm2 / fin.count;  // biased (paddle BatchNorm training)
      const double invstd = 1.0 / sqrt(var + (double)fin.eps);
      const float g = fin.gamma ? fin.gamma[c] : 1.f, b = fin.beta ? fin.beta[c] : 0.f;
      fin.save_mean[c] = (float)mu;
      fin.save_invstd[c] = (float)invstd;
      fin.scale[c] = (float)(g * invstd);
      fin.shift[c] = (float)(b - mu * g * invstd);
      if (fin.running_mean) fin.running_mean[c] = fin.momentum * fin.running_mean[c] + (1.f - fin.momentum) * (float)mu;
      if (fin.running_var) fin.running_var[c] = fin.momentum * fin.running_var[c] + (1.f - fin.momentum) * (float)var;
    }
  }
}

__global__ void bn_finalize_k(const float* __restrict__ gathered, int world, double cnt, int C,
                              const float* __restrict__ gamma, const float* __restrict__ beta,
                              float eps, float momentum, float* running_mean, float* running_var,
                              float* save_mean, float* save_invstd, float* scale, float* shift) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double n = 0, mean = 0, m2 = 0;
  for (int r = 0; r < world; ++r) {
    double bm = gathered[(long)r * 2 * C + c], bm2 = gathered[(long)r * 2 * C + C + c];
    double d = bm - mean, tot = n + cnt, f = cnt / tot;
    m2 = m2 + bm2 + d * d * n * f;
    mean = mean + d * f;
    n = tot;
  }
  double var = m2 / n;  // biased (paddle BatchNorm training)
  double invstd = 1.0 / sqrt(var + (double)eps);
  float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  save_mean[c] = (float)mean;
  save_invstd[c] = (float)invstd;
  float sc = (float)(g * invstd);
  scale[c] = sc;
  shift[c] = (float)(b - mean * g * invstd);
  if (running_mean) running_mean[c] = momentum * running_mean[c] + (1.f - momentum) * (float)mean;
  if (running_var) running_var[c] = momentum * running_var[c] + (1.f - momentum) * (float)var;
}

__global__ void bn_eval_coeffs_k(int C, const float* gamma, const float* beta, const float* rm,
                                 const float* rv, float eps, float* save_mean, float* save_invstd,
                                 float* scale, float* shift) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float invstd = 1.f / sqrtf(rv[c] + eps);
  float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  save_mean[c] = rm[c];
  save_invstd[c] = invstd;
  scale[c] = g * invstd;
  shift[c] = b - rm[c] * g * invstd;
}

// ---------------------------------------------------------------------------
// fused affine + residual + PReLU, forward
// ---------------------------------------------------------------------------
template <int V>
struct VecT;
template <>
struct VecT<1> { using T = float; };
template <>
struct VecT<4> { using T = float4; };

template <int V>
__device__ __forceinline__ void ldv(const float* p, float (&o)[V]) {
  if constexpr (V == 4) {
    float4 t = *reinterpret_cast<const float4*>(p);
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
  } else {
    o[0] = *p;
  }
}
template <int V>
__device__ __forceinline__ void stv(float* p, const float (&o)[V]) {
  if constexpr (V == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  } else {
    *p = o[0];
  }
}

// element index -> (voxel, channel group): shift/mask when the group count is a power of two (every VNet
// layer), a 64-bit division per element otherwise -- the division was a visible cost in these HBM-bound loops
__device__ __forceinline__ void split_vc(long i, int cv, int cshift, long& v, int& cg) {
  if (cshift >= 0) {
    v = i >> cshift;
    cg = (int)(i & (long)(cv - 1));
  } else {
    v = i / cv;
    cg = (int)(i - v * cv);
  }
}

// max of a block folded into an amax array (msk_wbf.h: kWbfAmaxWays floats, bits of non-negative floats): ONE atomic per
// block, on the way blockIdx % ways, nothing waits for it (per-wavefront atomics on a single address cost 0.2 ms per step)
__device__ __forceinline__ void block_atomic_max(unsigned* amax, float m) {
  __shared__ float shm_amax[kThreads / 64];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __syncthreads();  // a second call reuses the array
  if ((threadIdx.x & 63) == 0) shm_amax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, shm_amax[i]);
    if (m > 0.f) (void)atomicMax(amax + (blockIdx.x + blockIdx.y * 7u) % kWbfAmaxWays, __float_as_uint(m));
  }
}

template <int V>
__global__ void __launch_bounds__(kThreads)
affine_act_fwd_k(const float* __restrict__ x, int ldx, const float* __restrict__ scale,
                 const float* __restrict__ shift, const float* __restrict__ res, int ldr, int cres,
                 const float* __restrict__ alpha, float* __restrict__ out, int ldo, long voxels, int C,
                 const float* __restrict__ alpha_in, unsigned* __restrict__ out_amax) {
  const int cv = C / V;
  const int cshift = (cv & (cv - 1)) == 0 ? __ffs(cv) - 1 : -1;  // wave-uniform
  const long total = voxels * cv;
  float mx = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long v;
    int cg;
    split_vc(i, cv, cshift, v, cg);
    const int c = cg * V;
    float xv[V], o[V];
    ldv<V>(x + v * ldx + c, xv);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float u = xv[j];
      if (scale) u = fmaf(u, scale[c + j], shift[c + j]);
      if (alpha_in && !(u > 0.f)) u *= alpha_in[c + j];  // msk_affine_act_join_fwd: the unit's own PReLU ahead of the join
      o[j] = u;
    }
    if (res) {
      if (cres == C) {
        float rv[V];
        ldv<V>(res + v * ldr + c, rv);
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] += rv[j];
      } else {
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] += res[v * ldr + (c + j) % cres];
      }
    }
    if (alpha) {
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = o[j] > 0.f ? o[j] : alpha[c + j] * o[j];
    }
#pragma unroll
    for (int j = 0; j < V; ++j) mx = fmaxf(mx, fabsf(o[j]));
    stv<V>(out + v * ldo + c, o);
  }
  if (out_amax) block_atomic_max(out_amax, mx);  // max |out| for the consumer that scales it into fp16 range (msk_amax_new)
}

// Channel-stationary float4 variants (C/4 a power of two, i.e. every VNet trunk layer): a thread keeps
// ONE channel quad and strides over voxels, so the per-channel coefficients are loaded once into
// registers instead of 3-7 cached loads per element (the element-indexed kernels were bound by the
// texture-address path: 2.3 TB/s for bwd_apply vs 5 TB/s for a plain copy), two voxels in flight.
__global__ void EW_BOUNDS
affine_act_fwd_cs_k(const float* __restrict__ x, int ldx, const float* __restrict__ scale,
                    const float* __restrict__ shift, const float* __restrict__ res, int ldr, int cres,
                    const float* __restrict__ alpha, float* __restrict__ out, int ldo, long voxels, int C, int cshift,
                    const float* __restrict__ alpha_in, unsigned* __restrict__ out_amax, float* __restrict__ out2, int ldo2) {
  // out2 (nullable, round 5: msk_affine_act_fwd_amax2): a second copy of the result with its own voxel stride -- in_tr writes its
  // output into the skip half of up_tr32's concat buffer AND as a dense tensor for the readers of that half alone
  float mx = 0.f;
  const long g = (long)blockIdx.x * kThreads + threadIdx.x;
  const int c = (int)(g & ((C >> 2) - 1)) * 4;
  const long vstride = ((long)gridDim.x * kThreads) >> cshift;
  float sc[4], sf[4], al[4], ai[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sc[j] = scale ? scale[c + j] : 1.f;
    sf[j] = scale ? shift[c + j] : 0.f;
    al[j] = alpha ? alpha[c + j] : 1.f;
    ai[j] = alpha_in ? alpha_in[c + j] : 1.f;
  }
  auto load_res = [&](long v) {
    if (!res) return make_float4(0.f, 0.f, 0.f, 0.f);
    if (cres == C) return *reinterpret_cast<const float4*>(res + v * ldr + c);
    const float* r = res + v * ldr;
    if (cres == 1) {  // in_tr: the tiled one-channel input (vnet.py:76-78) -- one load, not four modulo-indexed ones
      const float r0 = r[0];
      return make_float4(r0, r0, r0, r0);
    }
    return make_float4(r[c % cres], r[(c + 1) % cres], r[(c + 2) % cres], r[(c + 3) % cres]);
  };
  auto body = [&](long v, const float4 xq, const float4 rq) {
    const float xv[4] = {xq.x, xq.y, xq.z, xq.w}, rv[4] = {rq.x, rq.y, rq.z, rq.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = fmaf(xv[j], sc[j], sf[j]);
      if (alpha_in && !(t > 0.f)) t *= ai[j];  // the unit's own PReLU ahead of the join (msk_affine_act_join_fwd)
      const float u = t + rv[j];
      o[j] = (alpha && !(u > 0.f)) ? al[j] * u : u;
    }
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
    *reinterpret_cast<float4*>(out + v * ldo + c) = make_float4(o[0], o[1], o[2], o[3]);
    if (out2) *reinterpret_cast<float4*>(out2 + v * ldo2 + c) = make_float4(o[0], o[1], o[2], o[3]);
  };
  long v = g >> cshift;
  for (; v + vstride < voxels; v += 2 * vstride) {
    const float4 xa = *reinterpret_cast<const float4*>(x + v * ldx + c);
    const float4 xb = *reinterpret_cast<const float4*>(x + (v + vstride) * ldx + c);
    const float4 ra = load_res(v), rb = load_res(v + vstride);
    body(v, xa, ra);
    body(v + vstride, xb, rb);
  }
  if (v < voxels) body(v, *reinterpret_cast<const float4*>(x + v * ldx + c), load_res(v));
  if (out_amax) block_atomic_max(out_amax, mx);
}

// ---------------------------------------------------------------------------
// backward pass 1: per-channel sums
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
affine_act_bwd_reduce_k(const float* __restrict__ x, int ldx, const float* __restrict__ scale,
                        const float* __restrict__ shift, const float* __restrict__ res, int ldr,
                        int cres, const float* __restrict__ alpha, const float* __restrict__ mean,
                        const float* __restrict__ invstd, const float* __restrict__ dout, int ldd,
                        long voxels, int C, int CB, int VPB, float* __restrict__ partial /*[cblocks][nb][3][CB]*/) {
  __shared__ float sh[3][kThreads];
  const int t = threadIdx.x;
  const int cl = t % CB, vl = t / CB;
  const int c = blockIdx.y * CB + cl;
  const int nb = gridDim.x;
  const long per = (voxels + nb - 1) / nb;
  const long v0 = (long)blockIdx.x * per;
  long v1 = v0 + per;
  if (v1 > voxels) v1 = voxels;
  float s_du = 0.f, s_dux = 0.f, s_da = 0.f;
  if (c < C) {
    const float sc = scale ? scale[c] : 1.f, sf = scale ? shift[c] : 0.f;
    const float al = alpha ? alpha[c] : 1.f;
    const float mu = mean ? mean[c] : 0.f, is = mean ? invstd[c] : 0.f;
    const int cr = res ? (cres == C ? c : c % cres) : 0;
    for (long v = v0 + vl; v < v1; v += VPB) {
      float xv = x[v * ldx + c];
      float u = fmaf(xv, sc, sf);
      if (res) u += res[v * ldr + cr];
      float d = dout[v * ldd + c];
      float du = d;
      if (alpha && !(u > 0.f)) {
        du = al * d;
        s_da = fmaf(d, u, s_da);
      }
      s_du += du;
      s_dux = fmaf(du, (xv - mu) * is, s_dux);
    }
  }
  sh[0][t] = s_du;
  sh[1][t] = s_dux;
  sh[2][t] = s_da;
  __syncthreads();
  for (int s = VPB >> 1; s > 0; s >>= 1) {
    if (vl < s) {
      sh[0][t] += sh[0][t + s * CB];
      sh[1][t] += sh[1][t + s * CB];
      sh[2][t] += sh[2][t + s * CB];
    }
    __syncthreads();
  }
  if (vl == 0) {
    float* p = partial + ((long)blockIdx.y * nb + blockIdx.x) * 3 * CB;
    p[cl] = sh[0][t];
    p[CB + cl] = sh[1][t];
    p[2 * CB + cl] = sh[2][t];
  }
}

// float4 variant: a thread owns one channel QUAD and every VL-th voxel (a wavefront reads 1 KiB
// contiguous), two voxels in flight per iteration.  The scalar kernel above moved 4 bytes per
// lane per load and reached ~1.2 TB/s on the 128^3 layers.
//   JOIN = false: the three BatchNorm/PReLU sums of a ConvBNAct unit
//   MODE 1 (JOIN): the residual join out = prelu(a + b) in ONE pass: writes da (and db, optionally
//                 accumulating) while reducing the alpha-gradient sum -- the data gradient of a
//                 join does not depend on any sum, so the separate reduce pass is not needed.
//   MODE 2:       the join fused with the unit in front of it (msk_add_act_join_bwd_ex): as MODE 1 with
//                 a = prelu(scale*x + shift, alpha_in), and in the same pass the unit's OWN three BatchNorm/PReLU
//                 sums (and maxima) of du = da * prelu'(.) -- the unit's reduce pass disappears.
//                 partial quantities: [0] sum du, [1] sum du*xhat, [2] d alpha_in, [3] d alpha (join)
template <int MODE>
__global__ void EW_BOUNDS
affine_act_bwd_reduce_v4_k(const float* __restrict__ x, int ldx, const float* __restrict__ scale,
                           const float* __restrict__ shift, const float* __restrict__ res, int ldr, int cres,
                           const float* __restrict__ alpha, const float* __restrict__ mean,
                           const float* __restrict__ invstd, const float* __restrict__ dout, int ldd, long voxels,
                           int C, int QCB, int VL, float* __restrict__ da, int ldda, float* __restrict__ db,
                           int lddb, int db_acc, float* __restrict__ partial /*[nb][NQ][4*QCB]*/,
                           unsigned* __restrict__ maxes /*[2] or null: max |du|, max |xhat| (bits of non-negative floats)*/,
                           const float* __restrict__ alpha_in /*JOIN: inner PReLU of the first operand, or null*/) {
  __builtin_amdgcn_s_setprio(3);  // HBM-bound pass on the critical path: issue ahead of the co-resident weight-gradient waves
  constexpr bool JOIN = MODE != 0, UNIT = MODE == 2;
  constexpr int NQ = MODE == 0 ? 3 : (MODE == 1 ? 1 : 4);
  float m_du = 0.f, m_xh = 0.f;
  __shared__ float sh[NQ * 4][kThreads];
  const int t = threadIdx.x;
  const int cq = t % QCB, vl = t / QCB;
  const int c = cq * 4;
  const int nb = gridDim.x;
  const long per = (voxels + nb - 1) / nb;
  const long v0 = (long)blockIdx.x * per;
  long v1 = v0 + per;
  if (v1 > voxels) v1 = voxels;
  float s_du[4] = {0.f, 0.f, 0.f, 0.f}, s_dux[4] = {0.f, 0.f, 0.f, 0.f}, s_da[4] = {0.f, 0.f, 0.f, 0.f};
  float s_dai[4] = {0.f, 0.f, 0.f, 0.f};  // MODE 2: gradient of the unit's own slope
  if (c < C) {
    float sc[4], sf[4], al[4], mu[4], is[4], ai[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sc[j] = scale ? scale[c + j] : 1.f;
      sf[j] = scale ? shift[c + j] : 0.f;
      al[j] = alpha ? alpha[c + j] : 1.f;
      ai[j] = (JOIN && alpha_in) ? alpha_in[c + j] : 1.f;
      mu[j] = mean ? mean[c + j] : 0.f;
      is[j] = mean ? invstd[c + j] : 0.f;
    }
    auto body = [&](long v, const float4 xq, const float4 dq, const float4 rq) {
      const float xv[4] = {xq.x, xq.y, xq.z, xq.w}, dv[4] = {dq.x, dq.y, dq.z, dq.w};
      const float rv[4] = {rq.x, rq.y, rq.z, rq.w};
      float du[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t = fmaf(xv[j], sc[j], sf[j]);
        const float tin = t;
        if (JOIN && alpha_in && !(t > 0.f)) t *= ai[j];
        const float u = t + rv[j];
        t = tin;  // pre-activation of the unit (MODE 2 below)
        float g = dv[j];
        if (alpha && !(u > 0.f)) {
          g = al[j] * dv[j];
          s_da[j] = fmaf(dv[j], u, s_da[j]);
        }
        du[j] = g;
        if (!JOIN || UNIT) {
          float gu = g;  // gradient w.r.t. the BatchNorm output of the unit
          if (UNIT && !(t > 0.f)) {
            gu = g * ai[j];
            s_dai[j] = fmaf(g, fmaf(xv[j], sc[j], sf[j]), s_dai[j]);
          }
          const float xh = (xv[j] - mu[j]) * is[j];
          s_du[j] += gu;
          s_dux[j] = fmaf(gu, xh, s_dux[j]);
          m_du = fmaxf(m_du, fabsf(gu));
          m_xh = fmaxf(m_xh, fabsf(xh));
        }
      }
      if (JOIN) {
        const float4 o = make_float4(du[0], du[1], du[2], du[3]);
        *reinterpret_cast<float4*>(da + v * ldda + c) = o;
        if (db) {   // (uniform; null: the consumer of db reads da instead)
          float4* bp = reinterpret_cast<float4*>(db + v * lddb + c);
          if (db_acc) {
            const float4 old = *bp;
            *bp = make_float4(old.x + o.x, old.y + o.y, old.z + o.z, old.w + o.w);
          } else {
            *bp = o;
          }
        }
      }
    };
    auto load_res = [&](long v) {
      if (!res) return make_float4(0.f, 0.f, 0.f, 0.f);
      if (cres == C) return *reinterpret_cast<const float4*>(res + v * ldr + c);
      const float* r = res + v * ldr;
      if (cres == 1) {  // in_tr: the tiled one-channel input (vnet.py:76-78) -- one load, not four modulo-indexed ones
        const float r0 = r[0];
        return make_float4(r0, r0, r0, r0);
      }
      return make_float4(r[c % cres], r[(c + 1) % cres], r[(c + 2) % cres], r[(c + 3) % cres]);
    };
    long v = v0 + vl;
    for (; v + VL < v1; v += 2 * VL) {
      const float4 xa = *reinterpret_cast<const float4*>(x + v * ldx + c);
      const float4 xb = *reinterpret_cast<const float4*>(x + (v + VL) * ldx + c);
      const float4 ga = *reinterpret_cast<const float4*>(dout + v * ldd + c);
      const float4 gb = *reinterpret_cast<const float4*>(dout + (v + VL) * ldd + c);
      const float4 ra = load_res(v), rb = load_res(v + VL);
      body(v, xa, ga, ra);
      body(v + VL, xb, gb, rb);
    }
    if (v < v1)
      body(v, *reinterpret_cast<const float4*>(x + v * ldx + c), *reinterpret_cast<const float4*>(dout + v * ldd + c),
           load_res(v));
  }
  if ((!JOIN || UNIT) && maxes) {  // uniform
    block_atomic_max(maxes, m_du);
    block_atomic_max(maxes + kWbfAmaxWays, m_xh);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (MODE == 1) {
      sh[j][t] = s_da[j];
    } else {
      sh[j][t] = s_du[j];
      sh[4 + j][t] = s_dux[j];
      sh[8 + j][t] = UNIT ? s_dai[j] : s_da[j];
      if (UNIT) sh[(NQ - 1) * 4 + j][t] = s_da[j];
    }
  }
  __syncthreads();
  for (int s = VL >> 1; s > 0; s >>= 1) {
    if (vl < s) {
#pragma unroll
      for (int k = 0; k < NQ * 4; ++k) sh[k][t] += sh[k][t + s * QCB];
    }
    __syncthreads();
  }
  if (vl == 0) {
    float* p = partial + (long)blockIdx.x * NQ * 4 * QCB;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) p[q * 4 * QCB + c + j] = sh[q * 4 + j][t];
  }
}

// One block per output (q, c): the lanes stride over the nb block partials in double, fixed-order tree over the lanes
// (deterministic).  64 lanes, or 256 when there are many partials (round 4: "reduce_vpl" 8 gives the small tensors up to 2048 of them).
__global__ void __launch_bounds__(256)
sums_merge_k(const float* __restrict__ partial, int nb, int C, int CB, int nq, float* __restrict__ sums /*[nq][C]*/,
             int accumulate, float* __restrict__ g0 = nullptr, float* __restrict__ g1 = nullptr, float* __restrict__ g2 = nullptr,
             float* __restrict__ g3 = nullptr) {
  // g0..g3 (nullable): quantity q of channel c is also ADDED to gq[c] -- the parameter gradients that
  // msk_affine_act_param_grads would take from sums afterwards (d beta, d gamma, d alpha; the join's d alpha)
  __shared__ double sh[256];
  const int i = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
  const int q = i / C, c = i % C;
  const int cb = c / CB, cl = c % CB;
  double s = 0;
  for (int b = t; b < nb; b += nt) s += partial[(((long)cb * nb + b) * nq + q) * CB + cl];
  sh[t] = s;
  __syncthreads();
  for (int k = nt >> 1; k > 0; k >>= 1) {
    if (t < k) sh[t] += sh[t + k];
    __syncthreads();
  }
  if (t == 0) {
    const float v = (float)sh[0];
    sums[i] = accumulate ? sums[i] + v : v;
    float* g = q == 0 ? g0 : (q == 1 ? g1 : (q == 2 ? g2 : g3));
    if (g) g[c] += v;
  }
}

// ---------------------------------------------------------------------------
// backward pass 2
// ---------------------------------------------------------------------------
template <int V>
__global__ void __launch_bounds__(kThreads)
affine_act_bwd_apply_k(const float* __restrict__ x, int ldx, const float* __restrict__ scale,
                       const float* __restrict__ shift, const float* __restrict__ res, int ldr,
                       int cres, const float* __restrict__ alpha, const float* __restrict__ mean,
                       const float* __restrict__ invstd, const float* __restrict__ dout, int ldd,
                       const float* __restrict__ sums, float invM, int bn_mode, float* __restrict__ dx,
                       int lddx, float* __restrict__ dres, int lddr, int dres_acc, long voxels, int C,
                       unsigned* __restrict__ dx_amax) {
  const int cv = C / V;
  const int cshift = (cv & (cv - 1)) == 0 ? __ffs(cv) - 1 : -1;  // wave-uniform
  const long total = voxels * cv;
  float mx = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long v;
    int cg;
    split_vc(i, cv, cshift, v, cg);
    const int c = cg * V;
    float xv[V], dv[V], du[V], o[V];
    ldv<V>(x + v * ldx + c, xv);
    ldv<V>(dout + v * ldd + c, dv);
    float rv[V];
#pragma unroll
    for (int j = 0; j < V; ++j) rv[j] = 0.f;
    if (res && alpha) {
      if (cres == C) {
        ldv<V>(res + v * ldr + c, rv);
      } else {
#pragma unroll
        for (int j = 0; j < V; ++j) rv[j] = res[v * ldr + (c + j) % cres];
      }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float d = dv[j];
      if (alpha) {
        float u = xv[j];
        if (scale) u = fmaf(u, scale[c + j], shift[c + j]);
        u += rv[j];
        if (!(u > 0.f)) d *= alpha[c + j];
      }
      du[j] = d;
      if (bn_mode == 1) {
        float xh = (xv[j] - mean[c + j]) * invstd[c + j];
        o[j] = scale[c + j] * (d - sums[c + j] * invM - xh * sums[C + c + j] * invM);
      } else if (bn_mode == 2) {
        o[j] = scale[c + j] * d;
      } else {
        o[j] = d;
      }
    }
    if (dx) stv<V>(dx + v * lddx + c, o);
#pragma unroll
    for (int j = 0; j < V; ++j) mx = fmaxf(mx, fabsf(o[j]));
    if (dres) {
      if (dres_acc) {
        float a[V];
        ldv<V>(dres + v * lddr + c, a);
#pragma unroll
        for (int j = 0; j < V; ++j) du[j] += a[j];
      }
      stv<V>(dres + v * lddr + c, du);
    }
  }
  if (dx_amax) block_atomic_max(dx_amax, mx);   // max |dx| for the gradient kernels that scale it into fp16 range (msk_amax_new)
}

__global__ void EW_BOUNDS
affine_act_bwd_apply_cs_k(const float* __restrict__ x, int ldx, const float* __restrict__ scale,
                          const float* __restrict__ shift, const float* __restrict__ res, int ldr, int cres,
                          const float* __restrict__ alpha, const float* __restrict__ mean,
                          const float* __restrict__ invstd, const float* __restrict__ dout, int ldd,
                          const float* __restrict__ sums, float invM, int bn_mode, float* __restrict__ dx, int lddx,
                          float* __restrict__ dres, int lddr, int dres_acc, long voxels, int C, int cshift,
                          unsigned* __restrict__ dx_amax) {
  __builtin_amdgcn_s_setprio(3);  // HBM-bound pass on the critical path: issue ahead of the co-resident weight-gradient waves
  const long g = (long)blockIdx.x * kThreads + threadIdx.x;
  const int c = (int)(g & ((C >> 2) - 1)) * 4;
  const long vstride = ((long)gridDim.x * kThreads) >> cshift;
  float sc[4], sf[4], al[4], mu[4], is[4], s1[4], s2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sc[j] = scale ? scale[c + j] : 1.f;
    sf[j] = scale ? shift[c + j] : 0.f;
    al[j] = alpha ? alpha[c + j] : 1.f;
    mu[j] = bn_mode == 1 ? mean[c + j] : 0.f;
    is[j] = bn_mode == 1 ? invstd[c + j] : 0.f;
    s1[j] = bn_mode == 1 ? sums[c + j] * invM : 0.f;
    s2[j] = bn_mode == 1 ? sums[C + c + j] * invM : 0.f;
  }
  const bool need_res = res && alpha;
  float mx = 0.f;
  auto load_res = [&](long v) {
    if (!need_res) return make_float4(0.f, 0.f, 0.f, 0.f);
    if (cres == C) return *reinterpret_cast<const float4*>(res + v * ldr + c);
    const float* r = res + v * ldr;
    if (cres == 1) {  // in_tr: the tiled one-channel input (vnet.py:76-78) -- one load, not four modulo-indexed ones
      const float r0 = r[0];
      return make_float4(r0, r0, r0, r0);
    }
    return make_float4(r[c % cres], r[(c + 1) % cres], r[(c + 2) % cres], r[(c + 3) % cres]);
  };
  auto body = [&](long v, const float4 xq, const float4 dq, const float4 rq) {
    const float xv[4] = {xq.x, xq.y, xq.z, xq.w}, dv[4] = {dq.x, dq.y, dq.z, dq.w}, rv[4] = {rq.x, rq.y, rq.z, rq.w};
    float du[4], o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float d = dv[j];
      if (alpha) {
        const float u = fmaf(xv[j], sc[j], sf[j]) + rv[j];
        if (!(u > 0.f)) d *= al[j];
      }
      du[j] = d;
      if (bn_mode == 1) {
        const float xh = (xv[j] - mu[j]) * is[j];
        o[j] = sc[j] * (d - s1[j] - xh * s2[j]);
      } else if (bn_mode == 2) {
        o[j] = sc[j] * d;
      } else {
        o[j] = d;
      }
    }
    if (dx) *reinterpret_cast<float4*>(dx + v * lddx + c) = make_float4(o[0], o[1], o[2], o[3]);
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
    if (dres) {
      float4* rp = reinterpret_cast<float4*>(dres + v * lddr + c);
      float4 w = make_float4(du[0], du[1], du[2], du[3]);
      if (dres_acc) {
        const float4 old = *rp;
        w = make_float4(w.x + old.x, w.y + old.y, w.z + old.z, w.w + old.w);
      }
      *rp = w;
    }
  };
  long v = g >> cshift;
  for (; v + vstride < voxels; v += 2 * vstride) {
    const float4 xa = *reinterpret_cast<const float4*>(x + v * ldx + c);
    const float4 xb = *reinterpret_cast<const float4*>(x + (v + vstride) * ldx + c);
    const float4 ga = *reinterpret_cast<const float4*>(dout + v * ldd + c);
    const float4 gb = *reinterpret_cast<const float4*>(dout + (v + vstride) * ldd + c);
    const float4 ra = load_res(v), rb = load_res(v + vstride);
    body(v, xa, ga, ra);
    body(v + vstride, xb, gb, rb);
  }
  if (v < voxels)
    body(v, *reinterpret_cast<const float4*>(x + v * ldx + c), *reinterpret_cast<const float4*>(dout + v * ldd + c),
         load_res(v));
  if (dx_amax) block_atomic_max(dx_amax, mx);
}

__global__ void param_grads_k(int C, const float* sums, float* dgamma, float* dbeta, float* dalpha, int acc) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (dbeta) dbeta[c] = (acc ? dbeta[c] : 0.f) + sums[c];
  if (dgamma) dgamma[c] = (acc ? dgamma[c] : 0.f) + sums[C + c];
  if (dalpha) dalpha[c] = (acc ? dalpha[c] : 0.f) + sums[2 * C + c];
}

__global__ void bias_grad_from_sums_k(int C, const float* sums, const float* scale, float* dbias, int acc) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) dbias[c] = (acc ? dbias[c] : 0.f) + scale[c] * sums[c];
}

// ---------------------------------------------------------------------------
// copy with per-(n,c) scale
// ---------------------------------------------------------------------------
template <int V>
__global__ void __launch_bounds__(kThreads)
copy_scale_k(const float* __restrict__ src, int lds_, const float* __restrict__ mask, float* __restrict__ dst,
             int ldd, long voxels, long vox_per_n, int C, int acc, unsigned* __restrict__ dst_amax) {
  const int cv = C / V;
  const int cshift = (cv & (cv - 1)) == 0 ? __ffs(cv) - 1 : -1;  // wave-uniform
  const long total = voxels * cv;
  float mx = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long v;
    int cg;
    split_vc(i, cv, cshift, v, cg);
    const int c = cg * V;
    float s[V];
    ldv<V>(src + v * lds_ + c, s);
    if (mask) {
      const long n = v / vox_per_n;
#pragma unroll
      for (int j = 0; j < V; ++j) s[j] *= mask[n * C + c + j];
    }
    if (acc) {
      float a[V];
      ldv<V>(dst + v * ldd + c, a);
#pragma unroll
      for (int j = 0; j < V; ++j) s[j] += a[j];
    }
#pragma unroll
    for (int j = 0; j < V; ++j) mx = fmaxf(mx, fabsf(s[j]));
    stv<V>(dst + v * ldd + c, s);
  }
  if (dst_amax) block_atomic_max(dst_amax, mx);
}

__global__ void __launch_bounds__(kThreads)
channel_sum_partial_k(const float* __restrict__ x, int ldx, long voxels, int C, int CB, int VPB,
                      float* __restrict__ partial /*[cblocks][nb][1][CB]*/) {
  __shared__ float sh[kThreads];
  const int t = threadIdx.x;
  const int cl = t % CB, vl = t / CB;
  const int c = blockIdx.y * CB + cl;
  const int nb = gridDim.x;
  const long per = (voxels + nb - 1) / nb;
  const long v0 = (long)blockIdx.x * per;
  long v1 = v0 + per;
  if (v1 > voxels) v1 = voxels;
  float s = 0.f;
  if (c < C)
    for (long v = v0 + vl; v < v1; v += VPB) s += x[v * ldx + c];
  sh[t] = s;
  __syncthreads();
  for (int k = VPB >> 1; k > 0; k >>= 1) {
    if (vl < k) sh[t] += sh[t + k * CB];
    __syncthreads();
  }
  if (vl == 0) partial[((long)blockIdx.y * nb + blockIdx.x) * CB + cl] = sh[t];
}

// splitmix64-based counter RNG for Dropout3D masks
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void dropout_mask_k(uint64_t seed, uint64_t step, uint32_t site, int count, float p, float* mask) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  uint64_t h = splitmix64(seed ^ splitmix64(step * 0x100000001B3ull + site));
  h = splitmix64(h + (uint64_t)i);
  float u = (float)(h >> 40) * (1.0f / 16777216.0f);  // [0,1)
  mask[i] = u >= p ? 1.0f / (1.0f - p) : 0.f;
}

__global__ void argmax_k(const float* __restrict__ x, int ld, long voxels, int C, int32_t* __restrict__ out) {
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < voxels; v += (long)gridDim.x * blockDim.x) {
    const float* p = x + v * ld;
    float best = p[0];
    int bi = 0;
    for (int c = 1; c < C; ++c) {
      float q = p[c];
      if (q > best) {  // first maximum wins (numpy/paddle argmax)
        best = q;
        bi = c;
      }
    }
    out[v] = bi;
  }
}

// softmax over the channel axis of every voxel (F.softmax(logits, axis=1) of the reference's AUC path, core/val.py:121-123)
__global__ void softmax_c_k(const float* __restrict__ x, int ld, long voxels, int C, float* __restrict__ out, int old) {
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < voxels; v += (long)gridDim.x * blockDim.x) {
    const float* p = x + v * ld;
    float m = p[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, p[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(p[c] - m);
    const float inv = 1.f / s;
    float* o = out + v * old;
    for (int c = 0; c < C; ++c) o[c] = expf(p[c] - m) * inv;
  }
}

// NCDHW <-> NDHWC through a 32x32 LDS tile (voxel x channel)
__global__ void __launch_bounds__(256)
ncdhw_to_ndhwc_k(const float* __restrict__ src, float* __restrict__ dst, int ld, long V, int C) {
  __shared__ float tile[32][33];
  const long n = blockIdx.z;
  const long v0 = (long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x % 32, ty = threadIdx.x / 32;  // ty in [0,8)
  for (int r = ty; r < 32; r += 8) {  // r = channel, tx = voxel
    int c = c0 + r;
    long v = v0 + tx;
    tile[r][tx] = (c < C && v < V) ? src[(n * C + c) * V + v] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {  // r = voxel, tx = channel
    long v = v0 + r;
    int c = c0 + tx;
    if (c < C && v < V) dst[(n * V + v) * ld + c] = tile[tx][r];
  }
}
__global__ void __launch_bounds__(256)
ndhwc_to_ncdhw_k(const float* __restrict__ src, int ld, float* __restrict__ dst, long V, int C) {
  __shared__ float tile[32][33];
  const long n = blockIdx.z;
  const long v0 = (long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x % 32, ty = threadIdx.x / 32;
  for (int r = ty; r < 32; r += 8) {  // r = voxel, tx = channel
    long v = v0 + r;
    int c = c0 + tx;
    tile[r][tx] = (c < C && v < V) ? src[(n * V + v) * ld + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {  // r = channel, tx = voxel
    int c = c0 + r;
    long v = v0 + tx;
    if (c < C && v < V) dst[(n * C + c) * V + v] = tile[tx][r];
  }
}

// ---------------------------------------------------------------------------
// Round 5: DENSE tensors with 1, 2, 3 or 6 channels (the ncls = 3 tensors of out_tr, vnet.py:159-175: its BatchNorm / PReLU and
// their adjoints, the bias gradient of the 1x1x1 head).  The scalar kernels above moved 4 bytes per lane per load and ran at
// 1.5-2.3 TB/s on these 50 MB tensors (213 us per step in five passes).  12 consecutive floats of a dense tensor are 12 / C whole
// voxels, so a thread that owns float4 triples sees a FIXED channel per slot (slot % C): 16-byte accesses, coefficients in
// registers, no index arithmetic.  Eligible: ld == C, 12 % C == 0, voxels * C % 12 == 0, 16-byte aligned.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void ld12(const float* __restrict__ p, float (&v)[12]) {
  const float4* q = reinterpret_cast<const float4*>(p);
  const float4 a = q[0], b = q[1], c = q[2];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
}
__device__ __forceinline__ void st12(float* __restrict__ p, const float (&v)[12]) {
  float4* q = reinterpret_cast<float4*>(p);
  q[0] = make_float4(v[0], v[1], v[2], v[3]);
  q[1] = make_float4(v[4], v[5], v[6], v[7]);
  q[2] = make_float4(v[8], v[9], v[10], v[11]);
}

template <int C>
__global__ void __launch_bounds__(kThreads)
affine_act_fwd_d12_k(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                     const float* __restrict__ res, const float* __restrict__ alpha, float* __restrict__ out, long groups,
                     unsigned* __restrict__ out_amax) {
  float sc[C], sf[C], al[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    sc[c] = scale ? scale[c] : 1.f;
    sf[c] = scale ? shift[c] : 0.f;
    al[c] = alpha ? alpha[c] : 1.f;
  }
  float mx = 0.f;
  for (long g = (long)blockIdx.x * kThreads + threadIdx.x; g < groups; g += (long)gridDim.x * kThreads) {
    float v[12], r[12], o[12];
    ld12(x + 12 * g, v);
    if (res) ld12(res + 12 * g, r);
#pragma unroll
    for (int s = 0; s < 12; ++s) {
      float u = scale ? fmaf(v[s], sc[s % C], sf[s % C]) : v[s];
      if (res) u += r[s];
      o[s] = (alpha && !(u > 0.f)) ? al[s % C] * u : u;
      mx = fmaxf(mx, fabsf(o[s]));
    }
    st12(out + 12 * g, o);
  }
  if (out_amax) block_atomic_max(out_amax, mx);
}

// BatchNorm statistics: per-slot shifted sums, slots of a channel and then the block's threads merged (Chan) -> one record
// per block in bn_stats_partial's layout [nb][CB][3]
template <int C>
__global__ void __launch_bounds__(kThreads)
bn_stats_partial_d12_k(const float* __restrict__ x, long groups, int CB, float* __restrict__ partial) {
  __shared__ WF sh[C][kThreads];
  const int t = threadIdx.x, nb = gridDim.x;
  const long per = (groups + nb - 1) / nb;
  const long g0 = (long)blockIdx.x * per;
  long g1 = g0 + per;
  if (g1 > groups) g1 = groups;
  float K[12], s1[12], s2[12];
  float n = 0.f;
#pragma unroll
  for (int s = 0; s < 12; ++s) { K[s] = 0.f; s1[s] = 0.f; s2[s] = 0.f; }
  if (g0 + t < g1) ld12(x + 12 * (g0 + t), K);
  for (long g = g0 + t; g < g1; g += kThreads) {
    float v[12];
    ld12(x + 12 * g, v);
#pragma unroll
    for (int s = 0; s < 12; ++s) {
      const float d = v[s] - K[s];
      s1[s] += d;
      s2[s] = fmaf(d, d, s2[s]);
    }
    n += 1.f;
  }
#pragma unroll
  for (int c = 0; c < C; ++c) {
    WF w = {0.f, 0.f, 0.f};
    if (n > 0.f) {
#pragma unroll
      for (int s = c; s < 12; s += C) {   // the slots of channel c, in slot order
        WF e;
        e.n = n;
        e.mean = K[s] + s1[s] / n;
        e.m2 = fmaxf(s2[s] - s1[s] * s1[s] / n, 0.f);
        w = wf_merge(w, e);
      }
    }
    sh[c][t] = w;
  }
  __syncthreads();
  for (int k = kThreads >> 1; k > 0; k >>= 1) {
    if (t < k) {
#pragma unroll
      for (int c = 0; c < C; ++c) sh[c][t] = wf_merge(sh[c][t], sh[c][t + k]);
    }
    __syncthreads();
  }
  if (t < C) {
    float* p = partial + ((long)blockIdx.x * CB + t) * 3;
    p[0] = sh[t][0].n;
    p[1] = sh[t][0].mean;
    p[2] = sh[t][0].m2;
  }
}

// backward pass 1 (affine_act_bwd_reduce_k's sums, same partial layout [nb][3][CB]); NQ == 1: plain channel sums of x
// (channel_sum_partial_k's layout [nb][1][CB])
template <int C, int NQ>
__global__ void __launch_bounds__(kThreads)
affine_act_bwd_reduce_d12_k(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                            const float* __restrict__ res, const float* __restrict__ alpha, const float* __restrict__ mean,
                            const float* __restrict__ invstd, const float* __restrict__ dout, long groups, int CB,
                            float* __restrict__ partial) {
  __shared__ float sh[NQ * C][kThreads];
  const int t = threadIdx.x, nb = gridDim.x;
  const long per = (groups + nb - 1) / nb;
  const long g0 = (long)blockIdx.x * per;
  long g1 = g0 + per;
  if (g1 > groups) g1 = groups;
  float sc[C], sf[C], al[C], mu[C], is[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    sc[c] = (NQ == 3 && scale) ? scale[c] : 1.f;
    sf[c] = (NQ == 3 && scale) ? shift[c] : 0.f;
    al[c] = (NQ == 3 && alpha) ? alpha[c] : 1.f;
    mu[c] = (NQ == 3 && mean) ? mean[c] : 0.f;
    is[c] = (NQ == 3 && mean) ? invstd[c] : 0.f;
  }
  float a_du[12], a_dux[12], a_da[12];
#pragma unroll
  for (int s = 0; s < 12; ++s) { a_du[s] = 0.f; a_dux[s] = 0.f; a_da[s] = 0.f; }
  for (long g = g0 + t; g < g1; g += kThreads) {
    float v[12];
    ld12(x + 12 * g, v);
    if (NQ == 1) {
#pragma unroll
      for (int s = 0; s < 12; ++s) a_du[s] += v[s];
    } else {
      float d[12], r[12];
      ld12(dout + 12 * g, d);
      if (res) ld12(res + 12 * g, r);
#pragma unroll
      for (int s = 0; s < 12; ++s) {
        float u = fmaf(v[s], sc[s % C], sf[s % C]);
        if (res) u += r[s];
        float du = d[s];
        if (alpha && !(u > 0.f)) {
          du = al[s % C] * d[s];
          a_da[s] = fmaf(d[s], u, a_da[s]);
        }
        a_du[s] += du;
        a_dux[s] = fmaf(du, (v[s] - mu[s % C]) * is[s % C], a_dux[s]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < C; ++c) {
    float q0 = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll
    for (int s = c; s < 12; s += C) { q0 += a_du[s]; q1 += a_dux[s]; q2 += a_da[s]; }
    sh[c][t] = q0;
    if (NQ == 3) { sh[C + c][t] = q1; sh[2 * C + c][t] = q2; }
  }
  __syncthreads();
  for (int k = kThreads >> 1; k > 0; k >>= 1) {
    if (t < k) {
#pragma unroll
      for (int j = 0; j < NQ * C; ++j) sh[j][t] += sh[j][t + k];
    }
    __syncthreads();
  }
  if (t < NQ * C) {
    const int q = t / C, c = t % C;
    partial[((long)blockIdx.x * NQ + q) * CB + c] = sh[t][0];
  }
}

template <int C>
__global__ void __launch_bounds__(kThreads)
affine_act_bwd_apply_d12_k(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                           const float* __restrict__ res, const float* __restrict__ alpha, const float* __restrict__ mean,
                           const float* __restrict__ invstd, const float* __restrict__ dout, const float* __restrict__ sums,
                           float invM, int bn_mode, float* __restrict__ dx, float* __restrict__ dres, int dres_acc, long groups,
                           int Ctot, unsigned* __restrict__ dx_amax) {
  float sc[C], sf[C], al[C], mu[C], is[C], s1[C], s2[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    sc[c] = scale ? scale[c] : 1.f;
    sf[c] = scale ? shift[c] : 0.f;
    al[c] = alpha ? alpha[c] : 1.f;
    mu[c] = bn_mode == 1 ? mean[c] : 0.f;
    is[c] = bn_mode == 1 ? invstd[c] : 0.f;
    s1[c] = bn_mode == 1 ? sums[c] * invM : 0.f;
    s2[c] = bn_mode == 1 ? sums[Ctot + c] * invM : 0.f;
  }
  float mx = 0.f;
  for (long g = (long)blockIdx.x * kThreads + threadIdx.x; g < groups; g += (long)gridDim.x * kThreads) {
    float v[12], d[12], r[12], o[12], du[12];
    ld12(x + 12 * g, v);
    ld12(dout + 12 * g, d);
    const bool need_res = res && alpha;
    if (need_res) ld12(res + 12 * g, r);
#pragma unroll
    for (int s = 0; s < 12; ++s) {
      float t = d[s];
      if (alpha) {
        float u = scale ? fmaf(v[s], sc[s % C], sf[s % C]) : v[s];
        if (need_res) u += r[s];
        if (!(u > 0.f)) t *= al[s % C];
      }
      du[s] = t;
      if (bn_mode == 1) {
        const float xh = (v[s] - mu[s % C]) * is[s % C];
        o[s] = sc[s % C] * (t - s1[s % C] - xh * s2[s % C]);
      } else if (bn_mode == 2) {
        o[s] = sc[s % C] * t;
      } else {
        o[s] = t;
      }
      mx = fmaxf(mx, fabsf(o[s]));
    }
    if (dx) st12(dx + 12 * g, o);
    if (dres) {
      if (dres_acc) {
        float a[12];
        ld12(dres + 12 * g, a);
#pragma unroll
        for (int s = 0; s < 12; ++s) du[s] += a[s];
      }
      st12(dres + 12 * g, du);
    }
  }
  if (dx_amax) block_atomic_max(dx_amax, mx);
}

// is every (non-null) tensor of a pass dense, 16-byte aligned, with a channel count that divides 12?
inline bool d12_ok(const msk_tensor& t) { return t.p == nullptr || (t.ld == t.c && (((uintptr_t)t.p) % 16 == 0)); }
inline bool d12_shape(const msk_tensor& x, long voxels) {
  return (x.c == 1 || x.c == 2 || x.c == 3 || x.c == 6) && (voxels * x.c) % 12 == 0;
}
int g_dense12 = 1;   // option "dense12": 0 = the scalar kernels (A/B)

inline bool vec4_ok(const msk_tensor& t) {
  return t.p == nullptr || ((t.c % 4 == 0) && (t.ld % 4 == 0) && (((uintptr_t)t.p) % 16 == 0));
}
inline int ew_blocks(long total, int num_cu) {
  long b = (total + kThreads - 1) / kThreads;
  long cap = (long)num_cu * g_ew_cap;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}
inline bool same_shape(const msk_tensor& a, const msk_tensor& b) {
  return a.n == b.n && a.d == b.d && a.h == b.h && a.w == b.w && a.c == b.c;
}

}  // namespace

extern "C" {

int msk_ncdhw_to_ndhwc(msk_ctx* ctx, const float* src, msk_tensor dst) {
  long V = (long)dst.d * dst.h * dst.w;
  msk_launch_scope ls(ctx, "layout_ncdhw_to_ndhwc");
  dim3 grid(msk_cdiv(V, 32), msk_cdiv(dst.c, 32), dst.n);
  hipLaunchKernelGGL(ncdhw_to_ndhwc_k, grid, dim3(256), 0, ctx->stream, src, (float*)dst.p, dst.ld, V, dst.c);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}
int msk_ndhwc_to_ncdhw(msk_ctx* ctx, msk_tensor src, float* dst) {
  long V = (long)src.d * src.h * src.w;
  msk_launch_scope ls(ctx, "layout_ndhwc_to_ncdhw");
  dim3 grid(msk_cdiv(V, 32), msk_cdiv(src.c, 32), src.n);
  hipLaunchKernelGGL(ndhwc_to_ncdhw_k, grid, dim3(256), 0, ctx->stream, (const float*)src.p, src.ld, dst, V, src.c);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

}  // extern "C"

namespace {
__global__ void noop_k() {}
inline void noop_launches(msk_ctx* ctx) {
  for (int i = 0; i < ctx->noop_after_merge; ++i) hipLaunchKernelGGL(noop_k, dim3(1), dim3(64), 0, ctx->stream);
}
}  // namespace

int msk_bn_stats_merge(msk_ctx* ctx, const float* partial, int nb, int C, float* stats, const msk_bn_fin* fin) {
  msk_launch_scope ls(ctx, "bn_stats_merge");
  hipLaunchKernelGGL(bn_stats_merge, dim3(C), dim3(merge_threads(nb)), 0, ctx->stream, partial, nb, C, C, stats, fin ? *fin : msk_bn_fin{});
  MSK_LAUNCH_CHECK(ctx);
  noop_launches(ctx);
  return 0;
}

extern "C" {

int msk_bn_stats(msk_ctx* ctx, msk_tensor x, float* stats_local) { return msk_bn_stats_fin(ctx, x, stats_local, nullptr); }

int msk_bn_stats_fin(msk_ctx* ctx, msk_tensor x, float* stats_local, const msk_bn_fin* fin) {
  const long voxels = msk_voxels(x);
  MSK_REQUIRE(ctx, voxels > 0 && x.c > 0, "empty tensor");
  if (x.c % 4 == 0 && x.c / 4 <= kThreads && vec4_ok(x)) {
    const int QCB = pow2ceil(x.c / 4), VL = kThreads / QCB;
    const int nb = reduce_blocks(voxels, VL, ctx->num_cu, 0);
    float* partial = (float*)msk_workspace(ctx, (size_t)nb * 4 * QCB * 3 * sizeof(float));
    if (!partial) return -1;
    {
      msk_launch_scope ls(ctx, "bn_stats_partial");
      hipLaunchKernelGGL(bn_stats_partial_v4, dim3(nb), dim3(kThreads), 0, ctx->stream, (const float*)x.p, voxels, x.c,
                         x.ld, QCB, VL, partial);
      MSK_LAUNCH_CHECK(ctx);
    }
    msk_launch_scope ls(ctx, "bn_stats_merge");
    hipLaunchKernelGGL(bn_stats_merge, dim3(x.c), dim3(merge_threads(nb)), 0, ctx->stream, partial, nb, x.c, 4 * QCB, stats_local,
                       fin ? *fin : msk_bn_fin{});
    MSK_LAUNCH_CHECK(ctx);
    noop_launches(ctx);
    return 0;
  }
  ChanGeom g = chan_geom(x.c);
  int nb = reduce_blocks(voxels, g.VPB, ctx->num_cu, 0);
  size_t bytes = (size_t)g.cblocks * nb * g.CB * 3 * sizeof(float);
  float* partial = (float*)msk_workspace(ctx, bytes);
  if (!partial) return -1;
  if (g_dense12 && d12_shape(x, voxels) && d12_ok(x)) {
    const long groups = voxels * x.c / 12;
    {
      msk_launch_scope ls(ctx, "bn_stats_partial");
#define D12_ST(C_) hipLaunchKernelGGL(bn_stats_partial_d12_k<C_>, dim3(nb), dim3(kThreads), 0, ctx->stream, (const float*)x.p, groups, g.CB, partial)
      if (x.c == 1) D12_ST(1); else if (x.c == 2) D12_ST(2); else if (x.c == 3) D12_ST(3); else D12_ST(6);
#undef D12_ST
      MSK_LAUNCH_CHECK(ctx);
    }
    msk_launch_scope ls(ctx, "bn_stats_merge");
    hipLaunchKernelGGL(bn_stats_merge, dim3(x.c), dim3(merge_threads(nb)), 0, ctx->stream, partial, nb, x.c, g.CB, stats_local,
                       fin ? *fin : msk_bn_fin{});
    MSK_LAUNCH_CHECK(ctx);
    noop_launches(ctx);
    return 0;
  }
  {
    msk_launch_scope ls(ctx, "bn_stats_partial");
    hipLaunchKernelGGL(bn_stats_partial, dim3(nb, g.cblocks), dim3(kThreads), 0, ctx->stream,
                       (const float*)x.p, voxels, x.c, x.ld, g.CB, g.VPB, partial);
    MSK_LAUNCH_CHECK(ctx);
  }
  {
    msk_launch_scope ls(ctx, "bn_stats_merge");
    hipLaunchKernelGGL(bn_stats_merge, dim3(x.c), dim3(merge_threads(nb)), 0, ctx->stream, partial, nb, x.c, g.CB, stats_local,
                       fin ? *fin : msk_bn_fin{});
    MSK_LAUNCH_CHECK(ctx);
    noop_launches(ctx);
  }
  return 0;
}

int msk_bn_finalize(msk_ctx* ctx, const float* gathered, int world, double count_per_rank, int C,
                    const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                    float* scale, float* shift) {
  msk_launch_scope ls(ctx, "bn_finalize");
  hipLaunchKernelGGL(bn_finalize_k, dim3(msk_cdiv(C, 64)), dim3(64), 0, ctx->stream, gathered, world,
                     count_per_rank, C, gamma, beta, eps, momentum, running_mean, running_var, save_mean,
                     save_invstd, scale, shift);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

int msk_bn_eval_coeffs(msk_ctx* ctx, int C, const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* save_mean, float* save_invstd,
                       float* scale, float* shift) {
  msk_launch_scope ls(ctx, "bn_eval_coeffs");
  hipLaunchKernelGGL(bn_eval_coeffs_k, dim3(msk_cdiv(C, 64)), dim3(64), 0, ctx->stream, C, gamma, beta,
                     running_mean, running_var, eps, save_mean, save_invstd, scale, shift);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

static int affine_act_fwd_impl(msk_ctx* ctx, msk_tensor x, const float* scale, const float* shift, msk_tensor res,
                               const float* alpha, msk_tensor out, const float* alpha_in, float* out_amax = nullptr,
                               msk_tensor out2 = msk_tensor{}) {
  MSK_REQUIRE(ctx, same_shape(x, out), "x/out shape mismatch");
  MSK_REQUIRE(ctx, (scale == nullptr) == (shift == nullptr), "scale and shift go together");
  if (res.p) MSK_REQUIRE(ctx, res.c > 0 && (res.c == x.c || x.c % res.c == 0), "residual channels must tile");
  const long voxels = msk_voxels(x);
  const bool v4 = vec4_ok(x) && vec4_ok(out) && (res.p == nullptr || res.c != x.c || vec4_ok(res));
  msk_launch_scope ls(ctx, "affine_act_fwd");
  const int cq = x.c / 4;
  if (g_dense12 && !v4 && !alpha_in && !out2.p && d12_shape(x, voxels) && d12_ok(x) && d12_ok(out) && (res.p == nullptr || (res.c == x.c && d12_ok(res)))) {
    const long groups = voxels * x.c / 12;
    const dim3 grid(ew_blocks(groups, ctx->num_cu));
#define D12_FWD(C_) hipLaunchKernelGGL(affine_act_fwd_d12_k<C_>, grid, dim3(kThreads), 0, ctx->stream, (const float*)x.p, scale, shift, \
                                       (const float*)res.p, alpha, (float*)out.p, groups, (unsigned*)out_amax)
    if (x.c == 1) D12_FWD(1); else if (x.c == 2) D12_FWD(2); else if (x.c == 3) D12_FWD(3); else D12_FWD(6);
#undef D12_FWD
    MSK_LAUNCH_CHECK(ctx);
    return 0;
  }
  if (v4 && cq >= 1 && (cq & (cq - 1)) == 0 && cq <= kThreads) {
    int cshift = 0;
    while ((1 << cshift) < cq) ++cshift;
    hipLaunchKernelGGL(affine_act_fwd_cs_k, dim3(ew_blocks(voxels * cq / 2, ctx->num_cu)), dim3(kThreads), 0, ctx->stream,
                       (const float*)x.p, x.ld, scale, shift, (const float*)res.p, res.ld, res.c, alpha, (float*)out.p,
                       out.ld, voxels, x.c, cshift, alpha_in, (unsigned*)out_amax, (float*)out2.p, out2.ld);
    MSK_LAUNCH_CHECK(ctx);
    return 0;
  } else if (out2.p) {
    return msk_fail(ctx, __FILE__, __LINE__, "msk_affine_act_fwd_amax2", "the second output needs the channel-stationary float4 kernel (C / 4 a power of two, aligned tensors)");
  } else if (v4) {
    hipLaunchKernelGGL(affine_act_fwd_k<4>, dim3(ew_blocks(voxels * x.c / 4, ctx->num_cu)), dim3(kThreads), 0,
                       ctx->stream, (const float*)x.p, x.ld, scale, shift, (const float*)res.p, res.ld, res.c,
                       alpha, (float*)out.p, out.ld, voxels, x.c, alpha_in, (unsigned*)out_amax);
  } else {
    hipLaunchKernelGGL(affine_act_fwd_k<1>, dim3(ew_blocks(voxels * x.c, ctx->num_cu)), dim3(kThreads), 0,
                       ctx->stream, (const float*)x.p, x.ld, scale, shift, (const float*)res.p, res.ld, res.c,
                       alpha, (float*)out.p, out.ld, voxels, x.c, alpha_in, (unsigned*)out_amax);
  }
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

float* msk_amax_new(msk_ctx* ctx, int n) { return msk_scalar_slots(ctx, n > 0 ? n : 1); }

int msk_affine_act_fwd_amax(msk_ctx* ctx, msk_tensor x, const float* scale, const float* shift, msk_tensor res,
                            const float* alpha, msk_tensor out, float* out_amax) {
  return affine_act_fwd_impl(ctx, x, scale, shift, res, alpha, out, nullptr, out_amax);
}

int msk_affine_act_fwd_amax2(msk_ctx* ctx, msk_tensor x, const float* scale, const float* shift, msk_tensor res,
                             const float* alpha, msk_tensor out, float* out_amax, msk_tensor out2) {
  if (out2.p) MSK_REQUIRE(ctx, same_shape(x, out2) && vec4_ok(out2), "out2 must have x's shape and float4 alignment");
  return affine_act_fwd_impl(ctx, x, scale, shift, res, alpha, out, nullptr, out_amax, out2);
}

int msk_affine_act_join_fwd_amax(msk_ctx* ctx, msk_tensor y, const float* scale, const float* shift, const float* alpha_inner,
                                 msk_tensor res, const float* alpha_outer, msk_tensor out, float* out_amax) {
  MSK_REQUIRE(ctx, scale && shift && alpha_inner && alpha_outer && res.p, "join of a conv -> BN -> PReLU unit with a residual");
  MSK_REQUIRE(ctx, res.c == y.c, "the residual has the unit's channel count");
  return affine_act_fwd_impl(ctx, y, scale, shift, res, alpha_outer, out, alpha_inner, out_amax);
}

int msk_affine_act_fwd(msk_ctx* ctx, msk_tensor x, const float* scale, const float* shift, msk_tensor res,
                       const float* alpha, msk_tensor out) {
  return affine_act_fwd_impl(ctx, x, scale, shift, res, alpha, out, nullptr);
}

int msk_affine_act_join_fwd(msk_ctx* ctx, msk_tensor y, const float* scale, const float* shift, const float* alpha_inner,
                            msk_tensor res, const float* alpha_outer, msk_tensor out) {
  MSK_REQUIRE(ctx, scale && shift && alpha_inner && alpha_outer && res.p, "join of a conv -> BN -> PReLU unit with a residual");
  MSK_REQUIRE(ctx, res.c == y.c, "the residual has the unit's channel count");
  return affine_act_fwd_impl(ctx, y, scale, shift, res, alpha_outer, out, alpha_inner);
}

int msk_affine_act_bwd_reduce(msk_ctx* ctx, msk_tensor x, const float* scale, const float* shift, msk_tensor res,
                              const float* alpha, const float* mean, const float* invstd, msk_tensor dout,
                              float* sums) {
  return msk_affine_act_bwd_reduce_ex(ctx, x, scale, shift, res, alpha, mean, invstd, dout, sums, nullptr);
}

int msk_affine_act_bwd_reduce_ex(msk_ctx* ctx, msk_tensor x, const float* scale, const float* shift, msk_tensor res,
                                 const float* alpha, const float* mean, const float* invstd, msk_tensor dout,
                                 float* sums, float* maxes) {
  return msk_affine_act_bwd_reduce_pg(ctx, x, scale, shift, res, alpha, mean, invstd, dout, sums, maxes, 1, nullptr, nullptr, nullptr);
}

int msk_affine_act_bwd_reduce_pg(msk_ctx* ctx, msk_tensor x, const float* scale, const float* shift, msk_tensor res,
                                 const float* alpha, const float* mean, const float* invstd, msk_tensor dout,
                                 float* sums, float* maxes, int clear_maxes, float* dgamma, float* dbeta, float* dalpha) {
  MSK_REQUIRE(ctx, same_shape(x, dout), "x/dout shape mismatch");
  const long voxels = msk_voxels(x);
  const bool v4 = x.c % 4 == 0 && x.c / 4 <= kThreads && vec4_ok(x) && vec4_ok(dout) &&
                  (res.p == nullptr || res.c != x.c || vec4_ok(res));
  if (maxes) {
    MSK_REQUIRE(ctx, v4, "maxes are produced by the float4 kernel only: channel count and strides multiples of 4, 16-byte aligned tensors");
    if (clear_maxes) MSK_CHECK_HIP(ctx, hipMemsetAsync(maxes, 0, 2 * kWbfAmaxWays * sizeof(float), ctx->stream));
  }
  if (v4) {
    const int QCB = pow2ceil(x.c / 4), VL = kThreads / QCB;
    const int nb = reduce_blocks(voxels, VL, ctx->num_cu, 1);
    float* partial = (float*)msk_workspace(ctx, (size_t)nb * 3 * 4 * QCB * sizeof(float));
    if (!partial) return -1;
    {
      msk_launch_scope ls(ctx, "affine_act_bwd_reduce");
      hipLaunchKernelGGL(affine_act_bwd_reduce_v4_k<0>, dim3(nb), dim3(kThreads), 0, ctx->stream,
                         (const float*)x.p, x.ld, scale, shift, (const float*)res.p, res.ld, res.c, alpha, mean, invstd,
                         (const float*)dout.p, dout.ld, voxels, x.c, QCB, VL, (float*)nullptr, 0, (float*)nullptr, 0, 0,
                         partial, (unsigned*)maxes, (const float*)nullptr);
      MSK_LAUNCH_CHECK(ctx);
    }
    msk_launch_scope ls(ctx, "sums_merge");
    hipLaunchKernelGGL(sums_merge_k, dim3(3 * x.c), dim3(nb > 256 ? 256 : 64), 0, ctx->stream, partial, nb, x.c, 4 * QCB, 3, sums, 0, dbeta, dgamma,
                       dalpha, (float*)nullptr);
    MSK_LAUNCH_CHECK(ctx);
    noop_launches(ctx);
    return 0;
  }
  ChanGeom g = chan_geom(x.c);
  int nb = reduce_blocks(voxels, g.VPB, ctx->num_cu, 1);
  size_t bytes = (size_t)g.cblocks * nb * 3 * g.CB * sizeof(float);
  float* partial = (float*)msk_workspace(ctx, bytes);
  if (!partial) return -1;
  if (g_dense12 && d12_shape(x, voxels) && d12_ok(x) && d12_ok(dout) && (res.p == nullptr || (res.c == x.c && d12_ok(res)))) {
    const long groups = voxels * x.c / 12;
    {
      msk_launch_scope ls(ctx, "affine_act_bwd_reduce");
#define D12_RD(C_) hipLaunchKernelGGL((affine_act_bwd_reduce_d12_k<C_, 3>), dim3(nb), dim3(kThreads), 0, ctx->stream, (const float*)x.p, scale, shift, \
                                      (const float*)res.p, alpha, mean, invstd, (const float*)dout.p, groups, g.CB, partial)
      if (x.c == 1) D12_RD(1); else if (x.c == 2) D12_RD(2); else if (x.c == 3) D12_RD(3); else D12_RD(6);
#undef D12_RD
      MSK_LAUNCH_CHECK(ctx);
    }
    msk_launch_scope ls(ctx, "sums_merge");
    hipLaunchKernelGGL(sums_merge_k, dim3(3 * x.c), dim3(nb > 256 ? 256 : 64), 0, ctx->stream, partial, nb, x.c,
                       g.CB, 3, sums, 0, dbeta, dgamma, dalpha, (float*)nullptr);
    MSK_LAUNCH_CHECK(ctx);
    noop_launches(ctx);
    return 0;
  }
  {
    msk_launch_scope ls(ctx, "affine_act_bwd_reduce");
    hipLaunchKernelGGL(affine_act_bwd_reduce_k, dim3(nb, g.cblocks), dim3(kThreads), 0, ctx->stream,
                       (const float*)x.p, x.ld, scale, shift, (const float*)res.p, res.ld, res.c, alpha, mean,
                       invstd, (const float*)dout.p, dout.ld, voxels, x.c, g.CB, g.VPB, partial);
    MSK_LAUNCH_CHECK(ctx);
  }
  {
    msk_launch_scope ls(ctx, "sums_merge");
    hipLaunchKernelGGL(sums_merge_k, dim3(3 * x.c), dim3(nb > 256 ? 256 : 64), 0, ctx->stream, partial, nb, x.c,
                       g.CB, 3, sums, 0, dbeta, dgamma, dalpha, (float*)nullptr);
    MSK_LAUNCH_CHECK(ctx);
    noop_launches(ctx);
  }
  return 0;
}

int msk_affine_act_bwd_apply(msk_ctx* ctx, msk_tensor x, const float* scale, const float* shift, msk_tensor res,
                             const float* alpha, const float* mean, const float* invstd, const float* gamma,
                             msk_tensor dout, const float* sums_total, double M_total, int bn_mode,
                             msk_tensor dx, msk_tensor dres, int dres_acc) {
  return msk_affine_act_bwd_apply_amax(ctx, x, scale, shift, res, alpha, mean, invstd, gamma, dout, sums_total, M_total, bn_mode, dx,
                                       dres, dres_acc, nullptr);
}

int msk_affine_act_bwd_apply_amax(msk_ctx* ctx, msk_tensor x, const float* scale, const float* shift, msk_tensor res,
                                  const float* alpha, const float* mean, const float* invstd, const float* gamma,
                                  msk_tensor dout, const float* sums_total, double M_total, int bn_mode,
                                  msk_tensor dx, msk_tensor dres, int dres_acc, float* dx_amax) {
  (void)gamma;
  MSK_REQUIRE(ctx, same_shape(x, dout), "x/dout shape mismatch");
  if (dx.p) MSK_REQUIRE(ctx, same_shape(x, dx), "x/dx shape mismatch");
  if (dres.p) MSK_REQUIRE(ctx, same_shape(x, dres), "dres must have the full channel count");
  if (bn_mode == 1) MSK_REQUIRE(ctx, sums_total && mean && invstd && scale, "training BN backward needs sums/mean/invstd/scale");
  if (bn_mode == 2) MSK_REQUIRE(ctx, scale != nullptr, "eval BN backward needs scale");
  const long voxels = msk_voxels(x);
  const bool v4 = vec4_ok(x) && vec4_ok(dout) && vec4_ok(dx) && vec4_ok(dres) &&
                  (res.p == nullptr || res.c != x.c || vec4_ok(res));
  const float invM = (float)(1.0 / M_total);
  msk_launch_scope ls(ctx, "affine_act_bwd_apply");
  const int cq = x.c / 4;
  if (g_dense12 && !v4 && d12_shape(x, voxels) && d12_ok(x) && d12_ok(dout) && d12_ok(dx) && d12_ok(dres) &&
      (res.p == nullptr || (res.c == x.c && d12_ok(res)))) {
    const long groups = voxels * x.c / 12;
    const dim3 grid(ew_blocks(groups, ctx->num_cu));
#define D12_AP(C_) hipLaunchKernelGGL(affine_act_bwd_apply_d12_k<C_>, grid, dim3(kThreads), 0, ctx->stream, (const float*)x.p, scale, shift, \
                                      (const float*)res.p, alpha, mean, invstd, (const float*)dout.p, sums_total, invM, bn_mode, (float*)dx.p, \
                                      (float*)dres.p, dres_acc, groups, x.c, (unsigned*)dx_amax)
    if (x.c == 1) D12_AP(1); else if (x.c == 2) D12_AP(2); else if (x.c == 3) D12_AP(3); else D12_AP(6);
#undef D12_AP
    MSK_LAUNCH_CHECK(ctx);
    return 0;
  }
  if (v4 && cq >= 1 && (cq & (cq - 1)) == 0 && cq <= kThreads) {
    int cshift = 0;
    while ((1 << cshift) < cq) ++cshift;
    hipLaunchKernelGGL(affine_act_bwd_apply_cs_k, dim3(ew_blocks(voxels * cq / 2, ctx->num_cu)), dim3(kThreads), 0,
                       ctx->stream, (const float*)x.p, x.ld, scale, shift, (const float*)res.p, res.ld, res.c, alpha, mean,
                       invstd, (const float*)dout.p, dout.ld, sums_total, invM, bn_mode, (float*)dx.p, dx.ld,
                       (float*)dres.p, dres.ld, dres_acc, voxels, x.c, cshift, (unsigned*)dx_amax);
  } else if (v4) {
    hipLaunchKernelGGL(affine_act_bwd_apply_k<4>, dim3(ew_blocks(voxels * x.c / 4, ctx->num_cu)), dim3(kThreads),
                       0, ctx->stream, (const float*)x.p, x.ld, scale, shift, (const float*)res.p, res.ld, res.c,
                       alpha, mean, invstd, (const float*)dout.p, dout.ld, sums_total, invM, bn_mode,
                       (float*)dx.p, dx.ld, (float*)dres.p, dres.ld, dres_acc, voxels, x.c, (unsigned*)dx_amax);
  } else {
    hipLaunchKernelGGL(affine_act_bwd_apply_k<1>, dim3(ew_blocks(voxels * x.c, ctx->num_cu)), dim3(kThreads), 0,
                       ctx->stream, (const float*)x.p, x.ld, scale, shift, (const float*)res.p, res.ld, res.c,
                       alpha, mean, invstd, (const float*)dout.p, dout.ld, sums_total, invM, bn_mode,
                       (float*)dx.p, dx.ld, (float*)dres.p, dres.ld, dres_acc, voxels, x.c, (unsigned*)dx_amax);
  }
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

static int add_act_bwd_impl(msk_ctx* ctx, msk_tensor a, const float* scale, const float* shift, const float* alpha_in,
                           msk_tensor b, const float* alpha, msk_tensor dout, msk_tensor da, msk_tensor db, int db_accumulate,
                           float* dalpha, const float* mean = nullptr, const float* invstd = nullptr, float* unit_sums = nullptr,
                           float* maxes = nullptr, int clear_maxes = 1, float* u_dgamma = nullptr, float* u_dbeta = nullptr,
                           float* u_dalpha = nullptr) {
  // db.p == null (round 6): the second operand's gradient is NOT written -- it equals da, and its consumer reads da instead
  // (msk_conv3d_bwd_bnact_acc's dx_old): one write of the tensor less per join
  MSK_REQUIRE(ctx, same_shape(a, b) && same_shape(a, dout) && same_shape(a, da) && (!db.p || same_shape(a, db)), "shape mismatch");
  MSK_REQUIRE(ctx, db.p || !db_accumulate, "a skipped second gradient cannot accumulate");
  MSK_REQUIRE(ctx, alpha != nullptr && dalpha != nullptr, "join needs alpha and its gradient");
  MSK_REQUIRE(ctx, a.c % 4 == 0 && a.c / 4 <= kThreads && vec4_ok(a) && vec4_ok(b) && vec4_ok(dout) && vec4_ok(da) &&
                       (!db.p || vec4_ok(db)), "join backward needs float4-aligned tensors with C % 4 == 0");
  const long voxels = msk_voxels(a);
  const int QCB = pow2ceil(a.c / 4), VL = kThreads / QCB;
  const int nb = reduce_blocks(voxels, VL, ctx->num_cu, 2);
  if (unit_sums) {  // the unit's BatchNorm/PReLU sums in the same pass
    float* partial4 = (float*)msk_workspace(ctx, (size_t)nb * 4 * 4 * QCB * sizeof(float));
    if (!partial4) return -1;
    if (maxes && clear_maxes) MSK_CHECK_HIP(ctx, hipMemsetAsync(maxes, 0, 2 * kWbfAmaxWays * sizeof(float), ctx->stream));
    {
      msk_launch_scope ls(ctx, "add_act_bwd_unit");
      hipLaunchKernelGGL(affine_act_bwd_reduce_v4_k<2>, dim3(nb), dim3(kThreads), 0, ctx->stream, (const float*)a.p, a.ld, scale,
                         shift, (const float*)b.p, b.ld, b.c, alpha, mean, invstd, (const float*)dout.p, dout.ld, voxels, a.c, QCB,
                         VL, (float*)da.p, da.ld, (float*)db.p, db.ld, db_accumulate, partial4, (unsigned*)maxes, alpha_in);
      MSK_LAUNCH_CHECK(ctx);
    }
    msk_launch_scope ls(ctx, "sums_merge");
    // quantities 0..2 -> unit_sums[3C] (overwritten), quantity 3 -> dalpha of the join (accumulated)
    // ... and the parameter gradients ride along: the unit's (d beta, d gamma, d alpha_inner; nullable) and the join's d alpha
    hipLaunchKernelGGL(sums_merge_k, dim3(4 * a.c), dim3(nb > 256 ? 256 : 64), 0, ctx->stream, partial4, nb, a.c, 4 * QCB, 4, unit_sums, 0, u_dbeta,
                       u_dgamma, u_dalpha, dalpha);
    MSK_LAUNCH_CHECK(ctx);
    noop_launches(ctx);
    return 0;
  }
  float* partial = (float*)msk_workspace(ctx, (size_t)nb * 4 * QCB * sizeof(float));
  if (!partial) return -1;
  {
    msk_launch_scope ls(ctx, "add_act_bwd");
    hipLaunchKernelGGL(affine_act_bwd_reduce_v4_k<1>, dim3(nb), dim3(kThreads), 0, ctx->stream, (const float*)a.p,
                       a.ld, scale, shift, (const float*)b.p, b.ld, b.c, alpha,
                       (const float*)nullptr, (const float*)nullptr, (const float*)dout.p, dout.ld, voxels, a.c, QCB, VL,
                       (float*)da.p, da.ld, (float*)db.p, db.ld, db_accumulate, partial, (unsigned*)nullptr, alpha_in);
    MSK_LAUNCH_CHECK(ctx);
  }
  msk_launch_scope ls(ctx, "sums_merge");
  hipLaunchKernelGGL(sums_merge_k, dim3(a.c), dim3(nb > 256 ? 256 : 64), 0, ctx->stream, partial, nb, a.c, 4 * QCB, 1, dalpha, 1);
  MSK_LAUNCH_CHECK(ctx);
  noop_launches(ctx);
  return 0;
}

int msk_add_act_bwd(msk_ctx* ctx, msk_tensor a, msk_tensor b, const float* alpha, msk_tensor dout, msk_tensor da,
                    msk_tensor db, int db_accumulate, float* dalpha) {
  return add_act_bwd_impl(ctx, a, nullptr, nullptr, nullptr, b, alpha, dout, da, db, db_accumulate, dalpha);
}

int msk_add_act_join_bwd(msk_ctx* ctx, msk_tensor y, const float* scale, const float* shift, const float* alpha_inner,
                         msk_tensor res, const float* alpha_outer, msk_tensor dout, msk_tensor da, msk_tensor dres,
                         int dres_accumulate, float* dalpha_outer) {
  MSK_REQUIRE(ctx, scale && shift && alpha_inner, "the unit's BatchNorm coefficients and PReLU slope");
  return add_act_bwd_impl(ctx, y, scale, shift, alpha_inner, res, alpha_outer, dout, da, dres, dres_accumulate, dalpha_outer);
}

int msk_add_act_join_bwd_ex(msk_ctx* ctx, msk_tensor y, const float* scale, const float* shift, const float* alpha_inner,
                            msk_tensor res, const float* alpha_outer, const float* mean, const float* invstd, msk_tensor dout,
                            msk_tensor da, msk_tensor dres, int dres_accumulate, float* dalpha_outer, float* unit_sums,
                            float* maxes) {
  MSK_REQUIRE(ctx, scale && shift && alpha_inner && mean && invstd && unit_sums, "the unit's BatchNorm coefficients, statistics and sums buffer");
  return add_act_bwd_impl(ctx, y, scale, shift, alpha_inner, res, alpha_outer, dout, da, dres, dres_accumulate, dalpha_outer, mean,
                          invstd, unit_sums, maxes);
}

int msk_add_act_join_bwd_pg(msk_ctx* ctx, msk_tensor y, const float* scale, const float* shift, const float* alpha_inner,
                            msk_tensor res, const float* alpha_outer, const float* mean, const float* invstd, msk_tensor dout,
                            msk_tensor da, msk_tensor dres, int dres_accumulate, float* dalpha_outer, float* unit_sums,
                            float* maxes, int clear_maxes, float* unit_dgamma, float* unit_dbeta, float* unit_dalpha) {
  MSK_REQUIRE(ctx, scale && shift && alpha_inner && mean && invstd && unit_sums, "the unit's BatchNorm coefficients, statistics and sums buffer");
  return add_act_bwd_impl(ctx, y, scale, shift, alpha_inner, res, alpha_outer, dout, da, dres, dres_accumulate, dalpha_outer, mean,
                          invstd, unit_sums, maxes, clear_maxes, unit_dgamma, unit_dbeta, unit_dalpha);
}

int msk_affine_act_param_grads(msk_ctx* ctx, int C, const float* sums, float* dgamma, float* dbeta, float* dalpha,
                               int accumulate) {
  msk_launch_scope ls(ctx, "affine_act_param_grads");
  hipLaunchKernelGGL(param_grads_k, dim3(msk_cdiv(C, 64)), dim3(64), 0, ctx->stream, C, sums, dgamma, dbeta,
                     dalpha, accumulate);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

int msk_bn_bias_grad(msk_ctx* ctx, int C, const float* sums, const float* scale, float* dbias, int accumulate) {
  MSK_REQUIRE(ctx, sums && scale && dbias, "null pointer");
  msk_launch_scope ls(ctx, "bn_bias_grad");
  hipLaunchKernelGGL(bias_grad_from_sums_k, dim3(msk_cdiv(C, 64)), dim3(64), 0, ctx->stream, C, sums, scale, dbias,
                     accumulate);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

int msk_copy_scale(msk_ctx* ctx, msk_tensor src, const float* mask, msk_tensor dst, int accumulate) {
  return msk_copy_scale_amax(ctx, src, mask, dst, accumulate, nullptr);
}

int msk_copy_scale_amax(msk_ctx* ctx, msk_tensor src, const float* mask, msk_tensor dst, int accumulate, float* dst_amax) {
  MSK_REQUIRE(ctx, same_shape(src, dst), "src/dst shape mismatch");
  const long voxels = msk_voxels(src);
  const long vpn = (long)src.d * src.h * src.w;
  msk_launch_scope ls(ctx, "copy_scale");
  if (vec4_ok(src) && vec4_ok(dst)) {
    hipLaunchKernelGGL(copy_scale_k<4>, dim3(ew_blocks(voxels * src.c / 4, ctx->num_cu)), dim3(kThreads), 0,
                       ctx->stream, (const float*)src.p, src.ld, mask, (float*)dst.p, dst.ld, voxels, vpn, src.c,
                       accumulate, (unsigned*)dst_amax);
  } else {
    hipLaunchKernelGGL(copy_scale_k<1>, dim3(ew_blocks(voxels * src.c, ctx->num_cu)), dim3(kThreads), 0,
                       ctx->stream, (const float*)src.p, src.ld, mask, (float*)dst.p, dst.ld, voxels, vpn, src.c,
                       accumulate, (unsigned*)dst_amax);
  }
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

int msk_dropout_mask(msk_ctx* ctx, uint64_t seed, uint64_t step, uint32_t site, int count, float p, float* mask) {
  MSK_REQUIRE(ctx, p >= 0.f && p < 1.f, "dropout p must be in [0,1)");
  msk_launch_scope ls(ctx, "dropout_mask");
  hipLaunchKernelGGL(dropout_mask_k, dim3(msk_cdiv(count, 256)), dim3(256), 0, ctx->stream, seed, step, site,
                     count, p, mask);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

int msk_channel_sum(msk_ctx* ctx, msk_tensor x, float* out, int accumulate) {
  const long voxels = msk_voxels(x);
  ChanGeom g = chan_geom(x.c);
  int nb = reduce_blocks(voxels, g.VPB, ctx->num_cu, 3);
  float* partial = (float*)msk_workspace(ctx, (size_t)g.cblocks * nb * g.CB * sizeof(float));
  if (!partial) return -1;
  if (g_dense12 && d12_shape(x, voxels) && d12_ok(x)) {
    const long groups = voxels * x.c / 12;
    {
      msk_launch_scope ls(ctx, "channel_sum_partial");
#define D12_CS(C_) hipLaunchKernelGGL((affine_act_bwd_reduce_d12_k<C_, 1>), dim3(nb), dim3(kThreads), 0, ctx->stream, (const float*)x.p, (const float*)nullptr, \
                                      (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, \
                                      (const float*)nullptr, groups, g.CB, partial)
      if (x.c == 1) D12_CS(1); else if (x.c == 2) D12_CS(2); else if (x.c == 3) D12_CS(3); else D12_CS(6);
#undef D12_CS
      MSK_LAUNCH_CHECK(ctx);
    }
    msk_launch_scope ls(ctx, "sums_merge");
    hipLaunchKernelGGL(sums_merge_k, dim3(x.c), dim3(nb > 256 ? 256 : 64), 0, ctx->stream, partial, nb, x.c, g.CB, 1,
                       out, accumulate);
    MSK_LAUNCH_CHECK(ctx);
    noop_launches(ctx);
    return 0;
  }
  {
    msk_launch_scope ls(ctx, "channel_sum_partial");
    hipLaunchKernelGGL(channel_sum_partial_k, dim3(nb, g.cblocks), dim3(kThreads), 0, ctx->stream,
                       (const float*)x.p, x.ld, voxels, x.c, g.CB, g.VPB, partial);
    MSK_LAUNCH_CHECK(ctx);
  }
  {
    msk_launch_scope ls(ctx, "sums_merge");
    hipLaunchKernelGGL(sums_merge_k, dim3(x.c), dim3(nb > 256 ? 256 : 64), 0, ctx->stream, partial, nb, x.c, g.CB, 1,
                       out, accumulate);
    MSK_LAUNCH_CHECK(ctx);
    noop_launches(ctx);
  }
  return 0;
}

int msk_softmax_c(msk_ctx* ctx, msk_tensor x, msk_tensor out) {
  MSK_REQUIRE(ctx, same_shape(x, out), "x/out shape mismatch");
  const long voxels = msk_voxels(x);
  msk_launch_scope ls(ctx, "softmax_c");
  hipLaunchKernelGGL(softmax_c_k, dim3(ew_blocks(voxels, ctx->num_cu)), dim3(kThreads), 0, ctx->stream, (const float*)x.p, x.ld,
                     voxels, x.c, (float*)out.p, out.ld);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

int msk_argmax_c(msk_ctx* ctx, msk_tensor x, int32_t* out) {
  const long voxels = msk_voxels(x);
  msk_launch_scope ls(ctx, "argmax_c");
  hipLaunchKernelGGL(argmax_k, dim3(ew_blocks(voxels, ctx->num_cu)), dim3(kThreads), 0, ctx->stream,
                     (const float*)x.p, x.ld, voxels, x.c, out);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// device scalars for the fp16 two-piece pipelines (msk_wbf.h, NP = 2): max |x| of a tensor
// ---------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(kThreads)
absmax_k(const float* __restrict__ x, int ld, int C, long voxels, unsigned* __restrict__ slot) {
  float m = 0.f;
  if (ld == C && (voxels * C) % 4 == 0 && (((uintptr_t)x) & 15) == 0) {
    // dense tensor: a flat float4 stream, eight loads in flight, no index arithmetic
    const float4* p = reinterpret_cast<const float4*>(x);
    const long total = voxels * C / 4, stride = (long)gridDim.x * blockDim.x;
    auto fold = [&](const float4 q) { m = fmaxf(fmaxf(m, fmaxf(fabsf(q.x), fabsf(q.y))), fmaxf(fabsf(q.z), fabsf(q.w))); };
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < total; i += 8 * stride) {
      float4 q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] = p[i + u * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) fold(q[u]);
    }
    for (; i < total; i += stride) fold(p[i]);
  } else if (C % 4 == 0 && ld % 4 == 0 && (((uintptr_t)x) & 15) == 0) {
    const int c4 = C >> 2;
    const long total = voxels * c4, stride = (long)gridDim.x * blockDim.x;
    auto at = [&](long i) {
      const long v = i / c4;
      return *reinterpret_cast<const float4*>(x + v * ld + (int)(i - v * c4) * 4);
    };
    auto fold = [&](const float4 q) { m = fmaxf(fmaxf(m, fmaxf(fabsf(q.x), fabsf(q.y))), fmaxf(fabsf(q.z), fabsf(q.w))); };
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < total; i += 4 * stride) {  // four loads in flight
      const float4 a = at(i), b = at(i + stride), c = at(i + 2 * stride), d = at(i + 3 * stride);
      fold(a); fold(b); fold(c); fold(d);
    }
    for (; i < total; i += stride) fold(at(i));
  } else {
    const long total = voxels * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
      const long v = i / C;
      m = fmaxf(m, fabsf(x[v * ld + (i - v * C)]));
    }
  }
  block_atomic_max(slot, m);
}
// amax[0] = bound of max |dy| of the BatchNorm/PReLU backward (WbfBnBwd):
//   |dy| <= max_c |scale_c| * (max|du| + max_c |s1_c| + max|xhat| * max_c |s2_c|)
__global__ void bn_bwd_bound_k(int C, const float* __restrict__ scale, const float* __restrict__ sums, float invM,
                               const float* __restrict__ maxes /*two amax arrays: max|du|, max|xhat|*/, float* __restrict__ out) {
  __shared__ float sh[3][64];
  float a = 0.f, b = 0.f, c_ = 0.f;
  for (int c = threadIdx.x; c < C; c += 64) {
    a = fmaxf(a, fabsf(scale[c]));
    b = fmaxf(b, fabsf(sums[c] * invM));
    c_ = fmaxf(c_, fabsf(sums[C + c] * invM));
  }
  sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = b; sh[2][threadIdx.x] = c_;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 64; ++i) {
      a = fmaxf(a, sh[0][i]); b = fmaxf(b, sh[1][i]); c_ = fmaxf(c_, sh[2][i]);
    }
    out[0] = a * (wbf_amax_of(maxes) + b + wbf_amax_of(maxes + kWbfAmaxWays) * c_);
  }
}
}  // namespace

// Ring of amax arrays, one ring per stream (msk_side_scope swaps them with the streams): a slot is handed out ZEROED without
// a memset of its own -- the ring is cleared half by half, on the stream that allocates from it, at the moment the
// allocation cursor enters a half.  The slots of that half were handed out at least half a ring (512 arrays = many training
// steps) ago and their consumers were enqueued before the streams last joined, so nothing still reads them.  (Round 2
// issued one hipMemsetAsync per request: ~45 fill kernels per step.)
float* msk_scalar_slots(msk_ctx* ctx, int n) {
  constexpr int kRing = 1024 * kWbfAmaxWays, kHalf = kRing / 2;
  n *= kWbfAmaxWays;
  if (n > kHalf) {
    msk_fail(ctx, __FILE__, __LINE__, "msk_scalar_slots", "request larger than half the ring");
    return nullptr;
  }
  if (!ctx->scalar_ring) {
    if (hipMalloc((void**)&ctx->scalar_ring, kRing * sizeof(float)) != hipSuccess) {
      msk_fail(ctx, __FILE__, __LINE__, "msk_scalar_slots", "hipMalloc failed");
      return nullptr;
    }
    ctx->scalar_next = 0;
  }
  int start = ctx->scalar_next;
  if (start % kHalf + n > kHalf) start = (start / kHalf + 1) * kHalf;  // a request never straddles the halves
  if (start >= kRing) start = 0;
  if (start % kHalf == 0) {
    if (hipMemsetAsync(ctx->scalar_ring + start, 0, kHalf * sizeof(float), ctx->stream) != hipSuccess) {
      msk_fail(ctx, __FILE__, __LINE__, "msk_scalar_slots", "hipMemsetAsync failed");
      return nullptr;
    }
  }
  ctx->scalar_next = start + n;
  ctx->scalar_served += n / kWbfAmaxWays;
  return ctx->scalar_ring + start;
}

const float* msk_absmax(msk_ctx* ctx, const float* x, int ld, int C, long voxels, float* dst) {
  float* slot = dst;
  if (slot) {
    if (hipMemsetAsync(slot, 0, kWbfAmaxWays * sizeof(float), ctx->stream) != hipSuccess) {
      msk_fail(ctx, __FILE__, __LINE__, "absmax", "hipMemsetAsync failed");
      return nullptr;
    }
  } else {
    slot = msk_scalar_slots(ctx, 1);
  }
  if (!slot) return nullptr;
  const char* tag = "absmax";
  if (ctx->prof && ctx->prof_shapes) {
    char buf[96];
    snprintf(buf, sizeof(buf), "absmax[c=%d,ld=%d,voxels=%ld]", C, ld, voxels);
    tag = msk_intern_tag(ctx, buf);
  }
  msk_launch_scope ls(ctx, tag);
  long ab = (voxels * C / 4 + kThreads - 1) / kThreads;
  if (ab > 8L * ctx->num_cu) ab = 8L * ctx->num_cu;
  if (ab < 1) ab = 1;
  hipLaunchKernelGGL(absmax_k, dim3((unsigned)ab), dim3(kThreads), 0, ctx->stream, x, ld, C, voxels,
                     (unsigned*)slot);
  if (hipGetLastError() != hipSuccess) {
    msk_fail(ctx, __FILE__, __LINE__, "absmax", "kernel launch");
    return nullptr;
  }
  return slot;
}

// bound of max |dy| for msk_conv3d_bwd_bnact's NP = 2 form from the maxima msk_affine_act_bwd_reduce_ex left in maxes[2]
const float* msk_bn_bwd_bound(msk_ctx* ctx, int C, const float* scale, const float* sums, double M_total, const float* maxes) {
  float* slot = msk_scalar_slots(ctx, 1);
  if (!slot) return nullptr;
  msk_launch_scope ls(ctx, "bn_bwd_bound");
  hipLaunchKernelGGL(bn_bwd_bound_k, dim3(1), dim3(64), 0, ctx->stream, C, scale, sums, (float)(1.0 / M_total), maxes, slot);
  if (hipGetLastError() != hipSuccess) {
    msk_fail(ctx, __FILE__, __LINE__, "bn_bwd_bound", "kernel launch");
    return nullptr;
  }
  return slot;
}


// ---------------------------------------------------------------------------
// ELUCons(elu=True) (vnet.py:25-29): paddle.nn.ELU(alpha) as its own pass.  The shipped configs use PReLU (elu: False; the
// reference notes NaN gradients with ELU, core/train.py:139), so ELU is not fused into the convolution / BatchNorm kernels:
// the units run their PReLU-less forms and these two kernels follow / precede them.  The derivative is taken from the
// OUTPUT (out > 0 ? 1 : out + alpha), so the pre-activation is not kept.
// ---------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(kThreads)
elu_fwd_k(const float* __restrict__ x, int xld, float* __restrict__ out, int old, int C, long voxels, float alpha) {
  const long total = voxels * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long v = i / C;
    const int c = (int)(i - v * C);
    const float a = x[v * xld + c];
    out[v * old + c] = a > 0.f ? a : alpha * expm1f(a);
  }
}
__global__ void __launch_bounds__(kThreads)
elu_bwd_k(const float* __restrict__ out, int old, const float* __restrict__ dout, int dld, float* __restrict__ dx, int xld,
          int C, long voxels, float alpha, int accumulate) {
  const long total = voxels * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long v = i / C;
    const int c = (int)(i - v * C);
    const float o = out[v * old + c];
    const float g = dout[v * dld + c] * (o > 0.f ? 1.f : o + alpha);
    float* d = dx + v * xld + c;
    *d = accumulate ? *d + g : g;
  }
}
}  // namespace

extern "C" {
int msk_elu_fwd(msk_ctx* ctx, msk_tensor x, float alpha, msk_tensor out) {
  MSK_REQUIRE(ctx, x.c == out.c && msk_voxels(x) == msk_voxels(out), "shape mismatch");
  const long voxels = msk_voxels(x);
  long blocks = (voxels * x.c + kThreads - 1) / kThreads;
  if (blocks > 16L * ctx->num_cu) blocks = 16L * ctx->num_cu;
  if (blocks < 1) blocks = 1;
  msk_launch_scope ls(ctx, "elu_fwd");
  hipLaunchKernelGGL(elu_fwd_k, dim3((unsigned)blocks), dim3(kThreads), 0, ctx->stream, (const float*)x.p, x.ld, (float*)out.p, out.ld,
                     x.c, voxels, alpha);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}
int msk_elu_bwd(msk_ctx* ctx, msk_tensor out, msk_tensor dout, float alpha, msk_tensor dx, int accumulate) {
  MSK_REQUIRE(ctx, out.c == dout.c && out.c == dx.c && msk_voxels(out) == msk_voxels(dout) && msk_voxels(out) == msk_voxels(dx),
              "shape mismatch");
  const long voxels = msk_voxels(out);
  long blocks = (voxels * out.c + kThreads - 1) / kThreads;
  if (blocks > 16L * ctx->num_cu) blocks = 16L * ctx->num_cu;
  if (blocks < 1) blocks = 1;
  msk_launch_scope ls(ctx, "elu_bwd");
  hipLaunchKernelGGL(elu_bwd_k, dim3((unsigned)blocks), dim3(kThreads), 0, ctx->stream, (const float*)out.p, out.ld,
                     (const float*)dout.p, dout.ld, (float*)dx.p, dx.ld, out.c, voxels, alpha, accumulate);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}
}  // extern "C"

void msk_set_dense12(int v) { g_dense12 = v; }
void msk_set_reduce_vpl(int v) { if (v > 0) g_reduce_vpl = v; }
void msk_set_reduce_vpl_site(int v) { if (v >= 0 && v < 4000) g_reduce_vpl_site[v / 1000] = v % 1000; }
void msk_set_ew_caps(int ew, int red) {
  if (ew > 0) g_ew_cap = ew;
  if (red > 0) g_reduce_cap = red;
}
