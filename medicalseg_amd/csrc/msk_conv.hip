// Convolution entry points of the C ABI: weight packing, dispatch between the MFMA
// kernels (msk_conv_mfma.hip) and the VALU reference kernels below, split-K reduce.
//
// The six public ops (Conv3D fwd/dgrad/wgrad, Conv3DTranspose fwd/dgrad/wgrad) reduce to
// two device problems (msk_conv.h): a "gather convolution" and a weight gradient.
#include "msk_conv.h"
#include "msk_wbf.h"

namespace {

constexpr int kThreads = 256;

// ---------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------
// one element of a packed weight image (layouts: msk_conv.h)
__device__ __forceinline__ float pack_weights_elem(const float* __restrict__ w, int B, int taps, int swap, int flip, int kd, int kh,
                                                   int kw, int mfma, int K, int N, int KC, int npad, long idx) {
  int tap, k, n;
  if (mfma) {
    // idx = ((((tap*KC + kc)*2 + h)*npad + n)*4 + q)
    int q = (int)(idx & 3);
    long r = idx >> 2;
    n = (int)(r % npad);
    r /= npad;
    int h = (int)(r & 1);
    r >>= 1;
    int kc = (int)(r % KC);
    tap = (int)(r / KC);
    k = kc * 8 + h * 4 + q;
  } else {
    const int NP = npad > 0 ? npad : N;  // direct layout with a padded row pitch (columns >= N are zero)
    n = (int)(idx % NP);
    long r = idx / NP;
    k = (int)(r % K);
    tap = (int)(r / K);
  }
  float v = 0.f;
  if (k < K && n < N) {
    int st = tap;
    if (flip) {
      int a = tap / (kh * kw), b = (tap / kw) % kh, c = tap % kw;
      st = ((kd - 1 - a) * kh + (kh - 1 - b)) * kw + (kw - 1 - c);
    }
    int ia = swap ? n : k, ib = swap ? k : n;  // w[a][b][tap]
    v = w[((long)ia * B + ib) * taps + st];
  }
  return v;
}

__global__ void __launch_bounds__(kThreads)
pack_weights_k(const float* __restrict__ w, int A, int B, int taps, int swap, int flip, int kd, int kh,
               int kw, int mfma, int K, int N, int KC, int npad, float* __restrict__ out, long total) {
  (void)A;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x)
    out[idx] = pack_weights_elem(w, B, taps, swap, flip, kd, kh, kw, mfma, K, N, KC, npad, idx);
}

// ---------------------------------------------------------------------------
// Round 5: the packed images of the non-Winograd kernels are CACHED (the Winograd pipelines have had WbfPackCache since round 3).
// Every call of a kernel == stride convolution, a gather kernel or a 1x1x1 head used to launch its own 5-us pack kernel in
// front of the convolution: 20 launches per VNet step on the compute stream.  A row = (canonical weight pointer, layout
// parameters) -> persistent device image; rows go stale through the same hooks as the Winograd rows (msk_weights_changed_impl /
// msk_weights_freed_impl) and are rebuilt TOGETHER, one launch over a device-side descriptor table (blockIdx.y = row), where
// the Winograd rows are: at the end of the optimizer kernels (msk_wbf_prepack_impl; per slice on the optimizer's stream in the
// eager form, msk_wbf_prepack_range_impl).  A row that is still stale when a convolution asks for it is rebuilt alone, on the
// asking stream.  kind 0: pack_weights_k layouts; kind 1: the scatter layout Bf[kc][h][jpad][4] (msk_conv_scatter.hip).
// ---------------------------------------------------------------------------
struct SmallPackDesc {
  const float* w;
  float* out;
  long total;      // elements (kind 0: floats, kind 1: float4)
  int kind, A, B, taps, swap, flip, kd, kh, kw, mfma, K, N, KC, npad;
};
struct SmallPackList {
  int n;
  unsigned char rows[60];
};
struct SmallPackEntry {
  SmallPackDesc d{};
  size_t bytes = 0;
  bool live = false, valid = false;
  long last_use = 0;
};
struct SmallPackCache {
  static constexpr int kRows = 128;
  SmallPackEntry e[kRows];
  SmallPackDesc* table = nullptr;   // device copy of the descriptors
  long tick = 0;
};

__global__ void small_pack_desc_store_k(SmallPackDesc* slot, SmallPackDesc d) { *slot = d; }

__global__ void __launch_bounds__(kThreads)
small_pack_batch_k(const SmallPackDesc* __restrict__ table, SmallPackList l) {
  const SmallPackDesc d = table[l.rows[blockIdx.y]];
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < d.total; idx += (long)gridDim.x * blockDim.x) {
    if (d.kind == 0) {
      d.out[idx] = pack_weights_elem(d.w, d.B, d.taps, d.swap, d.flip, d.kd, d.kh, d.kw, d.mfma, d.K, d.N, d.KC, d.npad, idx);
    } else {
      // Bf[kc][h][jpad][4]: k = kc*8 + h*4 + q, j = tap*CN + cn (zero padded to jpad = d.npad, K = CK to 8*KC)
      const int jpad = d.npad;
      const int j = (int)(idx % jpad);
      const int h = (int)((idx / jpad) & 1);
      const int kc = (int)(idx / (2L * jpad));
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (j < d.taps * d.N) {
        const int tap = j / d.N, n = j % d.N;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k = kc * 8 + h * 4 + q;
          if (k < d.K) {
            const int ia = d.swap ? n : k, ib = d.swap ? k : n;  // canonical w[a][b][tap]
            v[q] = d.w[((long)ia * d.B + ib) * d.taps + tap];
          }
        }
      }
      reinterpret_cast<float4*>(d.out)[idx] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

bool small_pack_same(const SmallPackDesc& a, const SmallPackDesc& b) {
  return a.w == b.w && a.kind == b.kind && a.A == b.A && a.B == b.B && a.taps == b.taps && a.swap == b.swap && a.flip == b.flip &&
         a.kd == b.kd && a.kh == b.kh && a.kw == b.kw && a.mfma == b.mfma && a.K == b.K && a.N == b.N && a.KC == b.KC &&
         a.npad == b.npad;
}

int small_pack_build(msk_ctx* ctx, SmallPackCache* c, const std::vector<int>& rows) {
  for (size_t i0 = 0; i0 < rows.size(); i0 += sizeof(SmallPackList::rows)) {
    SmallPackList l{};
    long most = 0;
    for (size_t i = i0; i < rows.size() && l.n < (int)sizeof(l.rows); ++i) {
      l.rows[l.n++] = (unsigned char)rows[i];
      if (c->e[rows[i]].d.total > most) most = c->e[rows[i]].d.total;
    }
    long gx = (most + kThreads - 1) / kThreads;
    if (gx > 2L * ctx->num_cu) gx = 2L * ctx->num_cu;
    if (gx < 1) gx = 1;
    msk_launch_scope ls(ctx, "pack_weights_batch");
    hipLaunchKernelGGL(small_pack_batch_k, dim3((unsigned)gx, (unsigned)l.n), dim3(kThreads), 0, ctx->stream, c->table, l);
    MSK_LAUNCH_CHECK(ctx);
    for (int i = 0; i < l.n; ++i) c->e[l.rows[i]].valid = true;
  }
  return 0;
}

// the cached image of a row (created on first use); nullptr on failure (error set)
const float* small_pack_get(msk_ctx* ctx, SmallPackDesc key, size_t bytes) {
  if (!ctx->spack) {
    SmallPackCache* nc = new SmallPackCache();
    if (hipMalloc((void**)&nc->table, sizeof(SmallPackDesc) * SmallPackCache::kRows) != hipSuccess) {
      delete nc;
      msk_fail(ctx, __FILE__, __LINE__, "small_pack_get", "hipMalloc failed");
      return nullptr;
    }
    ctx->spack = nc;
  }
  SmallPackCache* c = (SmallPackCache*)ctx->spack;
  c->tick += 1;
  int row = -1, free_row = -1, lru = -1;
  for (int r = 0; r < SmallPackCache::kRows; ++r) {
    SmallPackEntry& e = c->e[r];
    if (!e.live) {
      if (free_row < 0) free_row = r;
      continue;
    }
    if (small_pack_same(e.d, key)) {
      row = r;
      break;
    }
    if (lru < 0 || e.last_use < c->e[lru].last_use) lru = r;
  }
  if (row < 0) {
    row = free_row >= 0 ? free_row : lru;
    SmallPackEntry& e = c->e[row];
    // a row keeps its image's allocation across tenants (stream order protects a reused image: it is rewritten on the stream
    // whose kernels read it, or behind that stream's tail); a larger one is allocated behind everything in flight
    float* out = e.d.out;
    if (out && e.live && ctx->side != nullptr) {
      // a live row changes tenant (all rows in use: LRU): its image and descriptor are rewritten on THIS stream -- the other
      // stream may still hold launches that read the old tenant's image, so this one waits for the other's current tail first
      // (rare: more than kRows live weight images; advisor, round 5)
      static thread_local hipEvent_t ev = nullptr;
      if (!ev) hipEventCreateWithFlags(&ev, hipEventDisableTiming);
      if (ev) {
        hipEventRecord(ev, ctx->side);
        hipStreamWaitEvent(ctx->stream, ev, 0);
      }
    }
    if (out && e.bytes < bytes) {
      hipStreamSynchronize(ctx->stream);
      if (ctx->side) hipStreamSynchronize(ctx->side);
      hipFree(out);
      out = nullptr;
      e.d.out = nullptr;
      e.bytes = 0;
    }
    if (!out) {
      if (hipMalloc((void**)&out, bytes) != hipSuccess) {
        e.live = false;
        msk_fail(ctx, __FILE__, __LINE__, "small_pack_get", "hipMalloc failed");
        return nullptr;
      }
      e.bytes = bytes;
    }
    e.d = key;
    e.d.out = out;
    e.live = true;
    e.valid = false;
    hipLaunchKernelGGL(small_pack_desc_store_k, dim3(1), dim3(1), 0, ctx->stream, c->table + row, e.d);
    if (hipGetLastError() != hipSuccess) {
      msk_fail(ctx, __FILE__, __LINE__, "small_pack_get", "kernel launch");
      return nullptr;
    }
  }
  SmallPackEntry& e = c->e[row];
  e.last_use = c->tick;
  if (!e.valid && small_pack_build(ctx, c, std::vector<int>{row}) != 0) return nullptr;
  return e.d.out;
}

// ---------------------------------------------------------------------------
// VALU reference gather convolution (any shape).  One thread per (dst voxel, out channel).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
gconv_direct_k(GConv g, const float* __restrict__ wp /*[tap][CK][CN]*/) {
  const long M = (long)g.N * g.DD * g.DH * g.DW;
  const long total = M * g.CN;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long m = idx / g.CN;
    const int cn = (int)(idx - m * g.CN);
    const int ow = (int)(m % g.DW);
    const int oh = (int)((m / g.DW) % g.DH);
    const int od = (int)((m / ((long)g.DW * g.DH)) % g.DD);
    const int n = (int)(m / ((long)g.DW * g.DH * g.DD));
    float acc = g.bias ? g.bias[cn] : 0.f;
    for (int a = 0; a < g.kd; ++a) {
      int id;
      if (!g.transposed) {
        id = od * g.sd - g.pd + a;
      } else {
        int t = od + g.pd - a;
        if (t < 0 || t % g.sd) continue;
        id = t / g.sd;
      }
      if (id < 0 || id >= g.SD) continue;
      for (int b = 0; b < g.kh; ++b) {
        int ih;
        if (!g.transposed) {
          ih = oh * g.sh - g.ph + b;
        } else {
          int t = oh + g.ph - b;
          if (t < 0 || t % g.sh) continue;
          ih = t / g.sh;
        }
        if (ih < 0 || ih >= g.SH) continue;
        for (int c = 0; c < g.kw; ++c) {
          int iw;
          if (!g.transposed) {
            iw = ow * g.sw - g.pw + c;
          } else {
            int t = ow + g.pw - c;
            if (t < 0 || t % g.sw) continue;
            iw = t / g.sw;
          }
          if (iw < 0 || iw >= g.SW) continue;
          const float* sp = g.src + ((((long)n * g.SD + id) * g.SH + ih) * g.SW + iw) * g.sld;
          const int tap = (a * g.kh + b) * g.kw + c;
          const float* wq = wp + (long)tap * g.CK * g.CN + cn;
          for (int k = 0; k < g.CK; ++k) acc = fmaf(sp[k], wq[(long)k * g.CN], acc);
        }
      }
    }
    float* dp = g.dst + m * g.dld + cn;
    *dp = g.accumulate ? *dp + acc : acc;
  }
}

// ---------------------------------------------------------------------------
// VALU reference weight gradient: block = (tap, 16x16 (ca,cb) tile, voxel split)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
wgrad_direct_k(WGrad g, int splits, float* __restrict__ partial /*[split][tap][CA][CB]*/) {
  const int t = threadIdx.x;
  const int ca_tiles = (g.CA + 15) / 16, cb_tiles = (g.CB + 15) / 16;
  int b = blockIdx.x;
  const int cbt = b % cb_tiles;
  b /= cb_tiles;
  const int cat = b % ca_tiles;
  const int tap = b / ca_tiles;
  const int split = blockIdx.y;
  const int ca = cat * 16 + t / 16, cb = cbt * 16 + t % 16;
  const int a = tap / (g.kh * g.kw), bb = (tap / g.kw) % g.kh, c = tap % g.kw;
  const long M = (long)g.N * g.BD * g.BH * g.BW;
  const long per = (M + splits - 1) / splits;
  const long m0 = split * per;
  long m1 = m0 + per;
  if (m1 > M) m1 = M;
  float acc = 0.f;
  const bool live = ca < g.CA && cb < g.CB;
  for (long m = m0; m < m1; ++m) {
    const int ow = (int)(m % g.BW);
    const int oh = (int)((m / g.BW) % g.BH);
    const int od = (int)((m / ((long)g.BW * g.BH)) % g.BD);
    const int n = (int)(m / ((long)g.BW * g.BH * g.BD));
    const int id = od * g.sd - g.pd + a, ih = oh * g.sh - g.ph + bb, iw = ow * g.sw - g.pw + c;
    if (id < 0 || id >= g.AD || ih < 0 || ih >= g.AH || iw < 0 || iw >= g.AW) continue;
    if (live) {
      const float av = g.A[((((long)n * g.AD + id) * g.AH + ih) * g.AW + iw) * g.ald + ca];
      const float bv = g.B[m * g.bld + cb];
      acc = fmaf(av, bv, acc);
    }
  }
  if (live) partial[(((long)split * (g.kd * g.kh * g.kw) + tap) * g.CA + ca) * g.CB + cb] = acc;
}

__global__ void __launch_bounds__(kThreads)
wgrad_reduce_k(const float* __restrict__ partial, int splits, int stride, int taps, int CA, int CB,
               float* __restrict__ dw, int accumulate) {
  // block = 64 consecutive outputs x 4 split slices: coalesced 256-byte reads of every slab,
  // 4-way split parallelism + 4 independent accumulators per thread (fixed order: deterministic)
  __shared__ double sh[4][64];
  const long per = (long)taps * CA * CB;
  const long pitch = per * stride;  // slab k of the (pre-reduced) set sits at k*stride
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  for (long base = (long)blockIdx.x * 64; base < per; base += (long)gridDim.x * 64) {
    const long idx = base + lane;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (idx < per) {
      int k = slice;
      for (; k + 12 < splits; k += 16) {
        s0 += partial[(long)k * pitch + idx];
        s1 += partial[(long)(k + 4) * pitch + idx];
        s2 += partial[(long)(k + 8) * pitch + idx];
        s3 += partial[(long)(k + 12) * pitch + idx];
      }
      for (; k < splits; k += 4) s0 += partial[(long)k * pitch + idx];
    }
    sh[slice][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice == 0 && idx < per) {
      const double s = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
      // idx = (tap*CA + ca)*CB + cb
      const int cb = (int)(idx % CB);
      const long r = idx / CB;
      const int ca = (int)(r % CA);
      const int tap = (int)(r / CA);
      float* o = dw + ((long)cb * CA + ca) * taps + tap;
      *o = accumulate ? *o + (float)s : (float)s;
    }
    __syncthreads();
  }
}

// 1x1x1 convolution with few channels (out_tr.conv2, ncls -> ncls, vnet.py:169) and its data
// gradient: pure HBM streaming -- one thread per voxel, the CK x CN weight block in LDS.
__global__ void __launch_bounds__(kThreads)
pointwise_small_k(GConv g, const float* __restrict__ wp /*[CK][CN]*/) {
  __shared__ float ws[8 * 8];
  __shared__ float bs[8];
  for (int i = threadIdx.x; i < 64; i += blockDim.x) {
    const int k = i / 8, n = i % 8;
    ws[i] = (k < g.CK && n < g.CN) ? wp[k * g.CN + n] : 0.f;
  }
  for (int i = threadIdx.x; i < 8; i += blockDim.x) bs[i] = (g.bias && i < g.CN) ? g.bias[i] : 0.f;
  __syncthreads();
  const long M = (long)g.N * g.DD * g.DH * g.DW;
  for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
    float x[8];
    const float* sp = g.src + m * g.sld;
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = k < g.CK ? sp[k] : 0.f;
    float* dp = g.dst + m * g.dld;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      if (n < g.CN) {
        float acc = bs[n];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = fmaf(x[k], ws[k * 8 + n], acc);
        dp[n] = g.accumulate ? dp[n] + acc : acc;
      }
    }
  }
}

// 1x1x1 convolution with up to 32 channels on both sides (out_tr.conv2 of the 20-class MRI model, vnet.py:169, and its data
// gradient): 400 FMAs and 160 bytes per voxel -- HBM streaming.  One thread per voxel, the voxel's channels in registers
// (16-byte loads), the CK x CN weight block in LDS read as broadcast quads.  The general gather kernel spent 0.43 / 0.20 ms
// on the two 512 x 512 x 12 problems (251 MB in, 251 MB out: 0.1 ms at HBM speed).
template <int Q>   // channel quads per side held in registers (CK, CN <= 4 Q)
__global__ void __launch_bounds__(kThreads)
pointwise_mid_k(GConv g, const float* __restrict__ wp /*[CK][CN]*/) {
  __shared__ float4 ws[4 * Q * Q];   // [k][n quad]
  __shared__ float4 bs[Q];
  const int kq = g.CK >> 2, nq = g.CN >> 2;
  for (int i = threadIdx.x; i < 4 * Q * Q; i += blockDim.x) {
    const int k = i / Q, q = i % Q;
    ws[i] = (k < g.CK && q < nq) ? *reinterpret_cast<const float4*>(wp + k * g.CN + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int i = threadIdx.x; i < Q; i += blockDim.x)
    bs[i] = (g.bias && i < nq) ? *reinterpret_cast<const float4*>(g.bias + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const long M = (long)g.N * g.DD * g.DH * g.DW;
  for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
    float4 x[Q];
    const float4* sp = reinterpret_cast<const float4*>(g.src + m * g.sld);
#pragma unroll
    for (int q = 0; q < Q; ++q) x[q] = q < kq ? sp[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = bs[q];
#pragma unroll
    for (int kk = 0; kk < Q; ++kk) {
      if (kk < kq) {   // uniform
        const float xv[4] = {x[kk].x, x[kk].y, x[kk].z, x[kk].w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int q = 0; q < Q; ++q) {
            const float4 w4 = ws[(4 * kk + e) * Q + q];
            acc[q].x = fmaf(xv[e], w4.x, acc[q].x);
            acc[q].y = fmaf(xv[e], w4.y, acc[q].y);
            acc[q].z = fmaf(xv[e], w4.z, acc[q].z);
            acc[q].w = fmaf(xv[e], w4.w, acc[q].w);
          }
      }
    }
    float4* dp = reinterpret_cast<float4*>(g.dst + m * g.dld);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      if (q < nq) {
        float4 v = acc[q];
        if (g.accumulate) {
          const float4 o = dp[q];
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        dp[q] = v;
      }
    }
  }
}

// The same problem on DENSE tensors (sld == CK, dld == CN: the 20-class head of the MRI model): the voxel records travel through
// an LDS tile with whole-line accesses (msk_tile_load / msk_tile_store) -- one thread per voxel straight from HBM moved 0.5 GB
// in 0.345 ms (1.46 TB/s: a lane's 80-byte record makes every load instruction touch 40 lines).
template <int Q>
__global__ void __launch_bounds__(kThreads)
pointwise_mid_staged_k(GConv g, const float* __restrict__ wp /*[CK][CN]*/) {
  constexpr int P = Q | 1;
  __shared__ float4 ws[4 * Q * Q];   // [k][n quad]
  __shared__ float4 bs[Q];
  __shared__ float4 tile[kThreads * P];
  const int kq = g.CK >> 2, nq = g.CN >> 2;
  for (int i = threadIdx.x; i < 4 * Q * Q; i += blockDim.x) {
    const int k = i / Q, q = i % Q;
    ws[i] = (k < g.CK && q < nq) ? *reinterpret_cast<const float4*>(wp + k * g.CN + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int i = threadIdx.x; i < Q; i += blockDim.x)
    bs[i] = (g.bias && i < nq) ? *reinterpret_cast<const float4*>(g.bias + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
  const long M = (long)g.N * g.DD * g.DH * g.DW;
  const long tiles = (M + kThreads - 1) / kThreads;
  for (long tl = blockIdx.x; tl < tiles; tl += gridDim.x) {
    const long v0 = tl * kThreads;
    const int nv = (int)(M - v0 < kThreads ? M - v0 : kThreads);
    __syncthreads();   // the previous tile's store pass is done with the LDS tile (first pass: the weights are in place)
    msk_tile_load(g.src + v0 * g.CK, nv * kq, kq, P, tile);
    __syncthreads();
    float4 x[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) x[q] = (q < kq && (int)threadIdx.x < nv) ? tile[threadIdx.x * P + q] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = bs[q];
#pragma unroll
    for (int kk = 0; kk < Q; ++kk) {
      if (kk < kq) {   // uniform
        const float xv[4] = {x[kk].x, x[kk].y, x[kk].z, x[kk].w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int q = 0; q < Q; ++q) {
            const float4 w4 = ws[(4 * kk + e) * Q + q];
            acc[q].x = fmaf(xv[e], w4.x, acc[q].x);
            acc[q].y = fmaf(xv[e], w4.y, acc[q].y);
            acc[q].z = fmaf(xv[e], w4.z, acc[q].z);
            acc[q].w = fmaf(xv[e], w4.w, acc[q].w);
          }
      }
    }
    __syncthreads();   // every thread has taken its record
#pragma unroll
    for (int q = 0; q < Q; ++q)
      if (q < nq) tile[threadIdx.x * P + q] = acc[q];
    __syncthreads();
    msk_tile_store(g.dst + v0 * g.CN, nv * nq, nq, P, tile, g.accumulate != 0);
  }
}

// 1x1x1 convolution between a THICK side (8 .. 32 channels, a multiple of 4: 16-byte accesses) and a THIN side (<= 4 channels):
// the segmentation head of the builder-defined UNet3D (32 -> ncls and its data gradient ncls -> 32).  HBM streaming; the general
// gather kernel spent 0.38 / 0.48 ms on the two 2 x 192 x 192 x 64 problems (604 MB on the thick side: 0.12 ms at HBM speed).
// TQ > 0: TQ (= thick / 4, a power of two) adjacent lanes share a voxel, each owns one 16-byte quad of the thick side -- a
// wavefront's accesses are contiguous (one thread per voxel walked its 128-byte row alone: 64 lines per instruction, 1.9 TB/s);
// thin outputs are summed over the TQ lanes with shuffles.  TQ == 0: any thick width, one thread per voxel.
template <bool THIN_OUT, int TQ>
__global__ void __launch_bounds__(kThreads)
pointwise_thin_k(GConv g, const float* __restrict__ wp /*[CK][CN]*/) {
  __shared__ __attribute__((aligned(16))) float ws[32 * 4];
  __shared__ __attribute__((aligned(16))) float bs[32];
  const int thick = THIN_OUT ? g.CK : g.CN, thin = THIN_OUT ? g.CN : g.CK;
  for (int i = threadIdx.x; i < 128; i += blockDim.x) {
    // ws[t * 4 + j]: thick channel t, thin channel j
    const int t = i >> 2, j = i & 3;
    const bool ok = t < thick && j < thin;
    ws[i] = ok ? (THIN_OUT ? wp[t * g.CN + j] : wp[j * g.CN + t]) : 0.f;
  }
  for (int i = threadIdx.x; i < 32; i += blockDim.x) bs[i] = (g.bias && i < g.CN) ? g.bias[i] : 0.f;
  __syncthreads();
  const long M = (long)g.N * g.DD * g.DH * g.DW;
  if (TQ > 0) {
    const int q = threadIdx.x % TQ;
    float4 wq[4];   // this lane's quad: weights of thick channels 4 q .. 4 q + 3 against the (<= 4) thin channels
#pragma unroll
    for (int e = 0; e < 4; ++e) wq[e] = *reinterpret_cast<const float4*>(&ws[(4 * q + e) * 4]);
    const long total = M * TQ, stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i - q < total; i += stride) {   // the TQ lanes of a voxel stay together
      const long m = i / TQ;
      const bool live = m < M;
      if (THIN_OUT) {
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) x = reinterpret_cast<const float4*>(g.src + m * g.sld)[q];
        float acc[4];
        acc[0] = fmaf(x.w, wq[3].x, fmaf(x.z, wq[2].x, fmaf(x.y, wq[1].x, x.x * wq[0].x)));
        acc[1] = fmaf(x.w, wq[3].y, fmaf(x.z, wq[2].y, fmaf(x.y, wq[1].y, x.x * wq[0].y)));
        acc[2] = fmaf(x.w, wq[3].z, fmaf(x.z, wq[2].z, fmaf(x.y, wq[1].z, x.x * wq[0].z)));
        acc[3] = fmaf(x.w, wq[3].w, fmaf(x.z, wq[2].w, fmaf(x.y, wq[1].w, x.x * wq[0].w)));
#pragma unroll
        for (int o = 1; o < TQ; o <<= 1)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] += __shfl_xor(acc[j], o, 64);
        if (live && q == 0) {
          float* dp = g.dst + m * g.dld;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j < thin) {
              const float v = acc[j] + bs[j];
              dp[j] = g.accumulate ? dp[j] + v : v;
            }
        }
      } else if (live) {
        const float* sp = g.src + m * g.sld;
        float xv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = j < thin ? sp[j] : 0.f;
        const float4 b4 = *reinterpret_cast<const float4*>(&bs[4 * q]);
        float4 v;
        v.x = fmaf(xv[3], wq[0].w, fmaf(xv[2], wq[0].z, fmaf(xv[1], wq[0].y, fmaf(xv[0], wq[0].x, b4.x))));
        v.y = fmaf(xv[3], wq[1].w, fmaf(xv[2], wq[1].z, fmaf(xv[1], wq[1].y, fmaf(xv[0], wq[1].x, b4.y))));
        v.z = fmaf(xv[3], wq[2].w, fmaf(xv[2], wq[2].z, fmaf(xv[1], wq[2].y, fmaf(xv[0], wq[2].x, b4.z))));
        v.w = fmaf(xv[3], wq[3].w, fmaf(xv[2], wq[3].z, fmaf(xv[1], wq[3].y, fmaf(xv[0], wq[3].x, b4.w))));
        float4* dp = reinterpret_cast<float4*>(g.dst + m * g.dld) + q;
        if (g.accumulate) {
          const float4 old = *dp;
          v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
        }
        *dp = v;
      }
    }
    return;
  }
  const int tq = thick >> 2;
  for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
    if (THIN_OUT) {
      const float4* sp = reinterpret_cast<const float4*>(g.src + m * g.sld);
      float acc[4] = {bs[0], bs[1], bs[2], bs[3]};
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (q < tq) {   // uniform
          const float4 x = sp[q];
          const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float4 w4 = *reinterpret_cast<const float4*>(&ws[(4 * q + e) * 4]);
            acc[0] = fmaf(xv[e], w4.x, acc[0]);
            acc[1] = fmaf(xv[e], w4.y, acc[1]);
            acc[2] = fmaf(xv[e], w4.z, acc[2]);
            acc[3] = fmaf(xv[e], w4.w, acc[3]);
          }
        }
      }
      float* dp = g.dst + m * g.dld;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < thin) dp[j] = g.accumulate ? dp[j] + acc[j] : acc[j];
    } else {
      const float* sp = g.src + m * g.sld;
      float xv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) xv[j] = j < thin ? sp[j] : 0.f;
      float4* dp = reinterpret_cast<float4*>(g.dst + m * g.dld);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (q < tq) {   // uniform
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float4 w4 = *reinterpret_cast<const float4*>(&ws[(4 * q + e) * 4]);
            o[e] = fmaf(xv[3], w4.w, fmaf(xv[2], w4.z, fmaf(xv[1], w4.y, fmaf(xv[0], w4.x, bs[4 * q + e]))));
          }
          float4 v = make_float4(o[0], o[1], o[2], o[3]);
          if (g.accumulate) {
            const float4 old = dp[q];
            v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
          }
          dp[q] = v;
        }
      }
    }
  }
}

// Weight gradient of the 1x1x1 convolution with few channels (out_tr.conv2, ncls -> ncls, vnet.py:169):
// dW[cb][ca] = sum_v dy[v][cb] * x[v][ca] -- two streams of 12-16 B per voxel and CA x CB <= 16 running sums per thread
// (the generic split-K MFMA kernel spent 0.20 ms on 100 MB; this is one pass at HBM speed).
__global__ void __launch_bounds__(kThreads)
wgrad_pw_small_k(WGrad g, long M, float* __restrict__ partial /*[grid][CA][CB]*/) {
  float acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
  for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
    float xa[4], yb[4];
    const float* xp = g.A + m * g.ald;
    const float* yp = g.B + m * g.bld;
#pragma unroll
    for (int a = 0; a < 4; ++a) xa[a] = a < g.CA ? xp[a] : 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) yb[b] = b < g.CB ? yp[b] : 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(xa[a], yb[b], acc[a][b]);
  }
  __shared__ float sh[kThreads / 64][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      float v = acc[a][b];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);  // fixed tree: reproducible
      if (lane == 0) sh[wave][a * 4 + b] = v;
    }
  __syncthreads();
  if (threadIdx.x < 16) {
    const int a = threadIdx.x >> 2, b = threadIdx.x & 3;
    if (a < g.CA && b < g.CB) {
      float v = 0.f;
      for (int w = 0; w < kThreads / 64; ++w) v += sh[w][threadIdx.x];
      partial[((long)blockIdx.x * g.CA + a) * g.CB + b] = v;
    }
  }
}

inline int grid_for(long total, int num_cu) {
  long b = (total + kThreads - 1) / kThreads;
  long cap = (long)num_cu * 32;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

int check_conv_shapes(msk_ctx* ctx, const msk_conv_desc& cd, const msk_tensor& in, const msk_tensor& out,
                      bool transposed_op) {
  MSK_REQUIRE(ctx, cd.kd > 0 && cd.kh > 0 && cd.kw > 0 && cd.sd > 0 && cd.sh > 0 && cd.sw > 0, "bad kernel/stride");
  MSK_REQUIRE(ctx, in.n == out.n, "batch mismatch");
  if (!transposed_op) {
    MSK_REQUIRE(ctx,
                out.d == (in.d + 2 * cd.pd - cd.kd) / cd.sd + 1 && out.h == (in.h + 2 * cd.ph - cd.kh) / cd.sh + 1 &&
                    out.w == (in.w + 2 * cd.pw - cd.kw) / cd.sw + 1,
                "Conv3D output dims must be floor((in+2p-k)/s)+1");
  } else {
    MSK_REQUIRE(ctx, cd.pd == 0 && cd.ph == 0 && cd.pw == 0, "Conv3DTranspose supports padding 0 only");
    MSK_REQUIRE(ctx,
                out.d == (in.d - 1) * cd.sd + cd.kd && out.h == (in.h - 1) * cd.sh + cd.kh &&
                    out.w == (in.w - 1) * cd.sw + cd.kw,
                "Conv3DTranspose output dims must be (in-1)*s+k");
  }
  return 0;
}

// ---- channel-padding wrapper of the 16-bit Winograd pipeline ---------------------------------------------------------
// 'Same' 5^3 convolutions whose channel counts are not multiples of 32 (out_tr.conv1 of a 20-class model: 32 -> 20 and its
// data gradient 20 -> 32, vnet.py:165 with the MRI config) ran on the fp32 Winograd kernels (2.97 + 5.3 ms at 512x512x12).
// Here the deficient side is padded to 32 with ZEROS -- a zero-filled copy of the source tensor and/or a 32-channel
// scratch destination, weights padded with zero rows/columns -- the pipeline runs on the padded problem, and a copy-out pass
// adds the bias.  Exact: the padding contributes zero products.  (The weight gradient keeps its kernel: it runs on the
// side stream, which has no third scratch.)
__global__ void __launch_bounds__(kThreads)
pad_weights_k(const float* __restrict__ w, int A, int B, int Ap, int Bp, int taps, float* __restrict__ out) {
  const long total = (long)Ap * Bp * taps;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % taps);
    const long r = i / taps;
    const int b = (int)(r % Bp), a = (int)(r / Bp);
    out[i] = (a < A && b < B) ? w[((long)a * B + b) * taps + t] : 0.f;
  }
}
// dst[v][0..Cp) = src[v][0..C) then zeros (float4 granules: C % 4 == 0)
__global__ void __launch_bounds__(kThreads)
pad_channels_k(const float* __restrict__ src, int sld, int C, float* __restrict__ dst, int Cp, long voxels) {
  const int q = Cp >> 2;
  const long total = voxels * q;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long v = i / q;
    const int c = (int)(i - v * q) * 4;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) x = *reinterpret_cast<const float4*>(src + v * sld + c);
    *reinterpret_cast<float4*>(dst + v * Cp + c) = x;
  }
}
// dst[v][c] = (accumulate ? dst : 0) + tmp[v][c] + bias[c]   for c < C
__global__ void __launch_bounds__(kThreads)
unpad_channels_k(const float* __restrict__ tmp, int Cp, float* __restrict__ dst, int dld, int C, const float* __restrict__ bias,
                 int accumulate, long voxels) {
  const int q = C >> 2;
  const long total = voxels * q;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long v = i / q;
    const int c = (int)(i - v * q) * 4;
    float4 x = *reinterpret_cast<const float4*>(tmp + v * Cp + c);
    if (bias) {
      x.x += bias[c]; x.y += bias[c + 1]; x.z += bias[c + 2]; x.w += bias[c + 3];
    }
    float4* o = reinterpret_cast<float4*>(dst + v * dld + c);
    if (accumulate) {
      const float4 p = *o;
      x.x += p.x; x.y += p.y; x.z += p.z; x.w += p.w;
    }
    *o = x;
  }
}

// returns 1 when handled, 0 when not eligible (nothing launched), < 0 on error
int gconv_wbf_padded(msk_ctx* ctx, const GConv& g, const float* w, int A, int B, int swap) {
  if (ctx->conv_split != 2 || ctx->conv_impl != 0 || !ctx->wbf || ctx->no_winograd) return 0;
  const bool k5 = g.kd == 5 && g.kh == 5 && g.kw == 5 && g.pd == 2 && g.ph == 2 && g.pw == 2;
  const bool k3 = g.kd == 3 && g.kh == 3 && g.kw == 3 && g.pd == 1 && g.ph == 1 && g.pw == 1;  // the deep-supervision heads
  if (!(k5 || k3) || !(g.sd == 1 && g.sh == 1 && g.sw == 1)) return 0;
  const int taps = k5 ? 125 : 27;
  if (!(g.SD == g.DD && g.SH == g.DH && g.SW == g.DW)) return 0;
  if (g.prelu || g.xform || g.fuse) return 0;  // (g.stats: not taken here -- msk_conv3d_fwd_ex then runs msk_bn_stats on y)
  const int CKp = (g.CK + 31) / 32 * 32, CNp = (g.CN + 31) / 32 * 32;
  if (CKp == g.CK && CNp == g.CN) return 0;
  if (g.CK < 16 || g.CN < 16 || g.CK % 4 || g.CN % 4) return 0;  // at most half of a side is padding; float4 granules
  if (g.sld % 4 || g.dld % 4 || (((uintptr_t)g.src) & 15) || (((uintptr_t)g.dst) & 15)) return 0;
  const long voxels = (long)g.N * g.DD * g.DH * g.DW;
  if (voxels < ctx->wbf_pad_min_voxels) return 0;  // small problems: the two extra passes cost more than the kernels differ
  const int Ap = (A + 31) / 32 * 32, Bp = (B + 31) / 32 * 32;
  const size_t wb = ((size_t)Ap * Bp * taps * sizeof(float) + 255) & ~(size_t)255;
  // round 4: a narrower SOURCE needs no padded copy -- the input transform reads the ck_real channels that exist and takes the rest
  // as zeros (WbfTinArgs::c_real); only a narrower destination still goes through a 32-channel scratch
  const size_t sb = 0;
  const size_t db = CNp != g.CN ? (((size_t)voxels * CNp * sizeof(float) + 255) & ~(size_t)255) : 0;
  // the eligibility test needs the final pointers (alignment): reserve first -- grow-only, kept for the next call
  char* ws = (char*)msk_workspace3(ctx, wb + sb + db);
  if (!ws) return -1;
  float* wpad = (float*)ws;
  float* tsrc = sb ? (float*)(ws + wb) : nullptr;
  float* tdst = db ? (float*)(ws + wb + sb) : nullptr;
  GConv gp = g;
  if (tsrc) { gp.src = tsrc; gp.sld = CKp; }
  if (tdst) { gp.dst = tdst; gp.dld = CNp; gp.accumulate = 0; gp.bias = nullptr; }
  if (CKp != g.CK) gp.ck_real = g.CK;
  gp.CK = CKp; gp.CN = CNp;
  gp.stats = nullptr;
  gp.w_persistent = false;  // wpad is scratch, rewritten per call
  if (!msk_gconv_wino_bf3_accepts(ctx, gp)) return 0;
  {
    msk_launch_scope ls(ctx, "pad_weights");
    hipLaunchKernelGGL(pad_weights_k, dim3(grid_for((long)Ap * Bp * taps, ctx->num_cu)), dim3(kThreads), 0, ctx->stream, w, A, B, Ap, Bp,
                       taps, wpad);
    MSK_LAUNCH_CHECK(ctx);
  }
  if (tsrc) {
    msk_launch_scope ls(ctx, "pad_channels");
    hipLaunchKernelGGL(pad_channels_k, dim3(grid_for(voxels * (CKp / 4), ctx->num_cu)), dim3(kThreads), 0, ctx->stream, g.src, g.sld, g.CK,
                       tsrc, CKp, voxels);
    MSK_LAUNCH_CHECK(ctx);
  }
  const int r = msk_gconv_wino_bf3(ctx, gp, wpad, Ap, Bp, swap);
  if (r < 0) return r;
  if (r == 0) return msk_fail(ctx, __FILE__, __LINE__, "gconv_wbf_padded", "the pipeline declined a problem it accepted");
  if (tdst) {
    msk_launch_scope ls(ctx, "unpad_channels");
    hipLaunchKernelGGL(unpad_channels_k, dim3(grid_for(voxels * (g.CN / 4), ctx->num_cu)), dim3(kThreads), 0, ctx->stream, tdst, CNp, g.dst,
                       g.dld, g.CN, g.bias, g.accumulate, voxels);
    MSK_LAUNCH_CHECK(ctx);
  }
  return 1;
}

// dw[cb][ca][t] (+)= tmp[cb][ca][t] out of the padded [CBp][CAp][taps] gradient
__global__ void __launch_bounds__(kThreads)
unpad_dw_k(const float* __restrict__ tmp, int CAp, float* __restrict__ dw, int CA, int CB, int taps, int accumulate) {
  const long total = (long)CB * CA * taps;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % taps);
    const long r = i / taps;
    const int ca = (int)(r % CA), cb = (int)(r / CA);
    const float v = tmp[((long)cb * CAp + ca) * taps + t];
    dw[i] = accumulate ? dw[i] + v : v;
  }
}

// The weight-gradient side of gconv_wbf_padded: x and/or dy zero-padded to 32-channel multiples, the padded gradient
// [CBp][CAp][125] in scratch, the valid block copied (or added) out.  Runs wherever the caller runs (the weight-gradient side
// stream has its own third scratch).  1 = handled, 0 = not eligible, < 0 error.
int wgrad_wbf_padded(msk_ctx* ctx, const WGrad& g) {
  if (ctx->conv_split != 2 || ctx->conv_impl != 0 || !ctx->wbf || ctx->no_winograd) return 0;
  const bool k5 = g.kd == 5 && g.kh == 5 && g.kw == 5 && g.pd == 2 && g.ph == 2 && g.pw == 2;
  const bool k3 = g.kd == 3 && g.kh == 3 && g.kw == 3 && g.pd == 1 && g.ph == 1 && g.pw == 1;
  if (!(k5 || k3) || !(g.sd == 1 && g.sh == 1 && g.sw == 1)) return 0;
  const int taps = k5 ? 125 : 27;
  if (!(g.AD == g.BD && g.AH == g.BH && g.AW == g.BW)) return 0;
  if (g.xform || g.yform || g.yfuse) return 0;
  const int CAp = (g.CA + 31) / 32 * 32, CBp = (g.CB + 31) / 32 * 32;
  if (CAp == g.CA && CBp == g.CB) return 0;
  if (g.CA < 16 || g.CB < 16 || g.CA % 4 || g.CB % 4) return 0;
  if (g.ald % 4 || g.bld % 4 || (((uintptr_t)g.A) & 15) || (((uintptr_t)g.B) & 15)) return 0;
  const long voxels = (long)g.N * g.BD * g.BH * g.BW;
  if (voxels < ctx->wbf_pad_min_voxels) return 0;
  const size_t wb = ((size_t)CAp * CBp * taps * sizeof(float) + 255) & ~(size_t)255;
  const size_t ab = CAp != g.CA ? (((size_t)voxels * CAp * sizeof(float) + 255) & ~(size_t)255) : 0;
  const size_t bb = 0;   // round 4: a narrower dy needs no padded copy (WGrad::cb_real: the A dy transform takes the missing channels as zeros)
  char* ws = (char*)msk_workspace3(ctx, wb + ab + bb);
  if (!ws) return -1;
  float* dwp = (float*)ws;
  float* ta = ab ? (float*)(ws + wb) : nullptr;
  float* tb = bb ? (float*)(ws + wb + ab) : nullptr;
  WGrad gp = g;
  if (ta) { gp.A = ta; gp.ald = CAp; }
  if (tb) { gp.B = tb; gp.bld = CBp; }
  if (CBp != g.CB) gp.cb_real = g.CB;
  gp.CA = CAp; gp.CB = CBp; gp.dw = dwp; gp.accumulate = 0;   // (a_amax / b_amax stay: zero padding does not change a maximum)
  if (!msk_wgrad_wbf_accepts(ctx, gp)) return 0;
  if (ta) {
    msk_launch_scope ls(ctx, "pad_channels");
    hipLaunchKernelGGL(pad_channels_k, dim3(grid_for(voxels * (CAp / 4), ctx->num_cu)), dim3(kThreads), 0, ctx->stream, g.A, g.ald, g.CA, ta,
                       CAp, voxels);
    MSK_LAUNCH_CHECK(ctx);
  }
  if (tb) {
    msk_launch_scope ls(ctx, "pad_channels");
    hipLaunchKernelGGL(pad_channels_k, dim3(grid_for(voxels * (CBp / 4), ctx->num_cu)), dim3(kThreads), 0, ctx->stream, g.B, g.bld, g.CB, tb,
                       CBp, voxels);
    MSK_LAUNCH_CHECK(ctx);
  }
  const int r = msk_wgrad_wbf(ctx, gp);
  if (r < 0) return r;
  if (r == 0) return 0;  // (size limits inside the pipeline: the caller's other kernels take the original problem)
  msk_launch_scope ls(ctx, "unpad_dw");
  hipLaunchKernelGGL(unpad_dw_k, dim3(grid_for((long)g.CA * g.CB * taps, ctx->num_cu)), dim3(kThreads), 0, ctx->stream, dwp, CAp, g.dw, g.CA,
                     g.CB, taps, g.accumulate);
  MSK_LAUNCH_CHECK(ctx);
  return 1;
}

// Run a gather convolution.  w is canonical w[A][B][taps]; swap selects (k,n) = (b,a).
int run_gconv_dispatch(msk_ctx* ctx, GConv g, const float* w, int A, int B, int swap, const char* tag, bool* act_fused) {
  const int taps = g.kd * g.kh * g.kw;
  if (ctx->conv_impl != 1 && ctx->conv_impl != 4 && taps == 1 && g.sd == 1 && g.sh == 1 && g.sw == 1 && g.pd == 0 &&
      g.ph == 0 && g.pw == 0 && g.CK <= 8 && g.CN <= 8) {
    const float* wp1 = msk_pack_weights_get(ctx, w, A, B, 1, swap, 0, 1, 1, 1, 0, g.CK, g.CN, 0, 0);
    if (!wp1) return -1;
    const long M = (long)g.N * g.DD * g.DH * g.DW;
    msk_launch_scope ls(ctx, "pointwise_small");
    hipLaunchKernelGGL(pointwise_small_k, dim3(grid_for(M, ctx->num_cu)), dim3(kThreads), 0, ctx->stream, g, (const float*)wp1);
    MSK_LAUNCH_CHECK(ctx);
    return 0;
  }
  if (ctx->conv_impl != 1 && ctx->conv_impl != 4 && ctx->conv_impl != 6 && taps == 1 && g.sd == 1 && g.sh == 1 && g.sw == 1 && g.pd == 0 && g.ph == 0 &&
      g.pw == 0 && ((g.CN <= 4 && g.CK > 8 && g.CK <= 32 && g.CK % 4 == 0 && g.sld % 4 == 0 && (((uintptr_t)g.src) & 15) == 0) ||
                    (g.CK <= 4 && g.CN > 8 && g.CN <= 32 && g.CN % 4 == 0 && g.dld % 4 == 0 && (((uintptr_t)g.dst) & 15) == 0))) {
    // thick <-> thin 1x1x1 (the 32 -> ncls head of UNet3D and its data gradient): streaming VALU kernel; 6 = A/B: the general gather kernel
    const float* wp1 = msk_pack_weights_get(ctx, w, A, B, 1, swap, 0, 1, 1, 1, 0, g.CK, g.CN, 0, 0);
    if (!wp1) return -1;
    const long M = (long)g.N * g.DD * g.DH * g.DW;
    msk_launch_scope ls(ctx, "pointwise_thin");
    const int thick = g.CN <= 4 ? g.CK : g.CN;
    const int TQ = thick == 32 ? 8 : (thick == 16 ? 4 : 0);
    const dim3 grid(grid_for(M * (TQ ? TQ : 1), ctx->num_cu));
#define PW_THIN(out_, tq_) hipLaunchKernelGGL((pointwise_thin_k<out_, tq_>), grid, dim3(kThreads), 0, ctx->stream, g, (const float*)wp1)
    if (g.CN <= 4) {
      if (TQ == 8) PW_THIN(true, 8); else if (TQ == 4) PW_THIN(true, 4); else PW_THIN(true, 0);
    } else {
      if (TQ == 8) PW_THIN(false, 8); else if (TQ == 4) PW_THIN(false, 4); else PW_THIN(false, 0);
    }
#undef PW_THIN
    MSK_LAUNCH_CHECK(ctx);
    return 0;
  }
  if (ctx->conv_impl != 1 && ctx->conv_impl != 4 && ctx->conv_impl != 6 && taps == 1 && g.sd == 1 && g.sh == 1 && g.sw == 1 && g.pd == 0 && g.ph == 0 &&
      g.pw == 0 && g.CK <= 32 && g.CN <= 32 && g.CK % 4 == 0 && g.CN % 4 == 0 && g.sld % 4 == 0 && g.dld % 4 == 0 &&
      (((uintptr_t)g.src) & 15) == 0 && (((uintptr_t)g.dst) & 15) == 0 && (!g.bias || (((uintptr_t)g.bias) & 15) == 0)) {
    // 1x1x1 with 12 .. 32 channels (the 20-class heads): streaming VALU kernel; 6 = A/B: the general gather kernel
    const float* wp1 = msk_pack_weights_get(ctx, w, A, B, 1, swap, 0, 1, 1, 1, 0, g.CK, g.CN, 0, 0);
    if (!wp1) return -1;
    const long M = (long)g.N * g.DD * g.DH * g.DW;
    msk_launch_scope ls(ctx, "pointwise_mid");
    const int cmax = g.CK > g.CN ? g.CK : g.CN;
    if (g.sld == g.CK && g.dld == g.CN && (ctx->tile_staging & 1)) {   // dense on both sides: records through an LDS tile
      const long tiles = (M + kThreads - 1) / kThreads;
      const unsigned gb = (unsigned)(tiles < 8L * ctx->num_cu ? tiles : 8L * ctx->num_cu);
      if (cmax <= 16) hipLaunchKernelGGL((pointwise_mid_staged_k<4>), dim3(gb), dim3(kThreads), 0, ctx->stream, g, (const float*)wp1);
      else if (cmax <= 24) hipLaunchKernelGGL((pointwise_mid_staged_k<6>), dim3(gb), dim3(kThreads), 0, ctx->stream, g, (const float*)wp1);
      else hipLaunchKernelGGL((pointwise_mid_staged_k<8>), dim3(gb), dim3(kThreads), 0, ctx->stream, g, (const float*)wp1);
    } else
    if (cmax <= 16) hipLaunchKernelGGL((pointwise_mid_k<4>), dim3(grid_for(M, ctx->num_cu)), dim3(kThreads), 0, ctx->stream, g, (const float*)wp1);
    else if (cmax <= 24) hipLaunchKernelGGL((pointwise_mid_k<6>), dim3(grid_for(M, ctx->num_cu)), dim3(kThreads), 0, ctx->stream, g, (const float*)wp1);
    else hipLaunchKernelGGL((pointwise_mid_k<8>), dim3(grid_for(M, ctx->num_cu)), dim3(kThreads), 0, ctx->stream, g, (const float*)wp1);
    MSK_LAUNCH_CHECK(ctx);
    return 0;
  }
  if (ctx->conv_impl != 1 && ctx->conv_impl != 4) {
    int r = ctx->conv_impl == 23 ? 0 : msk_gconv_c1_h2(ctx, g, w, A, B, swap);    // 27 = A/B: the fp32-MFMA kernel below
    if (r < 0) return r;
    if (r == 1) return 0;
    r = ctx->conv_impl == 23 ? 0 : msk_gconv_c1_mfma(ctx, g, w, A, B, swap);  // 23 = A/B: VALU kernel for 1 -> 16
    if (r < 0) return r;
    if (r == 1) return 0;
    r = ctx->conv_impl == 8 ? 0 : msk_gconv_tk_h2(ctx, g, w, A, B, swap);
    if (r < 0) return r;
    if (r == 1) return 0;
    r = ctx->conv_impl == 8 ? 0 : msk_gconv_halo_tightk(ctx, g, w, A, B, swap);  // 8 = A/B: skip the tight-K kernels
    if (r < 0) return r;
    if (r == 1) return 0;
    // 22 = A/B: the VALU kernels instead of the folded-column MFMA kernel; 9 skips both (one-voxel VALU kernel)
    r = (ctx->conv_impl == 22 || ctx->conv_impl == 9) ? 0 : msk_gconv_halo_foldn(ctx, g, w, A, B, swap, act_fused);
    if (r < 0) return r;
    if (r == 1) return 0;
    r = ctx->conv_impl == 9 ? 0 : msk_gconv_halo_valu2(ctx, g, w, A, B, swap);  // 9 = A/B: one-voxel VALU kernel
    if (r < 0) return r;
    if (r == 1) return 0;
    r = gconv_wbf_padded(ctx, g, w, A, B, swap);  // channel counts that are not multiples of 32, padded with zeros
    if (r < 0) return r;
    if (r == 1) return 0;
    // 20 = fp32-MFMA Winograd kernels instead of the bf16x3 pipeline (A/B); option "wino_bf3" 0 does the same
    r = (ctx->conv_impl == 10 || ctx->conv_impl == 11 || ctx->conv_impl == 14 || ctx->conv_impl == 20 || ctx->no_winograd || !ctx->wbf)
            ? 0 : msk_gconv_wino_bf3(ctx, g, w, A, B, swap);
    if (r < 0) return r;
    if (r == 1) {
      *act_fused = true;  // wbf_tout_k applies g.prelu
      return 0;
    }
    r = (ctx->conv_impl == 11 || ctx->no_winograd) ? 0 : msk_gconv_halo_wino(ctx, g, w, A, B, swap);  // 11 = direct kernel only (A/B)
    if (r < 0) return r;
    if (r == 1) {
      *act_fused = true;  // the Winograd epilogues apply g.prelu
      return 0;
    }
    r = msk_gconv_halo_mfma(ctx, g, w, A, B, swap);
    if (r < 0) return r;
    if (r == 1) return 0;
    if (ctx->conv_impl != 6) {  // 6 = skip the scatter kernel (A/B against the parity-class gather kernel)
      r = msk_gconv_scatter_mfma(ctx, g, w, A, B, swap);
      if (r < 0) return r;
      if (r == 1) return 0;
    }
    if (ctx->conv_impl != 6 && ctx->conv_impl != 7) {  // 6 / 7 = A/B: the general gather kernel
      r = msk_gconv_ks_fwd(ctx, g, w, A, B, swap);
      if (r < 0) return r;
      if (r == 1) return 0;
      r = msk_gconv_kst(ctx, g, w, A, B, swap);
      if (r < 0) return r;
      if (r == 1) return 0;
    }
    r = msk_gconv_gather_mfma(ctx, g, w, A, B, swap);
    if (r < 0) return r;
    if (r == 1) return 0;
  }
  // reference path
  g.flip = 0;
  const long wn = (long)taps * g.CK * g.CN;
  const float* wp = msk_pack_weights_get(ctx, w, A, B, taps, swap, 0, g.kd, g.kh, g.kw, 0, g.CK, g.CN, 0, 0);
  if (!wp) return -1;
  const long total = (long)g.N * g.DD * g.DH * g.DW * g.CN;
  if (ctx->prof && ctx->prof_shapes) {
    char buf[160];
    snprintf(buf, sizeof(buf), "%s[ck=%d,cn=%d,k=%dx%dx%d,dst=%dx%dx%dx%d]", tag, g.CK, g.CN, g.kd, g.kh, g.kw, g.N, g.DD, g.DH, g.DW);
    tag = msk_intern_tag(ctx, buf);
  }
  msk_launch_scope ls(ctx, tag);
  hipLaunchKernelGGL(gconv_direct_k, dim3(grid_for(total, ctx->num_cu)), dim3(kThreads), 0, ctx->stream, g, (const float*)wp);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

// y = prelu(y) in place: the activation of msk_conv3d_fwd_act behind the kernels without a fused epilogue
__global__ void __launch_bounds__(256)
prelu_inplace_k(float* __restrict__ y, int ld, int C, long voxels, const float* __restrict__ slope) {
  const long total = voxels * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long v = i / C;
    const int c = (int)(i - v * C);
    float* o = y + v * ld + c;
    const float t = *o;
    if (t < 0.f) *o = t * slope[c];
  }
}

int run_gconv_one(msk_ctx* ctx, GConv g, const float* w, int A, int B, int swap, const char* tag) {
  bool fused = false;
  if (int rc = run_gconv_dispatch(ctx, g, w, A, B, swap, tag, &fused)) return rc;
  if (g.prelu && !fused) {
    const long voxels = (long)g.N * g.DD * g.DH * g.DW;
    msk_launch_scope ls(ctx, "prelu_inplace");
    hipLaunchKernelGGL(prelu_inplace_k, dim3(grid_for(voxels * g.CN, ctx->num_cu)), dim3(kThreads), 0, ctx->stream, g.dst,
                       g.dld, g.CN, voxels, g.prelu);
    MSK_LAUNCH_CHECK(ctx);
  }
  return 0;
}

// w'[co][...] = w[co][...] * scale[co],  b'[co] = b[co] * scale[co] + shift[co]   (eval-mode BatchNorm folded into
// the convolution in front of it; canonical Conv3D weight layout [Cout][Cin * taps])
__global__ void __launch_bounds__(256)
fold_bn_k(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ scale,
          const float* __restrict__ shift, int cout, long inner, float* __restrict__ wf, float* __restrict__ bf) {
  const long total = (long)cout * inner;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    wf[i] = w[i] * scale[i / inner];
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < cout) bf[t] = (b ? b[t] : 0.f) * scale[t] + shift[t];
}

int run_wgrad_one(msk_ctx* ctx, const WGrad& g) {
  if (g.kd == 1 && g.kh == 1 && g.kw == 1 && g.sd == 1 && g.sh == 1 && g.sw == 1 && g.pd == 0 && g.ph == 0 && g.pw == 0 &&
      g.CA <= 4 && g.CB <= 4 && ctx->conv_impl != 1 && ctx->conv_impl != 3) {
    const long M = (long)g.N * g.BD * g.BH * g.BW;
    long blocks = (M + kThreads * 16 - 1) / (kThreads * 16);
    if (blocks > 4L * ctx->num_cu) blocks = 4L * ctx->num_cu;
    if (blocks < 1) blocks = 1;
    float* partial = (float*)msk_workspace(ctx, (size_t)blocks * g.CA * g.CB * sizeof(float));
    if (!partial) return -1;
    {
      msk_launch_scope ls(ctx, "wgrad_pw_small");
      hipLaunchKernelGGL(wgrad_pw_small_k, dim3((unsigned)blocks), dim3(kThreads), 0, ctx->stream, g, M, partial);
      MSK_LAUNCH_CHECK(ctx);
    }
    return msk_wgrad_reduce(ctx, partial, (int)blocks, 1, g.CA, g.CB, g.dw, g.accumulate);
  }
  if (ctx->conv_impl != 1 && ctx->conv_impl != 3 && ctx->conv_impl != 18) {  // 18 = folded gather kernels for in_tr / out_tr (A/B)
    int r = msk_wgrad_c1(ctx, g);
    if (r < 0) return r;
    if (r == 1) return 0;
    r = msk_wgrad_cbs(ctx, g);
    if (r < 0) return r;
    if (r == 1) return 0;
  }
  if (ctx->conv_impl != 1 && ctx->conv_impl != 3 && ctx->conv_impl != 16) {  // 16 = generic tap-row kernel (A/B)
    int r = msk_wgrad_ks(ctx, g);
    if (r < 0) return r;
    if (r == 1) return 0;
  }
  {
    const int r = wgrad_wbf_padded(ctx, g);  // channel counts that are not multiples of 32, padded with zeros
    if (r < 0) return r;
    if (r == 1) return 0;
  }
  // 20 / 21 = fp32-MFMA Winograd weight gradient instead of the bf16x3 kernel (A/B; 21 keeps the bf16x3 forward)
  if (ctx->conv_impl != 1 && ctx->conv_impl != 3 && ctx->conv_impl != 12 && ctx->conv_impl != 13 && ctx->conv_impl != 14 &&
      ctx->conv_impl != 20 && ctx->conv_impl != 21 && !ctx->no_winograd && ctx->wbf) {
    int r = msk_wgrad_wbf(ctx, g);
    if (r < 0) return r;
    if (r == 1) return 0;
  }
  if (ctx->conv_impl != 1 && ctx->conv_impl != 3 && ctx->conv_impl != 13 && !ctx->no_winograd) {  // 13 = direct LDS kernel only (A/B)
    int r = msk_wgrad_wino(ctx, g);
    if (r < 0) return r;
    if (r == 1) return 0;
  }
  if (ctx->conv_impl != 1 && ctx->conv_impl != 3) {
    int r = msk_wgrad_mfma(ctx, g);
    if (r < 0) return r;
    if (r == 1) return 0;
  }
  const int taps = g.kd * g.kh * g.kw;
  const long M = (long)g.N * g.BD * g.BH * g.BW;
  const int ca_tiles = (g.CA + 15) / 16, cb_tiles = (g.CB + 15) / 16;
  const long blocks = (long)taps * ca_tiles * cb_tiles;
  long splits = (2048 + blocks - 1) / blocks;
  if (splits > M / 64) splits = M / 64;
  if (splits < 1) splits = 1;
  if (splits > 4096) splits = 4096;
  const size_t pbytes = (size_t)splits * taps * g.CA * g.CB * sizeof(float);
  float* partial = (float*)msk_workspace(ctx, pbytes);
  if (!partial) return -1;
  {
    msk_launch_scope ls(ctx, "wgrad_direct");
    hipLaunchKernelGGL(wgrad_direct_k, dim3((int)blocks, (int)splits), dim3(kThreads), 0, ctx->stream, g, (int)splits, partial);
    MSK_LAUNCH_CHECK(ctx);
  }
  return msk_wgrad_reduce(ctx, partial, (int)splits, taps, g.CA, g.CB, g.dw, g.accumulate);
}

}  // namespace

int msk_pack_weights(msk_ctx* ctx, const float* w, int A, int B, int taps, int swap, int flip_taps, int kd,
                     int kh, int kw, int mfma, int K, int N, int KC, int npad, float* out) {
  const long total = mfma ? (long)taps * KC * 2 * npad * 4 : (long)taps * K * (npad > 0 ? npad : N);
  msk_launch_scope ls(ctx, mfma ? "pack_weights_mfma" : "pack_weights_direct");
  hipLaunchKernelGGL(pack_weights_k, dim3(grid_for(total, ctx->num_cu)), dim3(kThreads), 0, ctx->stream, w, A, B, taps,
                     swap, flip_taps, kd, kh, kw, mfma, K, N, KC, npad, out, total);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

// Cached form of msk_pack_weights (see SmallPackCache above): the packed image of (w, layout); option "small_pack_cache" 0 =
// pack into the shared scratch on every call (A/B, the round-4 behaviour).  nullptr on failure.
const float* msk_pack_weights_get(msk_ctx* ctx, const float* w, int A, int B, int taps, int swap, int flip_taps, int kd, int kh,
                                  int kw, int mfma, int K, int N, int KC, int npad) {
  const long total = mfma ? (long)taps * KC * 2 * npad * 4 : (long)taps * K * (npad > 0 ? npad : N);
  if (!ctx->small_pack_cache) {
    float* out = (float*)msk_workspace2(ctx, (size_t)total * sizeof(float));
    if (!out) return nullptr;
    return msk_pack_weights(ctx, w, A, B, taps, swap, flip_taps, kd, kh, kw, mfma, K, N, KC, npad, out) == 0 ? out : nullptr;
  }
  SmallPackDesc d{};
  d.w = w; d.total = total; d.kind = 0; d.A = A; d.B = B; d.taps = taps; d.swap = swap; d.flip = flip_taps;
  d.kd = kd; d.kh = kh; d.kw = kw; d.mfma = mfma; d.K = K; d.N = N; d.KC = KC; d.npad = npad;
  return small_pack_get(ctx, d, (size_t)total * sizeof(float));
}
// the scatter layout of msk_conv_scatter.hip, cached the same way (the caller packs itself when this returns nullptr with the cache off)
const float* msk_pack_scatter_get(msk_ctx* ctx, const float* w, int A, int B, int taps, int swap, int CK, int CN, int KC, int jpad) {
  if (!ctx->small_pack_cache) return nullptr;
  SmallPackDesc d{};
  d.w = w; d.total = (long)KC * 2 * jpad; d.kind = 1; d.A = A; d.B = B; d.taps = taps; d.swap = swap;
  d.K = CK; d.N = CN; d.KC = KC; d.npad = jpad;
  return small_pack_get(ctx, d, (size_t)d.total * 4 * sizeof(float));
}
void msk_small_pack_changed(msk_ctx* ctx, const void* p, size_t bytes) {
  if (!ctx->spack || !p) return;
  SmallPackCache* c = (SmallPackCache*)ctx->spack;
  const char* a0 = (const char*)p;
  const char* a1 = a0 + bytes;
  for (int r = 0; r < SmallPackCache::kRows; ++r) {
    SmallPackEntry& e = c->e[r];
    if (!e.live) continue;
    const char* b0 = (const char*)e.d.w;
    const char* b1 = b0 + (size_t)e.d.A * e.d.B * e.d.taps * sizeof(float);
    if (a0 < b1 && b0 < a1) e.valid = false;
  }
}
void msk_small_pack_freed(msk_ctx* ctx, const void* p, size_t bytes) {
  if (!ctx->spack || !p) return;
  SmallPackCache* c = (SmallPackCache*)ctx->spack;
  const char* a0 = (const char*)p;
  const char* a1 = a0 + bytes;
  for (int r = 0; r < SmallPackCache::kRows; ++r) {
    SmallPackEntry& e = c->e[r];
    if (!e.live) continue;
    const char* b0 = (const char*)e.d.w;
    const char* b1 = b0 + (size_t)e.d.A * e.d.B * e.d.taps * sizeof(float);
    if (bytes ? (a0 < b1 && b0 < a1) : b0 == a0) {   // the image stays allocated for the row's next tenant
      e.live = false;
      e.valid = false;
      e.d.w = nullptr;
    }
  }
}
// rebuild the stale rows in use (all of them, or those whose weights lie wholly inside [p, p + bytes)) in one launch on the current stream
int msk_small_prepack(msk_ctx* ctx, const void* p, size_t bytes) {
  if (!ctx->spack) return 0;
  SmallPackCache* c = (SmallPackCache*)ctx->spack;
  const char* a0 = (const char*)p;
  const char* a1 = a0 + bytes;
  std::vector<int> rows;
  for (int r = 0; r < SmallPackCache::kRows; ++r) {
    const SmallPackEntry& e = c->e[r];
    if (!e.live || e.valid || e.last_use + 4096 < c->tick) continue;   // rows nothing has asked for in a long while wait for their next use
    if (p) {
      const char* b0 = (const char*)e.d.w;
      const char* b1 = b0 + (size_t)e.d.A * e.d.B * e.d.taps * sizeof(float);
      if (!(b0 >= a0 && b1 <= a1)) continue;
    }
    rows.push_back(r);
  }
  return rows.empty() ? 0 : small_pack_build(ctx, c, rows);
}
void msk_small_pack_free(msk_ctx* ctx) {
  if (!ctx->spack) return;
  SmallPackCache* c = (SmallPackCache*)ctx->spack;
  for (int r = 0; r < SmallPackCache::kRows; ++r)
    if (c->e[r].d.out) hipFree(c->e[r].d.out);
  if (c->table) hipFree(c->table);
  delete c;
  ctx->spack = nullptr;
}

// float4 variant (CB % 4 == 0): a lane owns 4 consecutive cb of one (tap, ca) row -> 1 KiB per wavefront load,
// two slabs in flight per slice (the scalar version moved 256 B per load and reached ~1.6 TB/s)
__global__ void __launch_bounds__(kThreads)
wgrad_reduce_v4_k(const float* __restrict__ partial, int splits, int stride, long pitch, int taps, int CA, int CB,
                  float* __restrict__ dw, int accumulate, int nbias, float* __restrict__ db, int db_accumulate) {
  __shared__ double sh[4][64][4];
  const long per = (long)taps * CA * CB, per4 = (pitch >> 2) * stride;  // slab pitch in float4 (slab k at k*stride)
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const long n4 = pitch >> 2;   // the slab's tail (pitch - per floats) holds bias sums
  for (long base = (long)blockIdx.x * 64; base < n4; base += (long)gridDim.x * 64) {
    const long i4 = base + lane;
    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
    if (i4 < n4) {
      const float4* p = reinterpret_cast<const float4*>(partial) + i4;
      int k = slice;
      for (; k + 12 < splits; k += 16) {  // four slabs in flight
        const float4 u = p[(long)k * per4], v = p[(long)(k + 4) * per4];
        const float4 y = p[(long)(k + 8) * per4], z = p[(long)(k + 12) * per4];
        a[0] += u.x; a[1] += u.y; a[2] += u.z; a[3] += u.w;
        b[0] += v.x; b[1] += v.y; b[2] += v.z; b[3] += v.w;
        a[0] += y.x; a[1] += y.y; a[2] += y.z; a[3] += y.w;
        b[0] += z.x; b[1] += z.y; b[2] += z.z; b[3] += z.w;
      }
      for (; k < splits; k += 4) {
        const float4 u = p[(long)k * per4];
        a[0] += u.x; a[1] += u.y; a[2] += u.z; a[3] += u.w;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) sh[slice][lane][j] = a[j] + b[j];
    __syncthreads();
    if (slice == 0 && i4 < n4) {
      const long idx = i4 << 2;  // = (tap*CA + ca)*CB + cb
      if (idx >= per) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const long b = idx - per + j;
          if (b < nbias) {
            const double sum = (sh[0][lane][j] + sh[1][lane][j]) + (sh[2][lane][j] + sh[3][lane][j]);
            db[b] = db_accumulate ? db[b] + (float)sum : (float)sum;
          }
        }
      } else {
        const int cb = (int)(idx % CB);
        const long r = idx / CB;
        const int ca = (int)(r % CA);
        const int tap = (int)(r / CA);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const double sum = (sh[0][lane][j] + sh[1][lane][j]) + (sh[2][lane][j] + sh[3][lane][j]);
          float* o = dw + ((long)(cb + j) * CA + ca) * taps + tap;
          *o = accumulate ? *o + (float)sum : (float)sum;
        }
      }
    }
    __syncthreads();
  }
}

// First stage for "few outputs, thousands of slabs" (in_tr / out_tr / the 2x2x2 convs: 2 K - 12 K outputs, up to
// 12 288 split slabs): block row y adds the slabs [y*chunk, (y+1)*chunk) in a fixed order and leaves the sum in the
// FIRST slab of its range (only slice 0 ever reads that slab, and it writes after its own reads).  The final kernels
// then walk splits/chunk slabs with a slab stride of `chunk`.  (Alone, the final kernel had 8 blocks walking 12 288
// slabs for in_tr: 0.36 ms.)
__global__ void __launch_bounds__(kThreads)
wgrad_prereduce_k(float* __restrict__ partial, int splits, int chunk, long per) {
  __shared__ double sh[3][64];
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int k0 = blockIdx.y * chunk;
  const int k1 = min(splits, k0 + chunk);
  for (long base = (long)blockIdx.x * 64; base < per; base += (long)gridDim.x * 64) {
    const long i = base + lane;
    double a = 0.0, b = 0.0;
    if (i < per) {
      const float* p = partial + i;
      int k = k0 + slice;
      for (; k + 4 < k1; k += 8) {
        a += (double)p[(long)k * per];
        b += (double)p[(long)(k + 4) * per];
      }
      if (k < k1) a += (double)p[(long)k * per];
    }
    if (slice > 0) sh[slice - 1][lane] = a + b;
    __syncthreads();
    if (slice == 0 && i < per) partial[(long)k0 * per + i] = (float)(((a + b) + sh[0][lane]) + (sh[1][lane] + sh[2][lane]));
    __syncthreads();
  }
}

int msk_wgrad_reduce(msk_ctx* ctx, const float* partial, int splits, int taps, int CA, int CB, float* dw,
                     int accumulate) {
  return msk_wgrad_reduce_ex(ctx, partial, splits, (long)taps * CA * CB, taps, CA, CB, dw, accumulate, 0, nullptr, 0);
}

int msk_wgrad_reduce_ex(msk_ctx* ctx, const float* partial, int splits, long pitch, int taps, int CA, int CB, float* dw,
                        int accumulate, int nbias, float* db, int db_accumulate) {
  const long per = pitch;   // floats every stage walks per slab (the bias tail included)
  MSK_REQUIRE(ctx, pitch == (long)taps * CA * CB || (CB % 4 == 0 && pitch % 4 == 0 && (((uintptr_t)partial) & 15) == 0),
              "msk_wgrad_reduce_ex: a bias tail needs the float4 reduction");
  msk_launch_scope ls(ctx, "wgrad_reduce");
  int stride = 1;
  {
    const long eb = (per + 63) / 64;                   // element blocks of the final kernels
    if (splits >= 256 && eb <= 2L * ctx->num_cu) {      // too few element blocks to keep the chip busy on their own
      long rows = 8L * ctx->num_cu / eb;                // block rows wanted
      if (rows > splits / 16) rows = splits / 16;       // at least 16 slabs per row
      const int chunk = (int)((splits + rows - 1) / rows);
      const int nrows = (splits + chunk - 1) / chunk;
      hipLaunchKernelGGL(wgrad_prereduce_k, dim3((unsigned)eb, (unsigned)nrows), dim3(kThreads), 0, ctx->stream,
                         const_cast<float*>(partial), splits, chunk, per);
      MSK_LAUNCH_CHECK(ctx);
      stride = chunk;
      splits = nrows;
    }
  }
  if (CB % 4 == 0 && (((uintptr_t)partial) & 15) == 0) {
    long rb = (per / 4 + 63) / 64;
    if (rb > (long)ctx->num_cu * 32) rb = (long)ctx->num_cu * 32;
    hipLaunchKernelGGL(wgrad_reduce_v4_k, dim3((unsigned)rb), dim3(kThreads), 0, ctx->stream, partial, splits, stride, pitch, taps,
                       CA, CB, dw, accumulate, nbias, db, db_accumulate);
    MSK_LAUNCH_CHECK(ctx);
    return 0;
  }
  long rblocks = (per + 63) / 64;
  if (rblocks > (long)ctx->num_cu * 32) rblocks = (long)ctx->num_cu * 32;
  hipLaunchKernelGGL(wgrad_reduce_k, dim3((unsigned)rblocks), dim3(kThreads), 0, ctx->stream, partial, splits, stride,
                     taps, CA, CB, dw, accumulate);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

// Several kernels address a tensor with 32-bit byte offsets (raw buffer loads), i.e. need every tensor below
// 4 GiB.  288 GB of HBM invites batches beyond that (32ch @ 128^3 crosses it at N = 17), so the batch is cut
// into chunks of whole samples here: forward / data-gradient chunks are independent, weight-gradient chunks
// accumulate in order (deterministic).
constexpr size_t kChunkBytes = (size_t)3900 << 20;

int run_gconv(msk_ctx* ctx, GConv g, const float* w, int A, int B, int swap, const char* tag) {
  const size_t sper = (size_t)g.SD * g.SH * g.SW * g.sld * sizeof(float), dper = (size_t)g.DD * g.DH * g.DW * g.dld * sizeof(float);
  const size_t per = sper > dper ? sper : dper;
  long nmax = per > 0 ? (long)(kChunkBytes / per) : g.N;
  if (nmax < 1) nmax = 1;
  if (g.N <= nmax) return run_gconv_one(ctx, g, w, A, B, swap, tag);
  const int total = g.N;
  for (int n0 = 0; n0 < total; n0 += (int)nmax) {
    GConv c = g;
    c.N = total - n0 < nmax ? total - n0 : (int)nmax;
    c.src = g.src + (size_t)n0 * (sper / sizeof(float));
    c.dst = g.dst + (size_t)n0 * (dper / sizeof(float));
    if (int rc = run_gconv_one(ctx, c, w, A, B, swap, tag)) return rc;
  }
  return 0;
}

int run_wgrad(msk_ctx* ctx, const WGrad& g, const msk_tensor& bias_src, float* db, int accumulate) {
  const size_t aper = (size_t)g.AD * g.AH * g.AW * g.ald * sizeof(float), bper = (size_t)g.BD * g.BH * g.BW * g.bld * sizeof(float);
  const size_t per = aper > bper ? aper : bper;
  long nmax = per > 0 ? (long)(kChunkBytes / per) : g.N;
  if (nmax < 1) nmax = 1;
  if (g.N <= nmax) {
    // the bias gradient is a column sum of a tensor the weight-gradient kernel reads anyway: a kernel that can fold it in
    // does (wgrad_db_done), otherwise the separate pass follows
    WGrad gb = g;
    if (db && (bias_src.p == g.A || bias_src.p == g.B)) {
      gb.db = db;
      gb.db_src = bias_src.p == g.B ? 1 : 2;
    }
    ctx->wgrad_db_done = false;
    if (int rc = run_wgrad_one(ctx, gb)) return rc;
    if (db && !ctx->wgrad_db_done) return msk_channel_sum(ctx, bias_src, db, accumulate) != 0 ? -1 : 0;
    return 0;
  }
  if (db) {
    if (msk_channel_sum(ctx, bias_src, db, accumulate) != 0) return -1;
  }
  const int total = g.N;
  for (int n0 = 0; n0 < total; n0 += (int)nmax) {
    WGrad c = g;
    c.xform = nullptr;
    c.N = total - n0 < nmax ? total - n0 : (int)nmax;
    c.A = g.A + (size_t)n0 * (aper / sizeof(float));
    c.B = g.B + (size_t)n0 * (bper / sizeof(float));
    if (n0 > 0) c.accumulate = 1;
    if (int rc = run_wgrad_one(ctx, c)) return rc;
  }
  return 0;
}

// msk_conv3d_wgrad_ex / msk_conv3d_dgrad with the maximum |dy| the caller may already hold (an amax array, msk_wbf.h)
int conv3d_wgrad_impl(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, msk_tensor dy, float* dw, float* db, int accumulate,
                      const void* xform, const float* dy_amax, const float* x_amax = nullptr) {
  if (check_conv_shapes(ctx, cd, x, dy, false) != 0) return -1;
  if (xform && !ctx->xform_ok.count(xform)) xform = nullptr;   // never filled by msk_conv3d_fwd_ex*
  msk_side_scope side(ctx, ctx->wgrad_async_max_m <= 0 || msk_voxels(dy) <= ctx->wgrad_async_max_m);
  WGrad g{};
  g.A = (const float*)x.p; g.ald = x.ld; g.B = (const float*)dy.p; g.bld = dy.ld;
  g.N = x.n; g.AD = x.d; g.AH = x.h; g.AW = x.w; g.BD = dy.d; g.BH = dy.h; g.BW = dy.w;
  g.CA = x.c; g.CB = dy.c;
  g.kd = cd.kd; g.kh = cd.kh; g.kw = cd.kw; g.sd = cd.sd; g.sh = cd.sh; g.sw = cd.sw;
  g.pd = cd.pd; g.ph = cd.ph; g.pw = cd.pw;
  g.dw = dw; g.accumulate = accumulate;
  g.xform = xform;
  g.b_amax = dy_amax;
  g.a_amax = x_amax;
  return run_wgrad(ctx, g, dy, db, accumulate);
}

int conv3d_dgrad_impl(msk_ctx* ctx, msk_conv_desc cd, msk_tensor dy, const float* w, msk_tensor dx, int accumulate,
                      const float* dy_amax) {
  if (check_conv_shapes(ctx, cd, dx, dy, false) != 0) return -1;
  GConv g{};
  g.src = (const float*)dy.p; g.sld = dy.ld; g.dst = (float*)dx.p; g.dld = dx.ld;
  g.N = dx.n; g.SD = dy.d; g.SH = dy.h; g.SW = dy.w; g.DD = dx.d; g.DH = dx.h; g.DW = dx.w;
  g.CK = dy.c; g.CN = dx.c;
  g.kd = cd.kd; g.kh = cd.kh; g.kw = cd.kw; g.sd = cd.sd; g.sh = cd.sh; g.sw = cd.sw;
  g.pd = cd.pd; g.ph = cd.ph; g.pw = cd.pw;
  g.transposed = 1; g.bias = nullptr; g.accumulate = accumulate; g.flip = 1;
  g.w_persistent = true;
  g.in_amax = dy_amax;
  // w[Cout][Cin][tap]: k = Cout = a, n = Cin = b -> no swap
  return run_gconv(ctx, g, w, dy.c, dx.c, 0, "conv3d_dgrad_direct");
}

extern "C" {

int msk_conv_fold_bn(msk_ctx* ctx, const float* w, const float* bias, const float* scale, const float* shift, int cout,
                     long inner, float* w_folded, float* b_folded) {
  MSK_REQUIRE(ctx, w && scale && shift && w_folded && b_folded && cout > 0 && inner > 0, "bad arguments");
  msk_weights_changed_impl(ctx, w_folded, (size_t)cout * inner * sizeof(float));
  msk_launch_scope ls(ctx, "fold_bn");
  hipLaunchKernelGGL(fold_bn_k, dim3(grid_for((long)cout * inner, ctx->num_cu)), dim3(kThreads), 0, ctx->stream, w, bias, scale,
                     shift, cout, inner, w_folded, b_folded);
  MSK_LAUNCH_CHECK(ctx);
  return 0;
}

int msk_conv3d_fwd_act(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, const float* bias,
                       const float* prelu_slope, msk_tensor y) {
  if (check_conv_shapes(ctx, cd, x, y, false) != 0) return -1;
  GConv g{};
  g.src = (const float*)x.p; g.sld = x.ld; g.dst = (float*)y.p; g.dld = y.ld;
  g.N = x.n; g.SD = x.d; g.SH = x.h; g.SW = x.w; g.DD = y.d; g.DH = y.h; g.DW = y.w;
  g.CK = x.c; g.CN = y.c;
  g.kd = cd.kd; g.kh = cd.kh; g.kw = cd.kw; g.sd = cd.sd; g.sh = cd.sh; g.sw = cd.sw;
  g.pd = cd.pd; g.ph = cd.ph; g.pw = cd.pw;
  g.transposed = 0; g.bias = bias; g.accumulate = 0; g.flip = 0;
  g.w_persistent = true;
  g.prelu = prelu_slope;
  return run_gconv(ctx, g, w, y.c, x.c, 1, "conv3d_fwd_direct");
}

int msk_conv3d_fwd(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, const float* bias, msk_tensor y) {
  if (check_conv_shapes(ctx, cd, x, y, false) != 0) return -1;
  GConv g{};
  g.src = (const float*)x.p; g.sld = x.ld; g.dst = (float*)y.p; g.dld = y.ld;
  g.N = x.n; g.SD = x.d; g.SH = x.h; g.SW = x.w; g.DD = y.d; g.DH = y.h; g.DW = y.w;
  g.CK = x.c; g.CN = y.c;
  g.kd = cd.kd; g.kh = cd.kh; g.kw = cd.kw; g.sd = cd.sd; g.sh = cd.sh; g.sw = cd.sw;
  g.pd = cd.pd; g.ph = cd.ph; g.pw = cd.pw;
  g.transposed = 0; g.bias = bias; g.accumulate = 0; g.flip = 0;
  g.w_persistent = true;
  // w[Cout][Cin][tap]: k = Cin = b, n = Cout = a -> swap
  return run_gconv(ctx, g, w, y.c, x.c, 1, "conv3d_fwd_direct");
}

size_t msk_conv3d_xform_bytes(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, int cout) {
  if (!ctx->wbf || ctx->no_winograd || ctx->conv_impl != 0) return 0;
  const bool k5 = cd.kd == 5 && cd.kh == 5 && cd.kw == 5 && cd.pd == 2 && cd.ph == 2 && cd.pw == 2;
  const bool k3 = cd.kd == 3 && cd.kh == 3 && cd.kw == 3 && cd.pd == 1 && cd.ph == 1 && cd.pw == 1;
  if (!(k5 || k3) || !(cd.sd == 1 && cd.sh == 1 && cd.sw == 1)) return 0;
  const size_t per = (size_t)x.d * x.h * x.w * (x.ld > cout ? x.ld : cout) * sizeof(float);
  if (per > 0 && (size_t)x.n > kChunkBytes / per) return 0;  // chunked batches do not keep the transform
  // out_tr.conv1 class (32 -> ncls <= 3): only the header -- max |x| travels from conv_foldn_h2_k to wgrad_cbs_h2_k
  if (msk_gconv_foldn_h2_accepts(ctx, cd, x, cout)) return kWbfXformHeader;
  if (cout < 32 || cout % 32) return 0;
  if (x.ld % 4 || (((uintptr_t)x.p) & 15)) return 0;
  return msk_wbf_fwd_xform_bytes(ctx, x.n, x.d, x.h, x.w, x.c, cout, k5 ? 5 : 3);
}

int msk_conv3d_fwd_ex(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, const float* bias, msk_tensor y,
                      float* stats_local, void* xform) {
  return msk_conv3d_fwd_ex2(ctx, cd, x, w, bias, y, stats_local, xform, nullptr);
}

int msk_conv3d_fwd_ex2(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, const float* bias, msk_tensor y,
                       float* stats_local, void* xform, const float* x_amax) {
  return msk_conv3d_fwd_ex3(ctx, cd, x, w, bias, y, stats_local, xform, x_amax, nullptr);
}

int msk_conv3d_fwd_ex3(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, const float* bias, msk_tensor y,
                       float* stats_local, void* xform, const float* x_amax, const msk_bn_fin* fin) {
  if (check_conv_shapes(ctx, cd, x, y, false) != 0) return -1;
  if (fin && !fin->scale) fin = nullptr;
  MSK_REQUIRE(ctx, !fin || stats_local, "msk_conv3d_fwd_ex3: fin needs stats_local");
  GConv g{};
  g.src = (const float*)x.p; g.sld = x.ld; g.dst = (float*)y.p; g.dld = y.ld;
  g.N = x.n; g.SD = x.d; g.SH = x.h; g.SW = x.w; g.DD = y.d; g.DH = y.h; g.DW = y.w;
  g.CK = x.c; g.CN = y.c;
  g.kd = cd.kd; g.kh = cd.kh; g.kw = cd.kw; g.sd = cd.sd; g.sh = cd.sh; g.sw = cd.sw;
  g.pd = cd.pd; g.ph = cd.ph; g.pw = cd.pw;
  g.transposed = 0; g.bias = bias; g.accumulate = 0; g.flip = 0;
  g.w_persistent = true;
  const size_t sper = (size_t)g.SD * g.SH * g.SW * g.sld * sizeof(float), dper = (size_t)g.DD * g.DH * g.DW * g.dld * sizeof(float);
  const size_t per = sper > dper ? sper : dper;
  const bool chunked = per > 0 && (size_t)g.N > kChunkBytes / per;
  ctx->stats_fused = false;
  ctx->xform_written = false;
  if (!chunked) {
    g.stats = stats_local;
    g.xform = xform;
    g.fin = fin;
  }
  g.in_amax = x_amax;  // max |x| from the pass that produced x (msk_amax_new): the fp16 two-piece pipeline skips its own read of x
  if (int rc = run_gconv(ctx, g, w, y.c, x.c, 1, "conv3d_fwd_direct")) return rc;
  if (xform) {
    // the eligibility test of msk_conv3d_xform_bytes sees x and cout only; if the pipeline declined after all (an unaligned y,
    // a size limit) the buffer stays unwritten and the gradient entry points ignore it (they recompute the transform)
    if (ctx->xform_ok.size() > 8192) ctx->xform_ok.clear();
    if (ctx->xform_written) ctx->xform_ok.insert(xform);
    else ctx->xform_ok.erase(xform);
  }
  if (stats_local && !ctx->stats_fused) return msk_bn_stats_fin(ctx, y, stats_local, fin);
  return 0;
}

int msk_conv3d_fwd_in(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, const float* bias, msk_tensor y,
                      float* stats, void* xform, const float* x_amax, const msk_bn_fin* fin, int fin_stride) {
  if (check_conv_shapes(ctx, cd, x, y, false) != 0) return -1;
  MSK_REQUIRE(ctx, stats != nullptr && fin != nullptr && fin->scale != nullptr && fin_stride >= 0, "msk_conv3d_fwd_in: stats and fin required");
  GConv g{};
  g.src = (const float*)x.p; g.sld = x.ld; g.dst = (float*)y.p; g.dld = y.ld;
  g.N = x.n; g.SD = x.d; g.SH = x.h; g.SW = x.w; g.DD = y.d; g.DH = y.h; g.DW = y.w;
  g.CK = x.c; g.CN = y.c;
  g.kd = cd.kd; g.kh = cd.kh; g.kw = cd.kw; g.sd = cd.sd; g.sh = cd.sh; g.sw = cd.sw;
  g.pd = cd.pd; g.ph = cd.ph; g.pw = cd.pw;
  g.transposed = 0; g.bias = bias; g.accumulate = 0; g.flip = 0;
  g.w_persistent = true;
  const size_t sper = (size_t)g.SD * g.SH * g.SW * g.sld * sizeof(float), dper = (size_t)g.DD * g.DH * g.DW * g.dld * sizeof(float);
  const size_t per = sper > dper ? sper : dper;
  const bool chunked = per > 0 && (size_t)g.N > kChunkBytes / per;
  ctx->stats_fused = false;
  ctx->xform_written = false;
  if (!chunked) {
    g.stats = stats;
    g.stats_ps = 1;
    g.fin = fin;
    g.fin_stride = fin_stride;
    g.xform = xform;
  }
  g.in_amax = x_amax;
  if (int rc = run_gconv(ctx, g, w, y.c, x.c, 1, "conv3d_fwd_direct")) return rc;
  if (xform) {
    if (ctx->xform_ok.size() > 8192) ctx->xform_ok.clear();
    if (ctx->xform_written) ctx->xform_ok.insert(xform);
    else ctx->xform_ok.erase(xform);
  }
  if (ctx->stats_fused) return 0;
  // the kernel that ran keeps no per-tile records: one statistics pass per sample
  const size_t svox = (size_t)y.d * y.h * y.w;
  for (int n = 0; n < y.n; ++n) {
    msk_tensor yn = y;
    yn.p = (float*)y.p + (size_t)n * svox * y.ld;
    yn.n = 1;
    msk_bn_fin fn = *fin;
    const long o = (long)n * fin_stride;
    fn.save_mean += o; fn.save_invstd += o; fn.scale += o; fn.shift += o;
    if (int rc = msk_bn_stats_fin(ctx, yn, stats + (size_t)n * 2 * y.c, &fn)) return rc;
  }
  return 0;
}

int msk_conv3d_wgrad_ex(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, msk_tensor dy, float* dw, float* db, int accumulate,
                        const void* xform) {
  return conv3d_wgrad_impl(ctx, cd, x, dy, dw, db, accumulate, xform, nullptr);
}

int msk_conv3d_dgrad(msk_ctx* ctx, msk_conv_desc cd, msk_tensor dy, const float* w, msk_tensor dx, int accumulate) {
  return conv3d_dgrad_impl(ctx, cd, dy, w, dx, accumulate, nullptr);
}

int msk_conv3d_dgrad_ex(msk_ctx* ctx, msk_conv_desc cd, msk_tensor dy, const float* w, msk_tensor dx, int accumulate,
                        const float* dy_amax) {
  return conv3d_dgrad_impl(ctx, cd, dy, w, dx, accumulate, dy_amax);
}

int msk_conv3d_wgrad_ex2(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, msk_tensor dy, float* dw, float* db, int accumulate,
                         const void* xform, const float* dy_amax) {
  return conv3d_wgrad_impl(ctx, cd, x, dy, dw, db, accumulate, xform, dy_amax);
}

int msk_conv3d_wgrad_ex3(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, msk_tensor dy, float* dw, float* db, int accumulate,
                         const void* xform, const float* dy_amax, const float* x_amax) {
  return conv3d_wgrad_impl(ctx, cd, x, dy, dw, db, accumulate, xform, dy_amax, x_amax);
}

int msk_conv3d_wgrad(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, msk_tensor dy, float* dw, float* db, int accumulate) {
  if (check_conv_shapes(ctx, cd, x, dy, false) != 0) return -1;
  msk_side_scope side(ctx, ctx->wgrad_async_max_m <= 0 || msk_voxels(dy) <= ctx->wgrad_async_max_m);
  WGrad g{};
  g.A = (const float*)x.p; g.ald = x.ld; g.B = (const float*)dy.p; g.bld = dy.ld;
  g.N = x.n; g.AD = x.d; g.AH = x.h; g.AW = x.w; g.BD = dy.d; g.BH = dy.h; g.BW = dy.w;
  g.CA = x.c; g.CB = dy.c;
  g.kd = cd.kd; g.kh = cd.kh; g.kw = cd.kw; g.sd = cd.sd; g.sh = cd.sh; g.sw = cd.sw;
  g.pd = cd.pd; g.ph = cd.ph; g.pw = cd.pw;
  g.dw = dw; g.accumulate = accumulate;
  return run_wgrad(ctx, g, dy, db, accumulate);
}

}  // extern "C"

// in_tr.conv1 class: weight gradient with dy = BatchNorm/PReLU backward of (y, dout) evaluated in the kernel.  0 done, 1 declined
// (nothing launched), < 0 error.  It is the LAST weight gradient of a backward pass: it runs on the calling stream -- the side
// stream may still be busy with the gradients queued before it, and nothing is left here for it to overlap with.
static int bwd_bnact_c1(msk_ctx* ctx, const WGrad& gw, msk_tensor y, const float* scale, const float* shift, const float* alpha,
                        const float* mean, const float* invstd, msk_tensor dout, const float* sums_total, double M_total,
                        int res_is_input) {
  if (ctx->bwd_fuse == 0 || ctx->conv_impl != 0) return 1;
  // the kernel's tiles are 32 voxels along W and every lane of a tile evaluates dy: on a narrow volume (the MRI slab: W = 12) the dead
  // lanes' share of that work costs more than the separate pass (0.80 vs 0.39 + 0.19 ms at 512 x 512 x 12)
  if (gw.BW * 4 < ((gw.BW + 31) / 32) * 32 * 3) return 1;
  WbfBnBwd bn{};
  bn.y = (const float*)y.p; bn.yld = y.ld; bn.dout = (const float*)dout.p; bn.dld = dout.ld;
  bn.scale = scale; bn.shift = shift; bn.alpha = alpha; bn.mean = mean; bn.invstd = invstd; bn.sums = sums_total;
  bn.invM = (float)(1.0 / M_total);
  bn.res_is_input = res_is_input;
  WGrad gc = gw;
  gc.B = nullptr;
  gc.xform = nullptr;
  gc.yfuse = &bn;
  if (ctx->late_split && ctx->wgrad_async && ctx->side != nullptr) {
    // round 4: at the END of the side stream, behind an event that marks everything queued before it -- msk_sgd_momentum waits
    // for that event only, updates every other parameter and re-packs the weights while this kernel (matrix-bound) still runs,
    // then joins and updates this tensor (msk_loss_optim.hip)
    msk_side_scope side(ctx, true);
    if (side.active) {
      // everything the kernel reads from the parameter arena (the PReLU slopes) as a snapshot taken BEFORE ev_late: the
      // optimizer updates the arena behind that event while the kernel is still running (read / write race otherwise)
      if (!ctx->late_alpha) MSK_CHECK_HIP(ctx, hipMalloc((void**)&ctx->late_alpha, 1024 * sizeof(float)));
      if (alpha && gw.CB <= 1024) {
        MSK_CHECK_HIP(ctx, hipMemcpyAsync(ctx->late_alpha, alpha, (size_t)gw.CB * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        bn.alpha = ctx->late_alpha;
      }
      hipEventRecord(ctx->ev_late, ctx->stream);
    }
    const int r = msk_wgrad_c1(ctx, gc);
    if (r == 1 && side.active) {
      ctx->late_valid = true;
      ctx->late_ptr = gw.dw;
      ctx->late_count = (size_t)gw.CA * gw.CB * gw.kd * gw.kh * gw.kw;
    }
    return r < 0 ? r : (r == 1 ? 0 : 1);
  }
  const int r = msk_wgrad_c1(ctx, gc);   // 0: not its shape class
  return r < 0 ? r : (r == 1 ? 0 : 1);
}

extern "C" {

int msk_conv3d_bwd_bnact_c1(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, msk_tensor y, const float* scale, const float* shift,
                            const float* alpha, const float* mean, const float* invstd, msk_tensor res, msk_tensor dout,
                            const float* sums_total, double M_total, float* dw, int dw_accumulate) {
  if (check_conv_shapes(ctx, cd, x, y, false) != 0) return -1;
  MSK_REQUIRE(ctx, scale && shift && mean && invstd && sums_total && M_total > 0, "training-mode BatchNorm coefficients required");
  MSK_REQUIRE(ctx, dout.n == y.n && dout.d == y.d && dout.h == y.h && dout.w == y.w && dout.c == y.c, "dout must match y");
  if (x.c != 1) return 1;
  // the residual this class knows is the unit's own input, tiled over the channels (in_tr: out = PReLU(BN(conv(x)) + x))
  if (res.p && !(res.p == x.p && res.ld == x.ld && res.c == 1 && res.n == x.n && res.d == x.d && res.h == x.h && res.w == x.w)) return 1;
  const size_t per = (size_t)x.d * x.h * x.w * (x.ld > y.ld ? x.ld : y.ld) * sizeof(float);
  if (per > 0 && (size_t)x.n > kChunkBytes / per) return 1;
  WGrad gw{};
  gw.A = (const float*)x.p; gw.ald = x.ld; gw.bld = y.ld;
  gw.N = x.n; gw.AD = x.d; gw.AH = x.h; gw.AW = x.w; gw.BD = y.d; gw.BH = y.h; gw.BW = y.w;
  gw.CA = x.c; gw.CB = y.c;
  gw.kd = cd.kd; gw.kh = cd.kh; gw.kw = cd.kw; gw.sd = cd.sd; gw.sh = cd.sh; gw.sw = cd.sw;
  gw.pd = cd.pd; gw.ph = cd.ph; gw.pw = cd.pw;
  gw.dw = dw; gw.accumulate = dw_accumulate;
  return bwd_bnact_c1(ctx, gw, y, scale, shift, alpha, mean, invstd, dout, sums_total, M_total, res.p ? 1 : 0);
}

size_t msk_conv3d_bwd_bnact_bytes(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, msk_tensor y) {
  // the A dy transform has the layout of the transformed input of a conv over y's shape: same geometry by construction
  if (x.c != y.c || x.d != y.d || x.h != y.h || x.w != y.w) return 0;
  return msk_conv3d_xform_bytes(ctx, cd, y, y.c);
}

// coef_stride / sums_stride != 0: per-sample statistics (msk_conv3d_bwd_inact) -- only the one-kernel fused form serves them;
// fused_only: return 1 with nothing launched when that form is not eligible
static int conv3d_bwd_bnact_impl(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, msk_tensor y, const float* scale,
                                 const float* shift, const float* alpha, const float* mean, const float* invstd, const float* gamma,
                                 msk_tensor dout, const float* sums_total, double M_total, msk_tensor dy_scratch, msk_tensor dx,
                                 int dx_accumulate, float* dw, int dw_accumulate, const void* xform, void* ybuf, const float* maxes,
                                 int coef_stride, int sums_stride, bool fused_only, float* dx_lo = nullptr, float* dx_hi = nullptr,
                                 int dx_csplit = 0, const float* dx_old = nullptr);

int msk_conv3d_bwd_bnact(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, msk_tensor y, const float* scale,
                         const float* shift, const float* alpha, const float* mean, const float* invstd, const float* gamma,
                         msk_tensor dout, const float* sums_total, double M_total, msk_tensor dy_scratch, msk_tensor dx,
                         int dx_accumulate, float* dw, int dw_accumulate, const void* xform, void* ybuf, const float* maxes) {
  return conv3d_bwd_bnact_impl(ctx, cd, x, w, y, scale, shift, alpha, mean, invstd, gamma, dout, sums_total, M_total, dy_scratch, dx,
                               dx_accumulate, dw, dw_accumulate, xform, ybuf, maxes, 0, 0, false);
}

// msk_conv3d_bwd_bnact for the layer behind a zero-copy concat (UpTransition: x = the concat buffer, vnet.py:152-154): dx
// accumulates into the interleaved gradient buffer dx as usual UNLESS the one-kernel matrix stage runs the data gradient -- then
// the sums are stored to the two dense half tensors dx_lo / dx_hi (channels [0, c/2) / [c/2, c), voxel stride c/2) and
// *split_done = 1: the consumers of the halves read dense voxels instead of half of every 128-byte line.  *split_done = 0: dx
// holds the result as with msk_conv3d_bwd_bnact (dx_lo / dx_hi untouched).
int msk_conv3d_bwd_bnact_split(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, msk_tensor y, const float* scale,
                               const float* shift, const float* alpha, const float* mean, const float* invstd, const float* gamma,
                               msk_tensor dout, const float* sums_total, double M_total, msk_tensor dy_scratch, msk_tensor dx,
                               int dx_accumulate, float* dw, int dw_accumulate, const void* xform, void* ybuf, const float* maxes,
                               msk_tensor dx_lo, msk_tensor dx_hi, int* split_done) {
  MSK_REQUIRE(ctx, split_done != nullptr, "msk_conv3d_bwd_bnact_split: split_done required");
  *split_done = 0;
  const bool ok = dx.p && dx_lo.p && dx_hi.p && dx.c % 2 == 0 && dx_lo.c == dx.c / 2 && dx_hi.c == dx.c / 2 && dx_lo.ld == dx.c / 2 &&
                  dx_hi.ld == dx.c / 2 && dx.ld == dx.c && msk_voxels(dx_lo) == msk_voxels(dx) && msk_voxels(dx_hi) == msk_voxels(dx) &&
                  dx_accumulate && ctx->dst_split != 0;
  const int rc = conv3d_bwd_bnact_impl(ctx, cd, x, w, y, scale, shift, alpha, mean, invstd, gamma, dout, sums_total, M_total, dy_scratch, dx,
                                       dx_accumulate, dw, dw_accumulate, xform, ybuf, maxes, 0, 0, false, ok ? (float*)dx_lo.p : nullptr,
                                       ok ? (float*)dx_hi.p : nullptr, ok ? dx.c / 2 : 0);
  if (rc == 0 && ctx->dst_split_done) *split_done = 1;
  ctx->dst_split_done = false;
  return rc;
}

// msk_conv3d_bwd_bnact / _split for a data gradient that accumulates onto values which live in ANOTHER tensor: dx_old (geometry and
// voxel stride of dx; e.g. the gradient a residual join hands to both of its operands, written once).  The output stage reads the
// old values from dx_old and writes the sums to dx (or, with dx_lo / dx_hi and the one-kernel matrix stage, to the dense halves:
// *split_done = 1) -- dx itself need not have been written.  dx_old.p == null: exactly msk_conv3d_bwd_bnact(_split) with
// dx_accumulate as given.  Replaces the in-place `x.grad += ...` of paddle's autograd behind vnet.py:110-111,154.
int msk_conv3d_bwd_bnact_acc(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, msk_tensor y, const float* scale,
                             const float* shift, const float* alpha, const float* mean, const float* invstd, const float* gamma,
                             msk_tensor dout, const float* sums_total, double M_total, msk_tensor dy_scratch, msk_tensor dx,
                             int dx_accumulate, float* dw, int dw_accumulate, const void* xform, void* ybuf, const float* maxes,
                             msk_tensor dx_old, msk_tensor dx_lo, msk_tensor dx_hi, int* split_done) {
  if (split_done) *split_done = 0;
  if (dx_old.p) {
    MSK_REQUIRE(ctx, dx.p && dx_old.n == dx.n && dx_old.d == dx.d && dx_old.h == dx.h && dx_old.w == dx.w && dx_old.c == dx.c &&
                         dx_old.ld == dx.ld && dx_old.p != dx.p, "dx_old must have the geometry and voxel stride of dx");
    dx_accumulate = 1;
  }
  const bool ok = split_done && dx.p && dx_lo.p && dx_hi.p && dx.c % 2 == 0 && dx_lo.c == dx.c / 2 && dx_hi.c == dx.c / 2 &&
                  dx_lo.ld == dx.c / 2 && dx_hi.ld == dx.c / 2 && dx.ld == dx.c && msk_voxels(dx_lo) == msk_voxels(dx) &&
                  msk_voxels(dx_hi) == msk_voxels(dx) && dx_accumulate && ctx->dst_split != 0;
  const int rc = conv3d_bwd_bnact_impl(ctx, cd, x, w, y, scale, shift, alpha, mean, invstd, gamma, dout, sums_total, M_total, dy_scratch, dx,
                                       dx_accumulate, dw, dw_accumulate, xform, ybuf, maxes, 0, 0, false, ok ? (float*)dx_lo.p : nullptr,
                                       ok ? (float*)dx_hi.p : nullptr, ok ? dx.c / 2 : 0, (const float*)dx_old.p);
  if (rc == 0 && ok && ctx->dst_split_done) *split_done = 1;
  ctx->dst_split_done = false;
  return rc;
}

int msk_conv3d_bwd_inact(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, msk_tensor y, const float* scale,
                         const float* shift, const float* alpha, const float* mean, const float* invstd, int coef_stride,
                         msk_tensor dout, const float* sums, int sums_stride, double M_sample, msk_tensor dx, int dx_accumulate,
                         float* dw, int dw_accumulate, const void* xform, void* ybuf, const float* maxes) {
  MSK_REQUIRE(ctx, coef_stride >= y.c && sums_stride >= 2 * y.c, "msk_conv3d_bwd_inact: strides smaller than the records");
  return conv3d_bwd_bnact_impl(ctx, cd, x, w, y, scale, shift, alpha, mean, invstd, nullptr, dout, sums, M_sample, msk_tensor{}, dx,
                               dx_accumulate, dw, dw_accumulate, xform, ybuf, maxes, coef_stride, sums_stride, true);
}

static int conv3d_bwd_bnact_impl(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, msk_tensor y, const float* scale,
                                 const float* shift, const float* alpha, const float* mean, const float* invstd, const float* gamma,
                                 msk_tensor dout, const float* sums_total, double M_total, msk_tensor dy_scratch, msk_tensor dx,
                                 int dx_accumulate, float* dw, int dw_accumulate, const void* xform, void* ybuf, const float* maxes,
                                 int coef_stride, int sums_stride, bool fused_only, float* dx_lo, float* dx_hi, int dx_csplit,
                                 const float* dx_old) {
  if (check_conv_shapes(ctx, cd, x, y, false) != 0) return -1;
  ctx->dst_split_done = false;
  if (!dx_accumulate || !dx.p) dx_old = nullptr;
  MSK_REQUIRE(ctx, scale && shift && mean && invstd && sums_total && M_total > 0, "training-mode BatchNorm coefficients required");
  MSK_REQUIRE(ctx, dout.n == y.n && dout.d == y.d && dout.h == y.h && dout.w == y.w && dout.c == y.c, "dout must match y");
  MSK_REQUIRE(ctx, fused_only || (dy_scratch.p && dy_scratch.n == y.n && dy_scratch.d == y.d && dy_scratch.h == y.h && dy_scratch.w == y.w &&
                                  dy_scratch.c == y.c), "dy_scratch must match y");
  if (xform && !ctx->xform_ok.count(xform)) xform = nullptr;   // never filled by msk_conv3d_fwd_ex*
  // ---- fused form: conditions under which BOTH gradient pipelines take pre-written transforms
  const bool split2 = wbf_pieces(ctx, cd.kd) != 3;   // fp16 operands (two pieces, or one: conv_fp16): power-of-two tensor scales
  bool fused = ctx->bwd_fuse != 0 && (!split2 || maxes != nullptr) && ybuf != nullptr && xform != nullptr && dx.p != nullptr && ctx->wbf && !ctx->no_winograd && ctx->conv_impl == 0 &&
               x.c == y.c && y.ld % 4 == 0 && dout.ld % 4 == 0 && (((uintptr_t)y.p) & 15) == 0 && (((uintptr_t)dout.p) & 15) == 0;
  const size_t per = (size_t)x.d * x.h * x.w * (x.ld > y.ld ? x.ld : y.ld) * sizeof(float);
  if (per > 0 && (size_t)x.n > kChunkBytes / per) fused = false;  // chunked batches
  WGrad gw{};
  gw.A = (const float*)x.p; gw.ald = x.ld; gw.B = (const float*)dy_scratch.p; gw.bld = dy_scratch.ld;
  gw.N = x.n; gw.AD = x.d; gw.AH = x.h; gw.AW = x.w; gw.BD = y.d; gw.BH = y.h; gw.BW = y.w;
  gw.CA = x.c; gw.CB = y.c;
  gw.kd = cd.kd; gw.kh = cd.kh; gw.kw = cd.kw; gw.sd = cd.sd; gw.sh = cd.sh; gw.sw = cd.sw;
  gw.pd = cd.pd; gw.ph = cd.ph; gw.pw = cd.pw;
  gw.dw = dw; gw.accumulate = dw_accumulate;
  gw.xform = xform;
  // ---- one input channel, no data gradient (in_tr.conv1, vnet.py:67): dy is evaluated inside the weight-gradient kernel
  if (x.c == 1 && !dx.p && !fused_only && !(per > 0 && (size_t)x.n > kChunkBytes / per)) {
    const int r = bwd_bnact_c1(ctx, gw, y, scale, shift, alpha, mean, invstd, dout, sums_total, M_total, 0);
    if (r <= 0) return r;   // 0 done, < 0 error, 1 declined: the general forms below
  }
  size_t y_bytes = 0;
  if (fused && !msk_wgrad_wbf_fusable(ctx, gw, &y_bytes)) fused = false;
  if (fused) {
    GConv g{};
    g.src = (const float*)dy_scratch.p; g.sld = dy_scratch.ld; g.dst = (float*)dx.p; g.dld = dx.ld;
    g.N = dx.n; g.SD = y.d; g.SH = y.h; g.SW = y.w; g.DD = dx.d; g.DH = dx.h; g.DW = dx.w;
    g.CK = y.c; g.CN = dx.c;
    g.kd = cd.kd; g.kh = cd.kh; g.kw = cd.kw; g.sd = cd.sd; g.sh = cd.sh; g.sw = cd.sw;
    g.pd = cd.pd; g.ph = cd.ph; g.pw = cd.pw;
    g.transposed = 1; g.bias = nullptr; g.accumulate = dx_accumulate; g.flip = 1;
    g.w_persistent = true;
    g.dst_lo = dx_lo; g.dst_hi = dx_hi; g.dst_csplit = dx_csplit;
    g.acc_src = dx_old;      // both output stages of the pipeline (one-kernel epilogue, wbf_tout_k) read the old values from there
    WbfBnBwd bn{};
    bn.y = (const float*)y.p; bn.yld = y.ld; bn.dout = (const float*)dout.p; bn.dld = dout.ld;
    bn.scale = scale; bn.shift = shift; bn.alpha = alpha; bn.mean = mean; bn.invstd = invstd; bn.sums = sums_total;
    bn.invM = (float)(1.0 / M_total);
    bn.Y = (char*)ybuf;
    bn.coef_stride = coef_stride; bn.sums_stride = sums_stride;
    bn.y_xi = (long)(y_bytes / ((cd.kd == 5) ? 8 : 6));
    if (split2) g.w_amax = (const float*)xform + kWbfAmaxWays;  // max |w| of this layer, left there by the forward pass

    // form 1 (one kernel writes both transforms; least traffic: best without a side stream) or form 2 (each stream's
    // transform evaluates dy itself: the main stream -- the critical path -- moves 20 instead of 28 B per element, the
    // weight-gradient stream 20 instead of 16).  Option "bwd_fuse": 0 = three calls, 1 / 2 = force a form, -1 = auto.
    const bool side_on = ctx->wgrad_async && ctx->side != nullptr &&
                         (ctx->wgrad_async_max_m <= 0 || msk_voxels(y) <= ctx->wgrad_async_max_m);
    // auto: with the two-piece fp16 operands the one-kernel form wins even with the side stream on (24.1 vs 24.7 ms per
    // step: its transforms are a third cheaper); with the exact bf16 x 3 split form 2 does (30.1-30.2 vs 30.3-30.4)
    int form = ctx->bwd_fuse > 0 ? ctx->bwd_fuse : ((side_on && !split2) ? 2 : 1);
    if (fused_only) form = 1;   // per-sample statistics: the one-kernel form (the separate bound kernel of form 2 knows no strides)
    if (fused_only && split2 && !maxes) return 1;
    if (wbf_pieces(ctx, cd.kd) == 2 && form == 1) {
      // per-channel max |dy| for the weight gradient's renormalisation, folded in by the dual transform (compute stream)
      bn.y_cmax = msk_scalar_slots(ctx, (y.c + kWbfAmaxWays - 1) / kWbfAmaxWays);
      if (!bn.y_cmax) return -1;
    }
    if (split2) {  // fp16 pieces: dy is scaled by a power of two from a device-side bound of its maximum
      if (form == 1) {
        // the dual transform evaluates the bound itself and leaves it in a (zeroed) ring array for the kernels behind it
        float* slot = msk_scalar_slots(ctx, 1);
        if (!slot) return -1;
        bn.amax = slot;
        bn.maxes = maxes;
      } else {
        bn.amax = msk_bn_bwd_bound(ctx, y.c, scale, sums_total, M_total, maxes);
        if (!bn.amax) return -1;
      }
    }
    if (form == 2) {
      // nothing is launched unless both pipelines accept: the weight gradient was planned above, ask the data gradient
      bn.Y = nullptr;
      g.fuse = &bn;
      if (msk_gconv_wino_bf3_accepts(ctx, g)) {
        {
          msk_side_scope side(ctx, side_on);
          gw.yfuse = &bn;
          const int rw = msk_wgrad_wbf(ctx, gw);
          if (rw < 0) return rw;
          if (rw == 0) return msk_fail(ctx, __FILE__, __LINE__, "msk_conv3d_bwd_bnact", "weight-gradient pipeline declined a problem its plan accepted");
        }
        const int r = msk_gconv_wino_bf3(ctx, g, w, y.c, x.c, 0);
        if (r < 0) return r;
        if (r == 0) return msk_fail(ctx, __FILE__, __LINE__, "msk_conv3d_bwd_bnact", "data-gradient pipeline declined a problem its plan accepted");
        return 0;
      }
    } else {
      g.fuse = &bn;
      const int r = msk_gconv_wino_bf3(ctx, g, w, y.c, x.c, 0);   // launches nothing when it returns 0
      if (r < 0) return r;
      if (r == 1) {
        msk_side_scope side(ctx, side_on);
        gw.yform = ybuf;
        gw.y_amax = bn.amax;
        gw.y_cmax = bn.y_cmax;
        const int rw = msk_wgrad_wbf(ctx, gw);
        if (rw < 0) return rw;
        if (rw == 0) return msk_fail(ctx, __FILE__, __LINE__, "msk_conv3d_bwd_bnact", "weight-gradient pipeline declined a problem its plan accepted");
        return 0;
      }
    }
  }
  if (fused_only) return 1;   // declined, nothing launched: the caller runs its own passes
  // ---- three-kernel form: dy through HBM
  (void)gamma;
  if (dx_old) {   // the general data-gradient kernels accumulate in place: put the old values where they expect them
    msk_tensor so = dx;
    so.p = (void*)dx_old;
    if (int rc = msk_copy_scale(ctx, so, nullptr, dx, 0)) return rc;
  }
  if (int rc = msk_affine_act_bwd_apply(ctx, y, scale, shift, msk_tensor{}, alpha, mean, invstd, gamma, dout, sums_total, M_total, 1,
                                        dy_scratch, msk_tensor{}, 0))
    return rc;
  // out_tr.conv1 class with fp16 two-piece operands: both gradient kernels scale dy by its maximum -- taken once here
  const float* dy_amax = nullptr;
  if (ctx->conv_split == 2 && ctx->conv_impl == 0 && y.c <= 4 && cd.kd == 5 && cd.kh == 5 && cd.kw == 5) {
    dy_amax = msk_absmax(ctx, (const float*)dy_scratch.p, dy_scratch.ld, dy_scratch.c, msk_voxels(dy_scratch));
    if (!dy_amax) return -1;
  }
  // the weight gradient first: it forks to the side stream and overlaps the data gradient enqueued behind it
  if (int rc = conv3d_wgrad_impl(ctx, cd, x, dy_scratch, dw, nullptr, dw_accumulate, xform, dy_amax)) return rc;
  if (dx.p) return conv3d_dgrad_impl(ctx, cd, dy_scratch, w, dx, dx_accumulate, dy_amax);
  return 0;
}

int msk_convT3d_fwd(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, const float* bias, msk_tensor y) {
  if (check_conv_shapes(ctx, cd, x, y, true) != 0) return -1;
  GConv g{};
  g.src = (const float*)x.p; g.sld = x.ld; g.dst = (float*)y.p; g.dld = y.ld;
  g.N = x.n; g.SD = x.d; g.SH = x.h; g.SW = x.w; g.DD = y.d; g.DH = y.h; g.DW = y.w;
  g.CK = x.c; g.CN = y.c;
  g.kd = cd.kd; g.kh = cd.kh; g.kw = cd.kw; g.sd = cd.sd; g.sh = cd.sh; g.sw = cd.sw;
  g.pd = 0; g.ph = 0; g.pw = 0;
  g.transposed = 1; g.bias = bias; g.accumulate = 0; g.flip = 0;
  // w[Cin][Cout][tap]: k = Cin = a, n = Cout = b -> no swap
  return run_gconv(ctx, g, w, x.c, y.c, 0, "convT3d_fwd_direct");
}

// msk_convT3d_fwd + the BatchNorm statistics of y (stats_local [2 C]: mean, M2; fin nullable: msk_bn_finalize(world 1) in the
// merge launch), taken in the store pass of the kernel where it can (convT_scatter_lds_k), by msk_bn_stats_fin otherwise
int msk_convT3d_fwd_ex(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, const float* bias, msk_tensor y,
                       float* stats_local, const msk_bn_fin* fin) {
  if (check_conv_shapes(ctx, cd, x, y, true) != 0) return -1;
  if (fin && !fin->scale) fin = nullptr;
  MSK_REQUIRE(ctx, stats_local != nullptr, "msk_convT3d_fwd_ex: stats_local required");
  GConv g{};
  g.src = (const float*)x.p; g.sld = x.ld; g.dst = (float*)y.p; g.dld = y.ld;
  g.N = x.n; g.SD = x.d; g.SH = x.h; g.SW = x.w; g.DD = y.d; g.DH = y.h; g.DW = y.w;
  g.CK = x.c; g.CN = y.c;
  g.kd = cd.kd; g.kh = cd.kh; g.kw = cd.kw; g.sd = cd.sd; g.sh = cd.sh; g.sw = cd.sw;
  g.pd = 0; g.ph = 0; g.pw = 0;
  g.transposed = 1; g.bias = bias; g.accumulate = 0; g.flip = 0;
  g.w_persistent = true;
  const size_t sper = (size_t)g.SD * g.SH * g.SW * g.sld * sizeof(float), dper = (size_t)g.DD * g.DH * g.DW * g.dld * sizeof(float);
  const size_t per = sper > dper ? sper : dper;
  const bool chunked = per > 0 && (size_t)g.N > kChunkBytes / per;
  ctx->stats_fused = false;
  if (!chunked) {
    g.stats = stats_local;
    g.fin = fin;
  }
  if (int rc = run_gconv(ctx, g, w, x.c, y.c, 0, "convT3d_fwd_direct")) return rc;
  if (!ctx->stats_fused) return msk_bn_stats_fin(ctx, y, stats_local, fin);
  return 0;
}

int msk_convT3d_dgrad(msk_ctx* ctx, msk_conv_desc cd, msk_tensor dy, const float* w, msk_tensor dx, int accumulate) {
  if (check_conv_shapes(ctx, cd, dx, dy, true) != 0) return -1;
  GConv g{};
  g.src = (const float*)dy.p; g.sld = dy.ld; g.dst = (float*)dx.p; g.dld = dx.ld;
  g.N = dx.n; g.SD = dy.d; g.SH = dy.h; g.SW = dy.w; g.DD = dx.d; g.DH = dx.h; g.DW = dx.w;
  g.CK = dy.c; g.CN = dx.c;
  g.kd = cd.kd; g.kh = cd.kh; g.kw = cd.kw; g.sd = cd.sd; g.sh = cd.sh; g.sw = cd.sw;
  g.pd = 0; g.ph = 0; g.pw = 0;
  g.transposed = 0; g.bias = nullptr; g.accumulate = accumulate; g.flip = 0;
  // w[Cin][Cout][tap]: k = Cout = b, n = Cin = a -> swap
  return run_gconv(ctx, g, w, dx.c, dy.c, 1, "convT3d_dgrad_direct");
}

// Backward of an up-convolution unit convT -> BatchNorm -> PReLU (vnet.py:133-150; autograd of core/train.py:139) behind its
// reduce pass (msk_affine_act_bwd_reduce*): the data gradient evaluates dy from (y, dout) in its own loads on the compute stream
// (gconv_ks_fwd_k<.., FUSE>), the pass that writes dy and the weight gradient that reads it run on the weight-gradient stream.
// Returns 0 = done, 1 = not eligible (nothing launched: the caller runs apply / wgrad / dgrad), < 0 error.
int msk_convT3d_bwd_bnact(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, const float* w, msk_tensor y, const float* scale,
                          const float* shift, const float* alpha, const float* mean, const float* invstd, msk_tensor dout,
                          const float* sums_total, double M_total, msk_tensor dy, msk_tensor dx, int dx_accumulate, float* dw,
                          int dw_accumulate) {
  if (check_conv_shapes(ctx, cd, x, y, true) != 0) return -1;
  MSK_REQUIRE(ctx, scale && shift && mean && invstd && sums_total && w && dw, "msk_convT3d_bwd_bnact: null argument");
  MSK_REQUIRE(ctx, dout.c == y.c && msk_voxels(dout) == msk_voxels(y) && dy.c == y.c && msk_voxels(dy) == msk_voxels(y) &&
                       dx.c == x.c && msk_voxels(dx) == msk_voxels(x), "msk_convT3d_bwd_bnact: shape mismatch");
  GConv g{};
  g.src = (const float*)y.p; g.sld = y.ld; g.dst = (float*)dx.p; g.dld = dx.ld;
  g.N = dx.n; g.SD = y.d; g.SH = y.h; g.SW = y.w; g.DD = dx.d; g.DH = dx.h; g.DW = dx.w;
  g.CK = y.c; g.CN = dx.c;
  g.kd = cd.kd; g.kh = cd.kh; g.kw = cd.kw; g.sd = cd.sd; g.sh = cd.sh; g.sw = cd.sw;
  g.transposed = 0; g.bias = nullptr; g.accumulate = dx_accumulate; g.flip = 0;
  if (ctx->conv_impl != 0 || ctx->no_winograd) return 1;
  const size_t per = (size_t)y.d * y.h * y.w * (y.ld > dout.ld ? y.ld : dout.ld) * sizeof(float);
  if (per > 0 && (size_t)g.N > kChunkBytes / per) return 1;     // chunked problems: the plain path
  // w[Cin][Cout][tap]: k = Cout = b, n = Cin = a -> swap (msk_convT3d_dgrad)
  const int ok = msk_gconv_ks_fwd_bnbwd(ctx, g, w, dx.c, y.c, 1, (const float*)y.p, y.ld, (const float*)dout.p, dout.ld, scale, shift,
                                        alpha, mean, invstd, sums_total, M_total, true);
  if (ok != 1) return ok < 0 ? ok : 1;
  {
    // weight-gradient stream (forks HERE, in front of the data gradient): dy for the weight gradient, then the weight gradient
    msk_side_scope side(ctx, ctx->wgrad_async_max_m <= 0 || msk_voxels(x) <= ctx->wgrad_async_max_m);
    msk_tensor none{};
    if (msk_affine_act_bwd_apply_amax(ctx, y, scale, shift, none, alpha, mean, invstd, nullptr, dout, sums_total, M_total, 1, dy,
                                      none, 0, nullptr) != 0) return -1;
    WGrad wg{};
    wg.A = (const float*)dy.p; wg.ald = dy.ld; wg.B = (const float*)x.p; wg.bld = x.ld;
    wg.N = x.n; wg.AD = dy.d; wg.AH = dy.h; wg.AW = dy.w; wg.BD = x.d; wg.BH = x.h; wg.BW = x.w;
    wg.CA = dy.c; wg.CB = x.c;
    wg.kd = cd.kd; wg.kh = cd.kh; wg.kw = cd.kw; wg.sd = cd.sd; wg.sh = cd.sh; wg.sw = cd.sw;
    wg.dw = dw; wg.accumulate = dw_accumulate;
    if (int rc = run_wgrad(ctx, wg, dy, nullptr, dw_accumulate)) return rc;
  }
  const int rc = msk_gconv_ks_fwd_bnbwd(ctx, g, w, dx.c, y.c, 1, (const float*)y.p, y.ld, (const float*)dout.p, dout.ld, scale, shift,
                                        alpha, mean, invstd, sums_total, M_total, false);
  return rc == 1 ? 0 : (rc < 0 ? rc : msk_fail(ctx, __FILE__, __LINE__, "msk_convT3d_bwd_bnact", "the data gradient declined after its dry run"));
}

int msk_convT3d_wgrad(msk_ctx* ctx, msk_conv_desc cd, msk_tensor x, msk_tensor dy, float* dw, float* db, int accumulate) {
  if (check_conv_shapes(ctx, cd, x, dy, true) != 0) return -1;
  msk_side_scope side(ctx, ctx->wgrad_async_max_m <= 0 || msk_voxels(x) <= ctx->wgrad_async_max_m);
  // dWT[ci][co][tap] = sum_ipos x[ipos][ci] * dy[ipos*s + k][co]: conv wgrad with A = dy, B = x
  WGrad g{};
  g.A = (const float*)dy.p; g.ald = dy.ld; g.B = (const float*)x.p; g.bld = x.ld;
  g.N = x.n; g.AD = dy.d; g.AH = dy.h; g.AW = dy.w; g.BD = x.d; g.BH = x.h; g.BW = x.w;
  g.CA = dy.c; g.CB = x.c;
  g.kd = cd.kd; g.kh = cd.kh; g.kw = cd.kw; g.sd = cd.sd; g.sh = cd.sh; g.sw = cd.sw;
  g.pd = 0; g.ph = 0; g.pw = 0;
  g.dw = dw; g.accumulate = accumulate;
  return run_wgrad(ctx, g, dy, db, accumulate);
}

}  // extern "C"
