// Internal definitions shared by the libmsegk translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "msegk.h"

struct msk_prof_entry {
  double total_ms = 0.0;
  long calls = 0;
};

struct msk_pending_event {
  hipEvent_t a, b;
  const char* tag;
};

struct msk_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // scratch workspace (grown on demand, never shrunk)
  void* ws = nullptr;
  size_t ws_bytes = 0;
  void* ws2 = nullptr;  // second independent scratch (packed weights)
  size_t ws2_bytes = 0;
  void* ws3 = nullptr;  // third: channel-padded operands of the wbf pipeline (msk_conv.hip, gconv_wbf_padded)
  size_t ws3_bytes = 0;
  long wbf_tin_groups = -1;        // option "wbf_tin_groups": workgroups below which the transform kernels cut their W tiles into chunks (-1 = 8 per CU, 0 = never)
  long wbf_pad_min_voxels = 1L << 18;  // option "wbf_pad_min_voxels"
  // timing
  hipEvent_t t0 = nullptr, t1 = nullptr;
  hipEvent_t ev_xctx = nullptr;   // msk_ctx_wait: recorded on ANOTHER context's stream, waited for on this one's
  // profiling
  bool prof = false;
  std::map<std::string, msk_prof_entry> prof_map;
  std::vector<msk_pending_event> prof_pending;
  std::vector<hipEvent_t> event_pool;
  std::vector<hipEvent_t> marks;   // msk_mark / msk_mark_elapsed
  std::set<std::string> tag_pool;  // interned dynamic tags (pointers stay valid)
  bool prof_shapes = false;        // append problem shapes to conv tags
  // options
  int conv_impl = 0;  // 0 auto, 1 direct, 3 wgrad direct only, 4 gather-conv direct only
  bool no_winograd = false;  // env MSEGK_DIRECT_CONV=1 / option "direct_conv": direct kernels only (bit-exact fp32 fmaf chains)
  bool wbf = true;  // env MSEGK_WBF=0 / option "wino_bf3" 0: keep the fp32-MFMA Winograd kernels (exact-fp32 products)
  bool dst_split_done = false;  // set by the one-kernel matrix stage when it honoured GConv::dst_lo / dst_hi
  bool stats_fused = false;  // set by a conv kernel that wrote GConv::stats itself
  bool wgrad_db_done = false;  // set by a weight-gradient kernel that produced WGrad::db itself
  bool xform_written = false;  // set when GConv::xform was filled
  std::set<const void*> xform_ok;  // xform buffers msk_conv3d_fwd_ex* really filled: a buffer the pipeline declined (alignment, size limits) is ignored by the weight gradient instead of failing the forward pass (advisor, round 2)
  bool conv_fp16 = false;  // option "conv_fp16": 3x3x3 convolutions with fp16 matrix operands (UNet3D precision='fp16')
  int wbf_tin_map = 1;  // lane mapping of wbf_tin_k (1: one channel group per wavefront, 1 KiB store runs; measured 3-10 % faster)
  int wbf_mr4 = 1;       // the MR = 4 tile variants of wbf_gemm_k first for the levels the one-kernel form does not take (pick_variant)
  int wbf_variant = -1;  // tuning: force a tile variant of wbf_gemm_k (-1 = least padding)
  int halo_tile = -1;  // tuning knob: force the MFMA halo tile (index into the tile table), -1 = pick by utilisation
  int wgrad_chunk = -1;  // same for the LDS wgrad chunk table
  char prof_prefix[48] = {0};  // empty = profile every launch
  long wgrad_async_max_m = 0;  // side stream only for weight gradients over <= this many voxels (0 = all)
  float* scalar_ring = nullptr;  // 1024 device floats handed out round-robin (msk_scalar_slots): amax scalars of the NP = 2 pipelines
  int scalar_next = 0;
  long scalar_served = 0;            // arrays handed out from this ring so far (msk_get_option "scalar_ring_served": the host checks a step's requests against the ring, advisor round 4)
  float* scalar_ring_side = nullptr;  // the weight-gradient stream's own ring (swapped with the stream)
  int scalar_next_side = 0;
  long scalar_served_side = 0;
  int wbf_tpb = 0;            // wbf_gemm_k: tiles per workgroup (0 = 1)
  void* wpack = nullptr;      // packed-weight cache of the Winograd pipelines (msk_conv_wbf.hip: WbfPackCache)
  int wbf_pack_cache = 1;     // 0 = pack the weights on every call (A/B)
  void* spack = nullptr;      // packed-weight cache of the other convolution kernels (msk_conv.hip: SmallPackCache)
  int wbf_pack_lds = 1;       // option "wbf_pack_lds": packed Winograd weights through an LDS tile with whole-run loads and 16-byte stores (wbf_pack_weights_lds_k); 0 = one thread per element (A/B)
  int wbf_bpf = 1;            // option "wbf_bpf": weight-fragment prefetch depth of wbf_gemm_k: 1 (default), 4 = four taps ahead, 0 = 4 for launches of <= 8 workgroups per CU (round 5 A/B: 18.45 ms with 1, 18.58 with 0 / 4 -- the deeper ring costs more than the waits it removes)
  int wbf_ks_blocks = 2;      // option "wbf_ks_blocks": workgroups per CU the split-K of wbf_gemm_k aims for (deep levels; every slab is a round trip of M through HBM)
  int noop_after_merge = 0;   // debug option "noop_after_merge": that many empty launches behind every merge kernel (what a 5-us launch costs the step)
  int wgrad_c1_wpc = 2;       // option "wgrad_c1_wpc": persistent workgroups per CU of wgrad_c1_mfma_k (LDS allows 3)
  int eager_tail_main = 1;    // option "eager_tail_main": msk_sgd_momentum_eager behind a late weight gradient runs on the compute stream (idle by then) instead of queueing behind that kernel on the side stream; 0 = side stream (A/B)
  int prof_paused = 0;        // option "prof_paused": 1 = the profile takes no events until it is set back (no drain, no host synchronisation: bench.py samples every Nth step)
  int prof_attach = 1;        // option "prof_attach": the profile's events ride on the dominant kernel's dispatch (MSK_LAUNCH_TIMED); 0 = marker brackets (A/B)
  int small_pack_cache = 1;   // option "small_pack_cache": 0 = those kernels pack into the shared scratch on every call (A/B)
  int wbf_prepack = 1;        // 1 = rebuild all packed weights in one launch at the end of the optimizer kernels
  int ks_legacy = 0;          // option "ks_legacy" (A/B): bit 0 = one-tap-per-tile k == s weight gradient, bit 1 = fragment-shaped k == s scatter kernel
  int wgrad_fork = 1;         // fused LUConv backward: 1 = the weight gradient forks after the data-gradient GEMM is enqueued (it then overlaps the HBM-bound passes of the next layer instead of stretching that GEMM by 30 %: -0.25 ms per step), 0 = right after the dual transform
  int wbf_fused_bl = 2;       // one-kernel matrix stage, weights through LDS (round 6): 2 = pipelined fills (32 and 64 output channels; default), 1 = whole stage behind one wait (32 channels), 0 = per-wavefront L1 loads (A/B)
  int wbf_fuse = 1;           // 1 = wbf_gemm_fused_k (matrix stage + output transform in one kernel) where eligible; 0 = three stages (A/B)
  int conv_split = 2;         // operand split of the Winograd pipelines: 2 = fp16 two-piece with per-tensor power-of-two scales (product), 3 = exact bf16x3
  int bwd_fuse = -1;          // msk_conv3d_bwd_bnact: -1 auto, 0 three calls, 1 one dual transform, 2 one transform per stream
  int foldn_wgs = 0;          // conv_foldn_k: workgroups per CU targeted by the D segmentation (0 = 2)
  int wgrad_wino_rounds = 0;  // Winograd wgrad kernels: 0 = wave-fitting cost model, > 0 = that many half-waves
  int wgrad_rounds = 8;  // LDS wgrad: target workgroups per CU (split-K granularity)
  int poison = -1;    // debug: byte used to fill freshly (re)allocated scratch
  // side stream: weight gradients run concurrently with the data-gradient chain (they only share
  // inputs), so HBM-bound elementwise backward kernels hide behind MFMA-bound wgrad kernels
  hipStream_t side = nullptr;
  void* ws_side = nullptr;
  size_t ws_side_bytes = 0;
  void* ws3_side = nullptr;  // the side stream's third scratch (wgrad_wbf_padded)
  size_t ws3_side_bytes = 0;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // "late" weight gradient (round 4): the in_tr.conv1 gradient is the LAST of a backward pass; it runs at the end of the side stream
  // behind ev_late, so that the optimizer kernel of everything else (and the weight re-pack in its epilogue) overlaps it
  hipEvent_t ev_late = nullptr;
  bool late_valid = false;
  const float* late_ptr = nullptr;   // gradient tensor still being written behind ev_late
  size_t late_count = 0;
  // the late kernel evaluates PReLU backward in its prologue: its slopes live in the PARAMETER arena, which the optimizer
  // updates behind ev_late while the kernel still runs -- it reads this snapshot (taken before ev_late) instead (advisor, round 5)
  float* late_alpha = nullptr;
  // the late tensor's slice of an eager update that ran on the compute stream (msk_sgd_momentum_eager): updated by
  // msk_sgd_momentum_finish after the join
  struct { float* param = nullptr; const float* grad = nullptr; float* velocity = nullptr; size_t count = 0;
           float lr = 0, momentum = 0, weight_decay = 0, grad_scale = 0; } late_piece;
  int late_split = 1;                // option "late_split": 0 = the late gradient on the calling stream, one optimizer launch (A/B)
  bool wgrad_async = false;
  bool side_dirty = false;
  hipEvent_t fork_event = nullptr;  // the event of that point when it is not ev_fork (the profile's stop event of the same dispatch)
  int fork_attach = 1;         // option "fork_attach": the fork point of a LUConv backward rides on its last data-gradient dispatch (MSK_LAUNCH_TIMED_F) instead of a marker packet behind it (~7 us of idle packet processor on the compute stream, 14 per step); 0 = marker (A/B)
  bool fork_recorded = false;  // ev_fork already recorded at the point the next side scope has to wait for (msk_conv3d_bwd_bnact)
  // msk_h2d staging: two pinned buffers so that a batch upload neither synchronises the stream nor waits for the
  // previous step (the host may run one upload ahead)
  void* stage[2] = {nullptr, nullptr};
  size_t stage_bytes[2] = {0, 0};
  hipEvent_t stage_ev[2] = {nullptr, nullptr};
  int stage_next = 0;
  // data parallel
  void* comm = nullptr;  // ncclComm_t (compute stream: SyncBatchNorm exchanges, broadcast, barrier)
  // gradient buckets: second communicator on its own stream, overlapped with the rest of backward
  void* comm_grad = nullptr;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_comm_main = nullptr, ev_comm_side = nullptr, ev_comm_done = nullptr, ev_comm_back = nullptr;
  int c1_h2 = 1;          // one-input-channel convolutions on the 16-bit pipe (conv_c1_h2_k): 1 = the 3^3 class (UNet3D), 2 = also 5^3 (in_tr), 0 = off
  int dst_split = 1;      // option "dst_split": msk_conv3d_bwd_bnact_split may store the concat gradient as two dense halves; 0 = never (A/B)
  int wgrad_reduce_rows = 1;   // option "wgrad_reduce_rows": split-K reduce of the deep Winograd weight gradients with 500-byte output runs (wbf_wgrad_reduce_rows_k); 0 = the lane-per-cb form (A/B)
  int ks_stats = 1;       // option "ks_stats": BatchNorm statistics in the store pass of the k == s kernels (convT_scatter_lds_k, gconv_ks_lds_k); 0 = the separate pass (A/B)
  int ks_lds = 1;         // option "ks_lds": 2x2x2 / stride 2 convolutions with <= 16 source channels through the LDS tile (gconv_ks_lds_k); 0 = the per-lane gather form (A/B)
  int ks_nr_max = 4;      // gconv_ks_fwd / gconv_kst: most N tiles per workgroup (4: x read once but 160 registers -- such a wavefront finds no room
                          // on a SIMD that holds three weight-gradient wavefronts of the side stream; 2: 94 registers)
  int wgrad_lds_pad = 0;  // bytes of dynamic LDS added to every wbf_wgrad_k launch: caps its workgroups per CU (33.5 KB static: 3 per CU
                          // by registers; >= 20 KB of padding: 2 per CU, >= 47 KB: 1) so that the compute stream's HBM-bound passes find
                          // free registers next to it (tools/stream_timeline.py)
  int kst_pair = 1;      // option "kst_pair": gconv_kst_k carries both h-parity classes of <= 16 output channels in one matrix instruction; 0 = one class per wavefront (A/B)
  int tile_staging = 7;  // option "tile_staging": dense 5..32-channel voxel records through an LDS tile (msk_tile_load); bit mask: 1 the 1x1x1 head, 2 the 5..32-class loss kernels, 4 the 2..4-class loss kernels (thread per voxel through a flat tile instead of the lane-per-class kernels); 0 = the direct forms (A/B)
  int wgrad_renorm = 1;  // NP = 2 weight gradient: per-channel renormalisation (msk_wbf.h: wbf_chan_shift); 0 = per-tensor scales only (A/B)
  int dy_bound_shift = 0;   // debug option "dy_bound_shift": the dual transform's bound of max |dy| times 2^shift
  int dp_mode = 0;   // msk_dp.hip: 0 = every collective on the compute stream (default), 1 = one communicator on the communication stream, 2 = two communicators, 3 = one communicator, buckets on the communication stream
  bool comm_pending = false;
  bool host_transport = false;   // env MSEGK_DP_TRANSPORT=host: collectives through host staging + TCP (single-GPU test tier)
  std::vector<int> host_fds;
  int rank = 0, world = 1;
  int num_cu = 256;
};

extern thread_local std::string g_msk_global_err;

int msk_fail(msk_ctx* ctx, const char* file, int line, const char* what, const char* detail);
void* msk_workspace(msk_ctx* ctx, size_t bytes);   // returns nullptr on failure (error set)
void* msk_workspace2(msk_ctx* ctx, size_t bytes);
void* msk_workspace3(msk_ctx* ctx, size_t bytes);
hipEvent_t msk_prof_event(msk_ctx* ctx);   // a timing event from the profile's pool (for brackets on other streams)
void msk_prof_begin(msk_ctx* ctx, const char* tag);
void msk_prof_end(msk_ctx* ctx);
const char* msk_intern_tag(msk_ctx* ctx, const std::string& s);
int msk_join_side_impl(msk_ctx* ctx);
void msk_set_dense12(int v);           // option "dense12" (msk_elementwise.hip)
void msk_set_reduce_vpl(int v);
void msk_set_reduce_vpl_site(int v);   // option "reduce_vpl_site"          // option "reduce_vpl"
void msk_set_ew_caps(int ew, int red);   // msk_elementwise.hip tuning knobs
// caller memory that may hold convolution weights was (or is about to be) written / freed: derived forms are stale
void msk_weights_changed_impl(msk_ctx* ctx, const void* p, size_t bytes);
void msk_weights_freed_impl(msk_ctx* ctx, const void* p, size_t bytes);  // msk_free: drop the rows inside [p, p + bytes)
int msk_wbf_prepack_impl(msk_ctx* ctx);   // rebuild every stale packed-weight row in use (end of the optimizer kernels)
int msk_wbf_prepack_range_impl(msk_ctx* ctx, const void* p, size_t bytes);   // the stale rows inside [p, p + bytes), current stream
void msk_wbf_pack_cache_free(msk_ctx* ctx);
void msk_small_pack_changed(msk_ctx* ctx, const void* p, size_t bytes);   // SmallPackCache hooks (msk_conv.hip), called by the *_impl functions above
void msk_small_pack_freed(msk_ctx* ctx, const void* p, size_t bytes);
int msk_small_prepack(msk_ctx* ctx, const void* p, size_t bytes);        // p == nullptr: every stale row in use
void msk_small_pack_free(msk_ctx* ctx);
int msk_dp_wait_impl(msk_ctx* ctx);

// Redirect launches of the enclosed scope to the side stream (with its own scratch) after making
// it wait for everything enqueued so far on the main stream.
struct msk_side_scope {
  msk_ctx* ctx;
  bool active;
  explicit msk_side_scope(msk_ctx* c, bool want = true) : ctx(c), active(want && c->wgrad_async && c->side != nullptr) {
    const bool forked = ctx->fork_recorded;  // an earlier fork point was recorded for this scope
    hipEvent_t fev = forked && ctx->fork_event ? ctx->fork_event : ctx->ev_fork;
    ctx->fork_recorded = false;
    ctx->fork_event = nullptr;
    if (!active) return;
    if (!forked) hipEventRecord(ctx->ev_fork, ctx->stream);
    hipStreamWaitEvent(ctx->side, fev, 0);
    std::swap(ctx->stream, ctx->side);
    std::swap(ctx->ws, ctx->ws_side);
    std::swap(ctx->ws_bytes, ctx->ws_side_bytes);
    std::swap(ctx->ws3, ctx->ws3_side);
    std::swap(ctx->ws3_bytes, ctx->ws3_side_bytes);
    std::swap(ctx->scalar_ring, ctx->scalar_ring_side);
    std::swap(ctx->scalar_next, ctx->scalar_next_side);
    std::swap(ctx->scalar_served, ctx->scalar_served_side);
    ctx->side_dirty = true;
  }
  ~msk_side_scope() {
    if (!active) return;
    std::swap(ctx->stream, ctx->side);
    std::swap(ctx->ws, ctx->ws_side);
    std::swap(ctx->ws_bytes, ctx->ws_side_bytes);
    std::swap(ctx->ws3, ctx->ws3_side);
    std::swap(ctx->ws3_bytes, ctx->ws3_side_bytes);
    std::swap(ctx->scalar_ring, ctx->scalar_ring_side);
    std::swap(ctx->scalar_next, ctx->scalar_next_side);
    std::swap(ctx->scalar_served, ctx->scalar_served_side);
  }
};

#define MSK_CHECK_HIP(ctx, expr)                                                        \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) return msk_fail(ctx, __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
  } while (0)

#define MSK_REQUIRE(ctx, cond, msg)                                       \
  do {                                                                    \
    if (!(cond)) return msk_fail(ctx, __FILE__, __LINE__, #cond, msg);    \
  } while (0)

// Launch bracket: set device stream, optional per-kernel profiling, error check.
// does the per-kernel profile take events for this tag right now?
static inline bool msk_prof_selected(const msk_ctx* c, const char* tag) {
  return c->prof && ((!c->prof_paused && (c->prof_prefix[0] == 0 || strncmp(tag, c->prof_prefix, strlen(c->prof_prefix)) == 0)) ||
                     strncmp(tag, "rccl_", 5) == 0);   // the (few) collectives are always bracketed (paused or not)
}
struct msk_launch_scope {
  msk_ctx* ctx;
  bool on;
  // prof_prefix (option "prof_only_halo"): bracket only the launches whose tag starts with it -- two events
  // around EVERY launch cost ~4 % of the training step (packet-processor barriers between back-to-back kernels)
  msk_launch_scope(msk_ctx* c, const char* tag)
      : ctx(c), on(msk_prof_selected(c, tag)) {
    if (on) msk_prof_begin(c, tag);
  }
  ~msk_launch_scope() { if (on) msk_prof_end(ctx); }
};

// HIP events ATTACHED to one launch (the start / stop events of hipExtLaunchKernelGGL: the dispatch's own begin / end timestamps).
// The bracket form above puts two marker packets around the launch, each ~6 us of idle packet processor (round 5: bench.py's
// brackets around the 28 launches of the dominant kernel were 0.33 ms of the step it measured: tools/gap_by_kernel.py); the
// attached form adds no packet.  Option "prof_attach" 0 = brackets (A/B).
struct msk_launch_events {
  hipEvent_t a = nullptr, b = nullptr;
};
bool msk_prof_attach(msk_ctx* ctx, const char* tag, msk_launch_events* ev);   // true: launch with ev->a / ev->b, then msk_prof_attached
void msk_prof_attached(msk_ctx* ctx, const msk_launch_events& ev, const char* tag);
#define MSK_LAUNCH_TIMED(ctx, tag, kernel, grid, block, shmem, ...) MSK_LAUNCH_TIMED_F(ctx, tag, false, kernel, grid, block, shmem, __VA_ARGS__)
// ... and, with `fork`, the point the next msk_side_scope waits for: the dispatch's own completion (its stop event) instead of a
// marker packet recorded behind it
#define MSK_LAUNCH_TIMED_F(ctx, tag, fork, kernel, grid, block, shmem, ...)                                             \
  do {                                                                                                                   \
    msk_launch_events _ev;                                                                                               \
    if (msk_prof_attach(ctx, tag, &_ev)) {                                                                               \
      hipExtLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, _ev.a, _ev.b, 0, __VA_ARGS__);                    \
      msk_prof_attached(ctx, _ev, tag);                                                                                  \
      if (fork) { (ctx)->fork_event = _ev.b; (ctx)->fork_recorded = true; }                                              \
    } else if ((fork) && !msk_prof_selected(ctx, tag)) {                                                                 \
      hipExtLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, (hipEvent_t) nullptr, (ctx)->ev_fork, 0, __VA_ARGS__); \
      (ctx)->fork_event = nullptr;                                                                                       \
      (ctx)->fork_recorded = true;                                                                                       \
    } else {                                                                                                             \
      msk_launch_scope _ls(ctx, tag);                                                                                    \
      hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__);                                        \
    }                                                                                                                    \
  } while (0)

#define MSK_LAUNCH_CHECK(ctx)                                                            \
  do {                                                                                   \
    hipError_t _e = hipGetLastError();                                                   \
    if (_e != hipSuccess) return msk_fail(ctx, __FILE__, __LINE__, "kernel launch", hipGetErrorString(_e)); \
  } while (0)

static inline int msk_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline long msk_voxels(const msk_tensor& t) { return (long)t.n * t.d * t.h * t.w; }

// device helpers ------------------------------------------------------------
// Dense voxel records of rq float4 quads (rq <= 8) between HBM and an LDS tile of 256 records with WHOLE-LINE accesses (round 4):
// a thread-per-voxel kernel whose lane reads its own 80-byte record touches 40 lines per 1 KiB load instruction and ran at
// 1-1.9 TB/s (the 20-class MRI head and its loss: pointwise_mid_k, loss_*_tpv_k); here consecutive lanes move consecutive
// 16-byte quads and a thread then takes its record from LDS.  pitch = rq | 1 slots: 16 consecutive records sit on 16 distinct
// 16-byte bank groups, so the per-record ds_read_b128 / ds_write_b128 are conflict-free.
__device__ __forceinline__ void msk_tile_load(const float* __restrict__ src, int nquads, int rq, int pitch, float4* __restrict__ tile) {
  const float4* s4 = reinterpret_cast<const float4*>(src);
  for (int i = threadIdx.x; i < nquads; i += blockDim.x) {
    const int v = i / rq, q = i - v * rq;
    tile[v * pitch + q] = s4[i];
  }
}
__device__ __forceinline__ void msk_tile_store(float* __restrict__ dst, int nquads, int rq, int pitch, const float4* __restrict__ tile, bool accumulate) {
  float4* d4 = reinterpret_cast<float4*>(dst);
  for (int i = threadIdx.x; i < nquads; i += blockDim.x) {
    const int v = i / rq, q = i - v * rq;
    float4 r = tile[v * pitch + q];
    if (accumulate) {
      const float4 o = d4[i];
      r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
    }
    d4[i] = r;
  }
}

__device__ __forceinline__ float msk_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double msk_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
