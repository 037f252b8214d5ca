// Context, memory, timing and per-kernel profiling for libmsegk (gfx950).
#include <cstdlib>

#include "msk_common.h"

thread_local std::string g_msk_global_err;

int msk_fail(msk_ctx* ctx, const char* file, int line, const char* what, const char* detail) {
  char buf[1024];
  const char* base = strrchr(file, '/');
  snprintf(buf, sizeof(buf), "%s:%d: %s: %s", base ? base + 1 : file, line, what, detail ? detail : "");
  if (ctx) ctx->err = buf;
  g_msk_global_err = buf;
  return -1;
}

static void* grow(msk_ctx* ctx, void** p, size_t* cur, size_t bytes) {
  if (bytes <= *cur) return *p;
  if (*p) {
    hipStreamSynchronize(ctx->stream);
    hipFree(*p);
    *p = nullptr;
    *cur = 0;
  }
  size_t want = bytes + (bytes >> 2);
  if (hipMalloc(p, want) != hipSuccess) {
    msk_fail(ctx, __FILE__, __LINE__, "workspace hipMalloc", "out of memory");
    *p = nullptr;
    return nullptr;
  }
  *cur = want;
  if (ctx->poison >= 0) hipMemsetAsync(*p, ctx->poison, want, ctx->stream);  // debug: make stale-scratch reads visible
  return *p;
}
void* msk_workspace(msk_ctx* ctx, size_t bytes) { return grow(ctx, &ctx->ws, &ctx->ws_bytes, bytes); }
void* msk_workspace2(msk_ctx* ctx, size_t bytes) { return grow(ctx, &ctx->ws2, &ctx->ws2_bytes, bytes); }
void* msk_workspace3(msk_ctx* ctx, size_t bytes) { return grow(ctx, &ctx->ws3, &ctx->ws3_bytes, bytes); }

static hipEvent_t get_event(msk_ctx* ctx) {
  if (!ctx->event_pool.empty()) {
    hipEvent_t e = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}

static void drain_prof(msk_ctx* ctx) {
  for (auto& pe : ctx->prof_pending) {
    hipEventSynchronize(pe.b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, pe.a, pe.b);
    auto& ent = ctx->prof_map[pe.tag];
    ent.total_ms += ms;
    ent.calls += 1;
    ctx->event_pool.push_back(pe.a);
    ctx->event_pool.push_back(pe.b);
  }
  ctx->prof_pending.clear();
}

const char* msk_intern_tag(msk_ctx* ctx, const std::string& s) { return ctx->tag_pool.insert(s).first->c_str(); }

hipEvent_t msk_prof_event(msk_ctx* ctx) { return get_event(ctx); }
void msk_prof_begin(msk_ctx* ctx, const char* tag) {
  msk_pending_event pe;
  pe.a = get_event(ctx);
  pe.b = get_event(ctx);
  pe.tag = tag;
  hipEventRecord(pe.a, ctx->stream);
  ctx->prof_pending.push_back(pe);
}
bool msk_prof_attach(msk_ctx* ctx, const char* tag, msk_launch_events* ev) {
  if (!ctx->prof_attach || !msk_prof_selected(ctx, tag)) return false;
  ev->a = get_event(ctx);
  ev->b = get_event(ctx);
  return true;
}
void msk_prof_attached(msk_ctx* ctx, const msk_launch_events& ev, const char* tag) {
  msk_pending_event pe;
  pe.a = ev.a;
  pe.b = ev.b;
  pe.tag = tag;
  ctx->prof_pending.push_back(pe);
  if (ctx->prof_pending.size() > 4096) drain_prof(ctx);
}
void msk_prof_end(msk_ctx* ctx) {
  hipEventRecord(ctx->prof_pending.back().b, ctx->stream);
  if (ctx->prof_pending.size() > 4096) drain_prof(ctx);
}

int msk_dp_wait_impl(msk_ctx* ctx) {
  if (ctx->comm_pending) {
    MSK_CHECK_HIP(ctx, hipEventRecord(ctx->ev_comm_done, ctx->comm_stream));
    MSK_CHECK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_comm_done, 0));
    ctx->comm_pending = false;
  }
  return 0;
}

int msk_join_side_impl(msk_ctx* ctx) {
  ctx->late_valid = false;   // a full join covers the late gradient as well
  if (ctx->side_dirty) {
    MSK_CHECK_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->side));
    MSK_CHECK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    ctx->side_dirty = false;
  }
  return 0;
}

extern "C" {

int msk_version(void) { return 100; }

int msk_device_count(int* count) {
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    return msk_fail(nullptr, __FILE__, __LINE__, "hipGetDeviceCount", hipGetErrorString(e));
  }
  return 0;
}

int msk_ctx_create(int device, msk_ctx** out) {
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0)
    return msk_fail(nullptr, __FILE__, __LINE__, "msk_ctx_create", "no HIP device visible (libmsegk has no CPU fallback)");
  if (device < 0 || device >= n) return msk_fail(nullptr, __FILE__, __LINE__, "msk_ctx_create", "bad device index");
  msk_ctx* ctx = new msk_ctx();
  ctx->device = device;
  ctx->wgrad_async = true;
  {
    const char* e = getenv("MSEGK_DIRECT_CONV");
    ctx->no_winograd = e && e[0] && e[0] != '0';
    const char* f = getenv("MSEGK_WBF");
    if (f && f[0] == '0') ctx->wbf = false;
    const char* sp = getenv("MSEGK_CONV_SPLIT");  // operand split of the Winograd pipelines (option "conv_split")
    if (sp && sp[0] == '2') ctx->conv_split = 2;
    if (sp && sp[0] == '3') ctx->conv_split = 3;
  }
  MSK_CHECK_HIP(ctx, hipSetDevice(device));
  MSK_CHECK_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  {
    // the weight-gradient stream only has to finish before the optimizer: lowest priority, so that the workgroups of
    // the data-gradient chain (the critical path) are dispatched first
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
    const char* e = getenv("MSEGK_SIDE_PRIORITY");   // A/B: "0" = normal priority
    const int prio = (e && e[0] == '0') ? 0 : least;
    const char* m = getenv("MSEGK_SIDE_CU_FRAC");    // A/B: confine the weight-gradient stream to 1/k of the CUs of every XCD
    const int frac = m ? atoi(m) : 0;
    if (frac > 1) {
      // bit i of the mask is a CU; both plausible orders (XCD-major and XCD-interleaved) give every XCD the same share
      // when whole groups of 8 consecutive bits are switched by (i / 8) % frac
      uint32_t mask[8];
      for (int wd = 0; wd < 8; ++wd) {
        mask[wd] = 0;
        for (int bit = 0; bit < 32; ++bit) {
          const int i = wd * 32 + bit;
          if ((i / 8) % frac == 0) mask[wd] |= 1u << bit;
        }
      }
      MSK_CHECK_HIP(ctx, hipExtStreamCreateWithCUMask(&ctx->side, 8, mask));
    } else {
      MSK_CHECK_HIP(ctx, hipStreamCreateWithPriority(&ctx->side, hipStreamNonBlocking, prio));
    }
  }
  MSK_CHECK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
  MSK_CHECK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  MSK_CHECK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_late, hipEventDisableTiming));
  MSK_CHECK_HIP(ctx, hipEventCreate(&ctx->t0));
  MSK_CHECK_HIP(ctx, hipEventCreate(&ctx->t1));
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cu = prop.multiProcessorCount;
  *out = ctx;
  return 0;
}

int msk_ctx_destroy(msk_ctx* ctx) {
  if (!ctx) return 0;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  msk_dp_destroy(ctx);
  drain_prof(ctx);
  for (auto e : ctx->event_pool) hipEventDestroy(e);
  for (auto e : ctx->marks) if (e) hipEventDestroy(e);
  if (ctx->ev_xctx) hipEventDestroy(ctx->ev_xctx);
  msk_wbf_pack_cache_free(ctx);
  msk_small_pack_free(ctx);
  if (ctx->ws) hipFree(ctx->ws);
  if (ctx->ws2) hipFree(ctx->ws2);
  if (ctx->ws3) hipFree(ctx->ws3);
  if (ctx->ws_side) hipFree(ctx->ws_side);
  if (ctx->ws3_side) hipFree(ctx->ws3_side);
  if (ctx->scalar_ring) hipFree(ctx->scalar_ring);
  if (ctx->scalar_ring_side) hipFree(ctx->scalar_ring_side);
  if (ctx->side) { hipStreamSynchronize(ctx->side); hipStreamDestroy(ctx->side); }
  for (int b = 0; b < 2; ++b) {
    if (ctx->stage_ev[b]) hipEventDestroy(ctx->stage_ev[b]);
    if (ctx->stage[b]) hipHostFree(ctx->stage[b]);
  }
  if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
  if (ctx->ev_late) hipEventDestroy(ctx->ev_late);
  if (ctx->late_alpha) hipFree(ctx->late_alpha);
  hipEventDestroy(ctx->t0);
  hipEventDestroy(ctx->t1);
  hipStreamDestroy(ctx->stream);
  delete ctx;
  return 0;
}

const char* msk_last_error(msk_ctx* ctx) { return ctx ? ctx->err.c_str() : g_msk_global_err.c_str(); }

int msk_sync(msk_ctx* ctx) {
  if (msk_join_side_impl(ctx) != 0) return -1;
  if (msk_dp_wait_impl(ctx) != 0) return -1;
  MSK_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int msk_join_side(msk_ctx* ctx) { return msk_join_side_impl(ctx); }

int msk_device_name(msk_ctx* ctx, char* buf, int buflen) {
  hipDeviceProp_t prop;
  MSK_CHECK_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
  snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return 0;
}

int msk_device_pci_bus_id(msk_ctx* ctx, char* buf, int buflen) {
  MSK_REQUIRE(ctx, buf != nullptr && buflen >= 16, "buffer of at least 16 bytes");
  MSK_CHECK_HIP(ctx, hipDeviceGetPCIBusId(buf, buflen, ctx->device));
  return 0;
}

int msk_malloc(msk_ctx* ctx, size_t bytes, void** out) {
  *out = nullptr;
  if (bytes == 0) bytes = 16;
  MSK_CHECK_HIP(ctx, hipMalloc(out, bytes));
  return 0;
}
int msk_free(msk_ctx* ctx, void* p) {
  if (!p) return 0;
  if (msk_join_side_impl(ctx) != 0) return -1;
  if (msk_dp_wait_impl(ctx) != 0) return -1;
  MSK_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  {
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) == hipSuccess && size > 0) {
      msk_weights_freed_impl(ctx, base, size);
    } else {
      (void)hipGetLastError();
      msk_weights_freed_impl(ctx, p, 0);
    }
  }
  MSK_CHECK_HIP(ctx, hipFree(p));
  return 0;
}
int msk_weights_changed(msk_ctx* ctx, const void* p, size_t bytes) {
  if (!ctx) return -1;
  if (p && bytes) msk_weights_changed_impl(ctx, p, bytes);
  return 0;
}
int msk_memset(msk_ctx* ctx, void* p, int value, size_t bytes) {
  if (bytes == 0) return 0;
  if (msk_join_side_impl(ctx) != 0) return -1;
  msk_weights_changed_impl(ctx, p, bytes);
  MSK_CHECK_HIP(ctx, hipMemsetAsync(p, value, bytes, ctx->stream));
  return 0;
}
int msk_h2d(msk_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return 0;
  msk_weights_changed_impl(ctx, dst, bytes);
  if (bytes >= (64u << 10) && bytes <= ((size_t)64 << 20)) {
    // batch-sized uploads (core/train.py:122-124 every iteration): through one of two pinned staging buffers, so the
    // call returns once src is copied out and the compute stream is NOT synchronised -- with the blocking path below
    // the GPU idled for the upload plus the first launches of every step (2.2 ms of a 44.5 ms step)
    const int b = ctx->stage_next;
    ctx->stage_next ^= 1;
    if (ctx->stage_ev[b] == nullptr) MSK_CHECK_HIP(ctx, hipEventCreateWithFlags(&ctx->stage_ev[b], hipEventDisableTiming));
    else MSK_CHECK_HIP(ctx, hipEventSynchronize(ctx->stage_ev[b]));  // the copy that last used this buffer is done
    if (ctx->stage_bytes[b] < bytes) {
      if (ctx->stage[b]) hipHostFree(ctx->stage[b]);
      ctx->stage[b] = nullptr;
      ctx->stage_bytes[b] = 0;
      MSK_CHECK_HIP(ctx, hipHostMalloc(&ctx->stage[b], bytes, hipHostMallocDefault));
      ctx->stage_bytes[b] = bytes;
    }
    memcpy(ctx->stage[b], src, bytes);
    MSK_CHECK_HIP(ctx, hipMemcpyAsync(dst, ctx->stage[b], bytes, hipMemcpyHostToDevice, ctx->stream));
    MSK_CHECK_HIP(ctx, hipEventRecord(ctx->stage_ev[b], ctx->stream));
    return 0;
  }
  MSK_CHECK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  // pageable host memory: the runtime stages it; make the call safe w.r.t. src reuse
  MSK_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}
int msk_h2d_async(msk_ctx* ctx, void* dst, const void* pinned_src, size_t bytes) {
  if (bytes == 0) return 0;
  msk_weights_changed_impl(ctx, dst, bytes);
  MSK_CHECK_HIP(ctx, hipMemcpyAsync(dst, pinned_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  return 0;
}
int msk_d2h(msk_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return 0;
  if (msk_join_side_impl(ctx) != 0) return -1;
  MSK_CHECK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  MSK_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}
int msk_d2d(msk_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return 0;
  msk_weights_changed_impl(ctx, dst, bytes);
  MSK_CHECK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}
int msk_pinned_alloc(msk_ctx* ctx, size_t bytes, void** out) {
  MSK_CHECK_HIP(ctx, hipHostMalloc(out, bytes, hipHostMallocDefault));
  return 0;
}
int msk_pinned_free(msk_ctx* ctx, void* p) {
  MSK_CHECK_HIP(ctx, hipHostFree(p));
  return 0;
}
int msk_mem_info(msk_ctx* ctx, size_t* free_bytes, size_t* total_bytes) {
  MSK_CHECK_HIP(ctx, hipMemGetInfo(free_bytes, total_bytes));
  return 0;
}

int msk_timer_start(msk_ctx* ctx) {
  MSK_CHECK_HIP(ctx, hipEventRecord(ctx->t0, ctx->stream));
  return 0;
}
int msk_timer_stop(msk_ctx* ctx, float* ms) {
  MSK_CHECK_HIP(ctx, hipEventRecord(ctx->t1, ctx->stream));
  MSK_CHECK_HIP(ctx, hipEventSynchronize(ctx->t1));
  MSK_CHECK_HIP(ctx, hipEventElapsedTime(ms, ctx->t0, ctx->t1));
  return 0;
}

// Marks on the compute stream for per-step timing without host synchronisation inside the timed region (bench.py: the
// median of the timed steps): msk_mark(i) records event i (0..1023), msk_mark_elapsed(a, b) waits for b.
int msk_mark(msk_ctx* ctx, int idx) {
  MSK_REQUIRE(ctx, idx >= 0 && idx < 1024, "mark index out of range");
  // (no join of the side / communication streams: a mark delimits the COMPUTE stream's progress -- the optimizer kernel has
  // already waited for the step's weight gradients and buckets; work it forks for the next step, like the rebuild of the
  // packed weights, overlaps that step and is covered by the synchronisation that ends the timed region)
  if ((int)ctx->marks.size() <= idx) ctx->marks.resize(idx + 1, nullptr);
  if (!ctx->marks[idx]) MSK_CHECK_HIP(ctx, hipEventCreate(&ctx->marks[idx]));
  MSK_CHECK_HIP(ctx, hipEventRecord(ctx->marks[idx], ctx->stream));
  return 0;
}
int msk_ctx_wait(msk_ctx* ctx, msk_ctx* other) {
  MSK_REQUIRE(ctx, other != nullptr && other->device == ctx->device, "msk_ctx_wait: two contexts on the same device");
  if (other == ctx) return 0;
  MSK_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  if (!ctx->ev_xctx) MSK_CHECK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_xctx, hipEventDisableTiming));
  MSK_CHECK_HIP(ctx, hipEventRecord(ctx->ev_xctx, other->stream));
  MSK_CHECK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_xctx, 0));
  return 0;
}
int msk_mark_elapsed(msk_ctx* ctx, int a, int b, float* ms) {
  MSK_REQUIRE(ctx, a >= 0 && b >= 0 && a < (int)ctx->marks.size() && b < (int)ctx->marks.size() && ctx->marks[a] && ctx->marks[b],
              "mark not recorded");
  MSK_CHECK_HIP(ctx, hipEventSynchronize(ctx->marks[b]));
  MSK_CHECK_HIP(ctx, hipEventElapsedTime(ms, ctx->marks[a], ctx->marks[b]));
  return 0;
}

int msk_prof_enable(msk_ctx* ctx, int on) {
  if (!on) drain_prof(ctx);
  ctx->prof = on != 0;
  return 0;
}
int msk_prof_reset(msk_ctx* ctx) {
  drain_prof(ctx);
  ctx->prof_map.clear();
  return 0;
}
int msk_prof_report(msk_ctx* ctx, char* buf, int buflen, int* len) {
  drain_prof(ctx);
  std::string s;
  char line[256];
  for (auto& kv : ctx->prof_map) {
    snprintf(line, sizeof(line), "%s\t%ld\t%.6f\n", kv.first.c_str(), kv.second.calls, kv.second.total_ms);
    s += line;
  }
  if (len) *len = (int)s.size() + 1;
  if (buf && buflen > 0) {
    strncpy(buf, s.c_str(), buflen - 1);
    buf[buflen - 1] = 0;
  }
  return 0;
}

int msk_get_option(msk_ctx* ctx, const char* key, int* value) {
  MSK_REQUIRE(ctx, key != nullptr && value != nullptr, "msk_get_option: null argument");
  // the EFFECTIVE values (after environment overrides and fall-backs inside msk_dp_init), for callers that must agree with them
  if (strcmp(key, "dp_mode") == 0) *value = ctx->dp_mode;
  else if (strcmp(key, "conv_split") == 0) *value = ctx->conv_split;
  else if (strcmp(key, "wgrad_async") == 0) *value = ctx->wgrad_async ? 1 : 0;
  else if (strcmp(key, "world") == 0) *value = ctx->world;
  else if (strcmp(key, "rank") == 0) *value = ctx->rank;
  // amax arrays handed out so far by the compute stream's ring / the weight-gradient stream's ring (modulo 2^30): an array is
  // zeroed again 512 requests after it was handed out -- a holder that spans a whole step (Tensor.amax: forward -> weight
  // gradient) is safe while a step stays under that (device.Arena.reset checks)
  else if (strcmp(key, "scalar_ring_served") == 0) *value = (int)(ctx->scalar_served & 0x3FFFFFFF);
  else if (strcmp(key, "scalar_ring_served_side") == 0) *value = (int)(ctx->scalar_served_side & 0x3FFFFFFF);
  else return msk_fail(ctx, __FILE__, __LINE__, "msk_get_option", "unknown key");
  return 0;
}
int msk_set_option(msk_ctx* ctx, const char* key, int value) {
  if (strcmp(key, "conv_impl") == 0) {
    ctx->conv_impl = value;
    return 0;
  }
  if (strcmp(key, "halo_tile") == 0) {
    ctx->halo_tile = value;
    return 0;
  }
  if (strcmp(key, "prof_only_halo") == 0) {  // per-kernel profile restricted to the MFMA halo convs (bench.py roofline)
    snprintf(ctx->prof_prefix, sizeof(ctx->prof_prefix), "%s", value ? (ctx->wbf ? "wbf_gemm_" : "conv_halo_") : "");
    return 0;
  }
  if (strcmp(key, "wgrad_async_max_m") == 0) {
    ctx->wgrad_async_max_m = value;
    return 0;
  }
  if (strcmp(key, "wino_bf3") == 0) {  // 0 = fp32-MFMA Winograd kernels instead of the bf16x3 three-stage pipeline
    ctx->wbf = value != 0;
    return 0;
  }
  if (strcmp(key, "conv_fp16") == 0) {
    ctx->conv_fp16 = value != 0;
    return 0;
  }
  if (strcmp(key, "wbf_tin_map") == 0) {
    ctx->wbf_tin_map = value;
    return 0;
  }
  if (strcmp(key, "wbf_mr4") == 0) {
    ctx->wbf_mr4 = value;
    return 0;
  }
  if (strcmp(key, "wbf_variant") == 0) {
    ctx->wbf_variant = value;
    return 0;
  }
  if (strcmp(key, "direct_conv") == 0) {  // 1 = no Winograd kernels (same as env MSEGK_DIRECT_CONV=1)
    ctx->no_winograd = value != 0;
    return 0;
  }
  if (strcmp(key, "wbf_tin_groups") == 0) {  // see wbf_tiles_per_group (msk_conv_wbf.hip); -1 = default
    ctx->wbf_tin_groups = value;
    return 0;
  }
  if (strcmp(key, "wbf_pad_min_voxels") == 0) {  // smallest problem the channel-padding wrapper of the wbf pipeline takes
    ctx->wbf_pad_min_voxels = value > 0 ? value : 0;
    return 0;
  }
  if (strcmp(key, "wbf_pack_cache") == 0) {  // 0 = pack the weights of the Winograd pipelines on every call (A/B)
    ctx->wbf_pack_cache = value;
    return 0;
  }
  if (strcmp(key, "wbf_pack_lds") == 0) {  // 0 = wbf_pack_weights_k (one thread per element) instead of the LDS-staged form (A/B)
    ctx->wbf_pack_lds = value;
    return 0;
  }
  if (strcmp(key, "wbf_bpf") == 0) {  // weight-fragment prefetch depth of wbf_gemm_k: 0 auto (by grid size), 1, 4
    ctx->wbf_bpf = value;
    return 0;
  }
  if (strcmp(key, "wbf_ks_blocks") == 0) {  // tuning: workgroups per CU targeted by the split-K of wbf_gemm_k (default 2; round 5 sweep on one box, two repetitions: 16 / 8 / 4 / 2 = 18.58 / 18.46 / 18.39 / 18.28 ms -- every slab is a round trip of M through HBM and a term of wbf_tout_k)
    ctx->wbf_ks_blocks = value > 0 ? value : 2;
    return 0;
  }
  if (strcmp(key, "wgrad_c1_wpc") == 0) {
    ctx->wgrad_c1_wpc = value < 1 ? 1 : value;
    return 0;
  }
  if (strcmp(key, "eager_tail_main") == 0) {
    ctx->eager_tail_main = value;
    return 0;
  }
  if (strcmp(key, "fork_attach") == 0) {
    ctx->fork_attach = value;
    return 0;
  }
  if (strcmp(key, "prof_paused") == 0) {
    ctx->prof_paused = value;
    return 0;
  }
  if (strcmp(key, "prof_attach") == 0) {
    ctx->prof_attach = value;
    return 0;
  }
  if (strcmp(key, "noop_after_merge") == 0) {  // debug: empty launches behind every bn_stats_merge / sums_merge (measures the price of a tiny launch in the step)
    ctx->noop_after_merge = value;
    return 0;
  }
  if (strcmp(key, "small_pack_cache") == 0) {  // 0 = the non-Winograd kernels pack their weights on every call (A/B; msk_conv.hip SmallPackCache)
    ctx->small_pack_cache = value;
    return 0;
  }
  if (strcmp(key, "wbf_prepack") == 0) {  // 0 = stale packed weights are rebuilt lazily at their next use only
    ctx->wbf_prepack = value;
    return 0;
  }
  if (strcmp(key, "dp_mode") == 0) {  // before msk_dp_init (mode 2 creates its second communicator there); 0 / 1 switch any time
    if (value < 0 || value > 3 || (value == 2 && ctx->comm != nullptr && ctx->comm_grad == nullptr))
      return msk_fail(ctx, __FILE__, __LINE__, "msk_set_option", "dp_mode: 0, 1, 2 or 3 (2 only before msk_dp_init)");
    ctx->dp_mode = value;
    return 0;
  }
  if (strcmp(key, "ew_cap") == 0) {      // tuning: blocks per CU of the elementwise kernels (default 32)
    msk_set_ew_caps(value, 0);
    return 0;
  }
  if (strcmp(key, "dst_split") == 0) {  // 0 = msk_conv3d_bwd_bnact_split never splits (A/B)
    ctx->dst_split = value;
    return 0;
  }
  if (strcmp(key, "wgrad_reduce_rows") == 0) {  // 0 = wbf_wgrad_reduce_k for the deep layers as well (A/B of wbf_wgrad_reduce_rows_k)
    ctx->wgrad_reduce_rows = value;
    return 0;
  }
  if (strcmp(key, "ks_stats") == 0) {  // 0 = BatchNorm statistics of the k == s convolutions' outputs as a pass of their own (A/B)
    ctx->ks_stats = value;
    return 0;
  }
  if (strcmp(key, "ks_lds") == 0) {  // 0 = gconv_ks_fwd_k for the 16-channel k == s problems as well (A/B of gconv_ks_lds_k)
    ctx->ks_lds = value;
    return 0;
  }
  if (strcmp(key, "dense12") == 0) {  // 0 = the scalar elementwise kernels for dense 1/2/3/6-channel tensors (A/B of the float4-triple kernels)
    msk_set_dense12(value);
    return 0;
  }
  if (strcmp(key, "reduce_vpl") == 0) {  // tuning: voxels per lane of the per-channel reduction kernels before more workgroups are added
    msk_set_reduce_vpl(value);
    return 0;
  }
  if (strcmp(key, "reduce_vpl_site") == 0) {  // debug: site * 1000 + voxels per lane (0 = follow reduce_vpl); sites in msk_elementwise.hip
    msk_set_reduce_vpl_site(value);
    return 0;
  }
  if (strcmp(key, "reduce_cap") == 0) {  // tuning: blocks per CU of the per-channel reduction kernels (default 4)
    msk_set_ew_caps(0, value);
    return 0;
  }
  if (strcmp(key, "dy_bound_shift") == 0) {   // debug: scale dy (fused BatchNorm backward, NP = 2) as if its bound were 2^value larger
    ctx->dy_bound_shift = value;
    return 0;
  }
  if (strcmp(key, "late_split") == 0) {
    ctx->late_split = value != 0;
    return 0;
  }
  if (strcmp(key, "c1_h2") == 0) {
    ctx->c1_h2 = value < 0 ? 0 : (value > 2 ? 2 : value);
    return 0;
  }
  if (strcmp(key, "ks_nr_max") == 0) {
    ctx->ks_nr_max = value >= 4 ? 4 : (value >= 2 ? 2 : 1);
    return 0;
  }
  if (strcmp(key, "wgrad_lds_pad") == 0) {
    ctx->wgrad_lds_pad = value > 0 ? value : 0;
    return 0;
  }
  if (strcmp(key, "kst_pair") == 0) {
    ctx->kst_pair = value != 0;
    return 0;
  }
  if (strcmp(key, "tile_staging") == 0) {
    ctx->tile_staging = value;   // bit 0: the 1x1x1 head, bit 1: the 5..32-class loss kernels, bit 2: the 2..4-class loss kernels
    return 0;
  }
  if (strcmp(key, "wgrad_renorm") == 0) {
    ctx->wgrad_renorm = value != 0;
    return 0;
  }
  if (strcmp(key, "ks_legacy") == 0) {
    ctx->ks_legacy = value;
    return 0;
  }
  if (strcmp(key, "wgrad_fork") == 0) {
    ctx->wgrad_fork = value;
    return 0;
  }
  if (strcmp(key, "wbf_fuse") == 0) {
    ctx->wbf_fuse = value;
    return 0;
  }
  if (strcmp(key, "wbf_fused_bl") == 0) {
    ctx->wbf_fused_bl = value;
    return 0;
  }
  if (strcmp(key, "wbf_tpb") == 0) {
    ctx->wbf_tpb = value > 0 ? value : 0;
    return 0;
  }
  if (strcmp(key, "conv_split") == 0) {
    ctx->conv_split = value == 2 ? 2 : 3;
    return 0;
  }
  if (strcmp(key, "bwd_fuse") == 0) {
    ctx->bwd_fuse = value < 0 ? -1 : (value > 2 ? 2 : value);
    return 0;
  }
  if (strcmp(key, "foldn_wgs") == 0) {
    ctx->foldn_wgs = value > 0 ? value : 0;
    return 0;
  }
  if (strcmp(key, "wgrad_wino_rounds") == 0) {
    ctx->wgrad_wino_rounds = value > 0 ? value : 0;
    return 0;
  }
  if (strcmp(key, "wgrad_rounds") == 0) {
    ctx->wgrad_rounds = value > 0 ? value : 8;
    return 0;
  }
  if (strcmp(key, "wgrad_chunk") == 0) {
    ctx->wgrad_chunk = value;
    return 0;
  }
  if (strcmp(key, "wgrad_async") == 0) {  // weight gradients on the side stream (default on)
    if (msk_join_side_impl(ctx) != 0) return -1;
    ctx->wgrad_async = value != 0;
    return 0;
  }
  if (strcmp(key, "prof_shapes") == 0) {
    ctx->prof_shapes = value != 0;
    return 0;
  }
  if (strcmp(key, "poison_scratch") == 0) {  // debug: fill (re)allocated scratch with this byte; -1 = off
    ctx->poison = value;
    if (value >= 0) {
      hipStreamSynchronize(ctx->stream);
      if (ctx->ws) hipMemsetAsync(ctx->ws, value, ctx->ws_bytes, ctx->stream);
      if (ctx->ws2) hipMemsetAsync(ctx->ws2, value, ctx->ws2_bytes, ctx->stream);
    }
    return 0;
  }
  return msk_fail(ctx, __FILE__, __LINE__, "msk_set_option", "unknown key");
}

}  // extern "C"
