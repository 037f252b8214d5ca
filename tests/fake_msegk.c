/* Host-only STAND-IN for libmsegk.so used by tests/test_host_dryrun.py to exercise the
 * Python plumbing (ctypes signatures, shape logic, arena, train loop control flow) without
 * a GPU.  It computes NOTHING: compute entry points are no-ops returning 0, memory entry
 * points use host malloc/memcpy.  It is never built into or loaded by the product. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

typedef struct { int dummy; } fake_ctx;
static fake_ctx g_ctx;
static long g_calls = 0;

long fake_calls(void) { return g_calls; }
int msk_version(void) { return -1; }
static int fake_env_int(const char* k, int dflt);
int msk_device_count(int* c) { *c = fake_env_int("FAKE_DEVICE_COUNT", fake_env_int("WORLD_SIZE", 1)); return 0; }
/* Fault injection for tests/test_launch_watchdog.py (the supervisor of launch.py must turn each of these into ONE JSON line):
 *   FAKE_FAULT=hang_ctx         the rank never gets past msk_ctx_create (never reaches the rendezvous)
 *   FAKE_FAULT=hang_collective  the rank blocks forever in its first collective (statistics exchange / gradient all-reduce)
 *   FAKE_FAULT=exit_collective  the rank exits with code 3 there
 * on rank FAKE_FAULT_RANK (env RANK), in attempts <= FAKE_FAULT_LAST_ATTEMPT (env MSEGK_ATTEMPT; default: every attempt), and,
 * for the collective faults, only while the arrangement is FAKE_FAULT_DP_MODE (default: any). */
static int fake_dp_mode = 0;
static int fake_env_int(const char* k, int dflt) { const char* v = getenv(k); return v && v[0] ? atoi(v) : dflt; }
static void fake_fault(const char* where) {
  const char* f = getenv("FAKE_FAULT");
  if (!f) return;
  if (fake_env_int("RANK", 0) != fake_env_int("FAKE_FAULT_RANK", 1)) return;
  if (fake_env_int("MSEGK_ATTEMPT", 0) > fake_env_int("FAKE_FAULT_LAST_ATTEMPT", 1 << 30)) return;
  if (strcmp(where, "ctx") == 0) {
    if (strcmp(f, "hang_ctx") == 0) for (;;) sleep(1);
    return;
  }
  const int m = fake_env_int("FAKE_FAULT_DP_MODE", -1);
  if (m >= 0 && m != fake_dp_mode) return;
  if (strcmp(f, "hang_collective") == 0) for (;;) sleep(1);
  if (strcmp(f, "exit_collective") == 0) _exit(3);
}
int msk_ctx_create(int dev, void** out) { (void)dev; fake_fault("ctx"); *out = &g_ctx; return 0; }
int msk_ctx_destroy(void* c) { (void)c; return 0; }
const char* msk_last_error(void* c) { (void)c; return "fake"; }
int msk_sync(void* c) { (void)c; return 0; }
int msk_join_side(void* c) { (void)c; return 0; }
int msk_device_name(void* c, char* buf, int n) { (void)c; strncpy(buf, "fake-host-device", n); return 0; }
int msk_device_pci_bus_id(void* c, char* buf, int n) { (void)c; strncpy(buf, "0000:00:00.0", n); return 0; }
int msk_dp_rccl_version(int* v) { *v = 0; return 0; }
int msk_malloc(void* c, size_t b, void** out) { (void)c; *out = calloc(1, b ? b : 16); return *out ? 0 : -1; }
int msk_free(void* c, void* p) { (void)c; free(p); return 0; }
int msk_memset(void* c, void* p, int v, size_t b) { (void)c; memset(p, v, b); return 0; }
int msk_weights_changed(void* c, const void* p, size_t b) { (void)c; (void)p; (void)b; return 0; }
int msk_h2d(void* c, void* d, const void* s, size_t b) { (void)c; memcpy(d, s, b); return 0; }
int msk_d2h(void* c, void* d, const void* s, size_t b) { (void)c; memcpy(d, s, b); return 0; }
int msk_h2d_async(void* c, void* d, const void* s, size_t b) { (void)c; memcpy(d, s, b); return 0; }
int msk_d2d(void* c, void* d, const void* s, size_t b) { (void)c; memmove(d, s, b); return 0; }
int msk_pinned_alloc(void* c, size_t b, void** out) { return msk_malloc(c, b, out); }
int msk_pinned_free(void* c, void* p) { return msk_free(c, p); }
int msk_mem_info(void* c, size_t* f, size_t* t) { (void)c; *f = *t = (size_t)1 << 38; return 0; }
int msk_timer_start(void* c) { (void)c; return 0; }
int msk_mark(void* c, int i) { (void)c; (void)i; return 0; }
int msk_ctx_wait(void* c, void* o) { (void)c; (void)o; return 0; }
int msk_mark_elapsed(void* c, int a, int b, float* ms) { (void)c; (void)a; (void)b; *ms = 1.0f; return 0; }
int msk_timer_stop(void* c, float* ms) { (void)c; *ms = 1.0f; return 0; }
int msk_prof_enable(void* c, int on) { (void)c; (void)on; return 0; }
int msk_prof_reset(void* c) { (void)c; return 0; }
int msk_prof_report(void* c, char* buf, int n, int* len) { (void)c; if (buf && n > 0) buf[0] = 0; if (len) *len = 1; return 0; }
int msk_set_option(void* c, const char* k, int v) { (void)c; if (strcmp(k, "dp_mode") == 0) fake_dp_mode = v; return 0; }
int msk_get_option(void* c, const char* k, int* v) { (void)c; *v = strcmp(k, "dp_mode") == 0 ? fake_dp_mode : 0; return 0; }
int msk_dp_unique_id(char* id) { memset(id, 7, 128); return 0; }
size_t msk_conv3d_xform_bytes() { return 0; }
size_t msk_conv3d_bwd_bnact_bytes() { return 0; }
#define NOOP(name) int name() { ++g_calls; return 0; }
NOOP(msk_ncdhw_to_ndhwc) NOOP(msk_ndhwc_to_ncdhw)
NOOP(msk_conv3d_fwd_ex2) NOOP(msk_affine_act_fwd_amax) NOOP(msk_affine_act_fwd_amax2) NOOP(msk_conv3d_bwd_bnact_split) NOOP(msk_conv3d_bwd_bnact_acc) NOOP(msk_affine_act_join_fwd_amax) NOOP(msk_copy_scale_amax) NOOP(msk_conv3d_dgrad_ex) NOOP(msk_conv3d_bwd_bnact_c1) NOOP(msk_convT3d_bwd_bnact) NOOP(msk_conv3d_bwd_inact) NOOP(msk_conv3d_wgrad_ex2) NOOP(msk_conv3d_wgrad_ex3) NOOP(msk_affine_act_bwd_apply_amax)
static float fake_amax[128];
float* msk_amax_new(void* c, int n) { (void)c; (void)n; return fake_amax; }
NOOP(msk_conv3d_fwd_ex3) NOOP(msk_conv3d_fwd_in) NOOP(msk_bn_stats_fin) NOOP(msk_affine_act_bwd_reduce_pg) NOOP(msk_add_act_join_bwd_pg)
NOOP(msk_conv3d_fwd) NOOP(msk_conv_fold_bn) NOOP(msk_conv3d_fwd_act) NOOP(msk_conv3d_fwd_ex) NOOP(msk_conv3d_wgrad_ex) NOOP(msk_conv3d_bwd_bnact) NOOP(msk_conv3d_dgrad) NOOP(msk_conv3d_wgrad)
NOOP(msk_convT3d_fwd) NOOP(msk_convT3d_fwd_ex) NOOP(msk_convT3d_dgrad) NOOP(msk_convT3d_wgrad)
NOOP(msk_bn_stats) NOOP(msk_bn_finalize) NOOP(msk_bn_eval_coeffs)
NOOP(msk_affine_act_fwd) NOOP(msk_affine_act_bwd_reduce) NOOP(msk_affine_act_bwd_reduce_ex) NOOP(msk_affine_act_join_fwd) NOOP(msk_add_act_join_bwd) NOOP(msk_add_act_join_bwd_ex) NOOP(msk_affine_act_bwd_apply) NOOP(msk_affine_act_param_grads)
NOOP(msk_add_act_bwd) NOOP(msk_bn_bias_grad) NOOP(msk_copy_scale) NOOP(msk_dropout_mask) NOOP(msk_channel_sum) NOOP(msk_argmax_c) NOOP(msk_softmax_c)
NOOP(msk_class_weights) NOOP(msk_loss_fwd) NOOP(msk_loss_bwd) NOOP(msk_sgd_momentum) NOOP(msk_sgd_momentum_eager) NOOP(msk_sgd_momentum_finish) NOOP(msk_loss_fwd_ex) NOOP(msk_loss_bwd_ex) NOOP(msk_adam) NOOP(msk_elu_fwd) NOOP(msk_elu_bwd)
NOOP(msk_resample3d) NOOP(msk_hu_norm) NOOP(msk_minmax_norm) NOOP(msk_max_norm) NOOP(msk_label_remap)
NOOP(msk_crop_resample3d) NOOP(msk_flip3d) NOOP(msk_rotate3d)
NOOP(msk_interp_trilinear_fwd) NOOP(msk_interp_trilinear_bwd)
typedef struct { void* p; int32_t n, d, h, w, c, ld; } fake_tensor;
int msk_interp_scratch_bytes(void* c, fake_tensor s, fake_tensor d, size_t* b) {
  (void)c; *b = ((size_t)d.n * d.d * d.h * s.w + (size_t)d.n * d.d * s.h * s.w) * s.c * 4; return 0;
}
#define FAULTY(name) int name() { ++g_calls; fake_fault("collective"); return 0; }
NOOP(msk_dp_init) FAULTY(msk_dp_allreduce_sum) FAULTY(msk_dp_allreduce_stats) FAULTY(msk_dp_allreduce_async) NOOP(msk_dp_wait) NOOP(msk_dp_allgather) NOOP(msk_dp_broadcast) NOOP(msk_dp_barrier) NOOP(msk_dp_destroy)
