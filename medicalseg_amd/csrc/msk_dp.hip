// Data-parallel collectives: RCCL over xGMI, one process per GPU.
// Replaces paddle.distributed.fleet DataParallel / SyncBatchNorm communication
// (core/train.py:81-85, cvlibs/config.py:322).  All collectives are enqueued on the
// context's compute stream so they order with the kernels that produce/consume them.
#include <rccl/rccl.h>
#include <unistd.h>

#include <cstdio>

#include "msk_common.h"

#define MSK_CHECK_NCCL(ctx, expr)                                                        \
  do {                                                                                   \
    ncclResult_t _r = (expr);                                                            \
    if (_r != ncclSuccess) return msk_fail(ctx, __FILE__, __LINE__, #expr, ncclGetErrorString(_r)); \
  } while (0)

namespace {
// RCCL 2.27 prints a version banner ("RCCL version : ...", 5 lines) to STDOUT when a communicator is created; through
// a pipe it sits in the C stdio buffer until exit, i.e. it would land after bench.py's single JSON line.  While RCCL
// initialises, fd 1 points at stderr, and the buffer is flushed before fd 1 is restored.
struct StdoutToStderr {
  int saved;
  StdoutToStderr() {
    fflush(stdout);
    saved = dup(1);
    if (saved >= 0) dup2(2, 1);
  }
  ~StdoutToStderr() {
    fflush(stdout);
    if (saved >= 0) {
      dup2(saved, 1);
      close(saved);
    }
  }
};
}  // namespace

extern "C" {

int msk_dp_unique_id(char* id128) {
  StdoutToStderr quiet;
  static_assert(sizeof(ncclUniqueId) <= MSK_UNIQUE_ID_BYTES, "unique id size");
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return msk_fail(nullptr, __FILE__, __LINE__, "ncclGetUniqueId", ncclGetErrorString(r));
  memset(id128, 0, MSK_UNIQUE_ID_BYTES);
  memcpy(id128, &id, sizeof(id));
  return 0;
}

int msk_dp_init(msk_ctx* ctx, const char* id128, int rank, int world) {
  MSK_REQUIRE(ctx, ctx->comm == nullptr, "communicator already initialised");
  MSK_REQUIRE(ctx, world >= 1 && rank >= 0 && rank < world, "bad rank/world");
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  MSK_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  StdoutToStderr quiet;
  ncclComm_t comm;
  MSK_CHECK_NCCL(ctx, ncclCommInitRank(&comm, world, id, rank));
  ctx->comm = (void*)comm;
  ctx->rank = rank;
  ctx->world = world;
  // second communicator + stream for the gradient buckets (collective: every rank is inside msk_dp_init)
  ncclComm_t comm_grad = nullptr;
  if (ncclCommSplit(comm, 0, rank, &comm_grad, nullptr) != ncclSuccess || comm_grad == nullptr) {
    // no second communicator: the buckets then go through the first one ON THE COMPUTE STREAM (correct, not
    // overlapped) -- see msk_dp_allreduce_async
    comm_grad = nullptr;
  }
  ctx->comm_grad = (void*)comm_grad;
  MSK_CHECK_HIP(ctx, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
  MSK_CHECK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_comm_main, hipEventDisableTiming));
  MSK_CHECK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_comm_side, hipEventDisableTiming));
  MSK_CHECK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_comm_done, hipEventDisableTiming));
  ctx->comm_pending = false;
  return 0;
}

int msk_dp_allreduce_sum(msk_ctx* ctx, float* buf, size_t count) {
  MSK_REQUIRE(ctx, ctx->comm != nullptr, "msk_dp_init not called");
  if (msk_join_side_impl(ctx) != 0) return -1;  // the gradient arena includes side-stream weight gradients
  msk_launch_scope ls(ctx, "rccl_allreduce");
  MSK_CHECK_NCCL(ctx, ncclAllReduce(buf, buf, count, ncclFloat, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
  return 0;
}

int msk_dp_allreduce_async(msk_ctx* ctx, float* buf, size_t count) {
  MSK_REQUIRE(ctx, ctx->comm != nullptr, "msk_dp_init not called");
  if (count == 0) return 0;
  if (ctx->comm_grad == nullptr) return msk_dp_allreduce_sum(ctx, buf, count);  // ncclCommSplit was not available
  // the bucket's gradients come from the compute stream (data-gradient chain, bias/BN/PReLU gradients) and from the
  // weight-gradient side stream: wait for the current tail of both, block neither
  MSK_CHECK_HIP(ctx, hipEventRecord(ctx->ev_comm_main, ctx->stream));
  MSK_CHECK_HIP(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->ev_comm_main, 0));
  if (ctx->side_dirty) {
    MSK_CHECK_HIP(ctx, hipEventRecord(ctx->ev_comm_side, ctx->side));
    MSK_CHECK_HIP(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->ev_comm_side, 0));
  }
  MSK_CHECK_NCCL(ctx, ncclAllReduce(buf, buf, count, ncclFloat, ncclSum, (ncclComm_t)ctx->comm_grad, ctx->comm_stream));
  ctx->comm_pending = true;
  return 0;
}

int msk_dp_wait(msk_ctx* ctx) { return msk_dp_wait_impl(ctx); }

int msk_dp_allreduce_stats(msk_ctx* ctx, float* buf, size_t count) {
  // per-channel BatchNorm-backward sums: produced on the main stream, so the weight-gradient side stream
  // keeps running (msk_dp_allreduce_sum would join it 24 times per step and undo the overlap)
  MSK_REQUIRE(ctx, ctx->comm != nullptr, "msk_dp_init not called");
  msk_launch_scope ls(ctx, "rccl_allreduce_stats");
  MSK_CHECK_NCCL(ctx, ncclAllReduce(buf, buf, count, ncclFloat, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
  return 0;
}

int msk_dp_allgather(msk_ctx* ctx, const float* send, float* recv, size_t count_per_rank) {
  MSK_REQUIRE(ctx, ctx->comm != nullptr, "msk_dp_init not called");
  msk_launch_scope ls(ctx, "rccl_allgather");
  MSK_CHECK_NCCL(ctx, ncclAllGather(send, recv, count_per_rank, ncclFloat, (ncclComm_t)ctx->comm, ctx->stream));
  return 0;
}

int msk_dp_broadcast(msk_ctx* ctx, float* buf, size_t count, int root) {
  MSK_REQUIRE(ctx, ctx->comm != nullptr, "msk_dp_init not called");
  msk_launch_scope ls(ctx, "rccl_broadcast");
  MSK_CHECK_NCCL(ctx, ncclBroadcast(buf, buf, count, ncclFloat, root, (ncclComm_t)ctx->comm, ctx->stream));
  return 0;
}

int msk_dp_barrier(msk_ctx* ctx) {
  MSK_REQUIRE(ctx, ctx->comm != nullptr, "msk_dp_init not called");
  float* tok = (float*)msk_workspace(ctx, 256);
  if (!tok) return -1;
  MSK_CHECK_NCCL(ctx, ncclAllReduce(tok, tok, 1, ncclFloat, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
  MSK_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int msk_dp_destroy(msk_ctx* ctx) {
  if (ctx && ctx->comm_stream) {
    hipStreamSynchronize(ctx->comm_stream);
    if (ctx->comm_grad) ncclCommDestroy((ncclComm_t)ctx->comm_grad);
    ctx->comm_grad = nullptr;
    hipStreamDestroy(ctx->comm_stream);
    hipEventDestroy(ctx->ev_comm_main);
    hipEventDestroy(ctx->ev_comm_side);
    hipEventDestroy(ctx->ev_comm_done);
    ctx->comm_stream = nullptr;
    ctx->comm_pending = false;
  }
  if (ctx && ctx->comm) {
    ncclCommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->world = 1;
    ctx->rank = 0;
  }
  return 0;
}

}  // extern "C"
