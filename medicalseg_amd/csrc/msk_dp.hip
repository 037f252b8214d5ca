// Data-parallel collectives: RCCL over xGMI, one process per GPU.
// Replaces paddle.distributed.fleet DataParallel / SyncBatchNorm communication
// (core/train.py:81-85, cvlibs/config.py:322).
//
// Stream / communicator arrangements (ctx->dp_mode, option "dp_mode", env MSEGK_DP_MODE), all MEASURED on one MI355X with a
// 1-rank RCCL communicator and the 48 SyncBatchNorm collectives of a VNet step forced on (bench.py
// --force-syncbn-collectives; plain step 20.7 ms):
//   0 (default)  every collective on the compute stream, ONE communicator: a total order by construction, nothing to
//                deadlock; the gradient arena is reduced by one all-reduce after backward.           21.3 ms (+0.6)
//   2            gradient buckets on a second communicator (ncclCommSplit) and the communication stream, overlapped with
//                backward; statistics on the compute stream's communicator.  Two communicators executing concurrently are
//                deadlock-free only while their kernels can be co-resident (they are small next to 256 CUs, but nothing
//                enforces it): opt-in.                                                                21.5 ms (+0.8)
//   1            the round-2 review's proposal: ONE communicator on ONE communication stream carries statistics exchanges
//                (the compute stream hands over through an event and waits for the result) and buckets in program order.
//                Correct (bit-identical parameters, tests/test_gpu_dp.py) -- and 42.6 ms per step: +22 ms.
//   3            one communicator, statistics on the compute stream, buckets on the communication stream, a statistics
//                exchange first waits for the bucket in flight (<= 10 stream alternations per step): 35.4 ms (+15 ms).
// A bare event hand-over between two streams costs 24 us here (tools/probes/xstream_probe.hip, the same with a 1-rank RCCL
// call in between: tools/probes/xstream_rccl_probe.hip), so 48 of them would be ~1 ms; what modes 1 and 3 pay is RCCL's
// handling of a communicator that is driven from a second stream inside a deep asynchronous launch queue.  Until that can
// be examined on a multi-GPU node the default is the arrangement with the fewest moving parts.
#include <rccl/rccl.h>
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>

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
#ifdef MSK_TEST_TRANSPORT   // compiled into libmsegk_test.so only (build.sh); the release library has no such path
// ---------------------------------------------------------------------------------------------------------
// Host transport (env MSEGK_DP_TRANSPORT=host): the same collectives through pinned-less host staging and TCP sockets
// in a star around rank 0, which reduces in rank order (deterministic).  It exists so that the N-rank code paths --
// SyncBatchNorm exchanges, gradient buckets, 1/nranks, broadcast -- can run with SEVERAL PROCESSES ON ONE GPU (RCCL
// refuses two ranks on one device), i.e. in the single-GPU test tier; it is never selected implicitly and is orders of
// magnitude slower than RCCL over xGMI.
// ---------------------------------------------------------------------------------------------------------
bool send_all(int fd, const void* p, size_t n) {
  const char* c = (const char*)p;
  while (n > 0) {
    const ssize_t k = send(fd, c, n, MSG_NOSIGNAL);
    if (k <= 0) return false;
    c += k;
    n -= (size_t)k;
  }
  return true;
}
bool recv_all(int fd, void* p, size_t n) {
  char* c = (char*)p;
  while (n > 0) {
    const ssize_t k = recv(fd, c, n, 0);
    if (k <= 0) return false;
    c += k;
    n -= (size_t)k;
  }
  return true;
}

int host_connect(msk_ctx* ctx, int rank, int world) {
  const char* pe = getenv("MASTER_PORT");
  const char* ae = getenv("MASTER_ADDR");
  const int port = (pe ? atoi(pe) : 29500) + 40;
  sockaddr_in sa{};
  sa.sin_family = AF_INET;
  sa.sin_port = htons((uint16_t)port);
  if (inet_pton(AF_INET, (ae && ae[0] && strcmp(ae, "localhost") != 0) ? ae : "127.0.0.1", &sa.sin_addr) != 1)
    return msk_fail(ctx, __FILE__, __LINE__, "host transport", "MASTER_ADDR must be an IPv4 address");
  ctx->host_fds.assign(world, -1);
  const int one = 1;
  if (rank == 0) {
    const int srv = socket(AF_INET, SOCK_STREAM, 0);
    setsockopt(srv, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    if (bind(srv, (sockaddr*)&sa, sizeof(sa)) != 0 || listen(srv, world) != 0) {
      close(srv);
      return msk_fail(ctx, __FILE__, __LINE__, "host transport", "cannot listen on MASTER_PORT + 40");
    }
    for (int k = 1; k < world; ++k) {
      const int fd = accept(srv, nullptr, nullptr);
      int peer = -1;
      if (fd < 0 || !recv_all(fd, &peer, sizeof(peer)) || peer <= 0 || peer >= world || ctx->host_fds[peer] >= 0) {
        close(srv);
        return msk_fail(ctx, __FILE__, __LINE__, "host transport", "bad peer during connect");
      }
      setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
      ctx->host_fds[peer] = fd;
    }
    close(srv);
  } else {
    int fd = -1;
    for (int attempt = 0; attempt < 600 && fd < 0; ++attempt) {   // up to ~60 s for rank 0 to come up
      fd = socket(AF_INET, SOCK_STREAM, 0);
      if (connect(fd, (sockaddr*)&sa, sizeof(sa)) != 0) {
        close(fd);
        fd = -1;
        usleep(100000);
      }
    }
    if (fd < 0 || !send_all(fd, &rank, sizeof(rank)))
      return msk_fail(ctx, __FILE__, __LINE__, "host transport", "cannot reach rank 0");
    setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
    ctx->host_fds[0] = fd;
  }
  return 0;
}

// gather everybody's `count` floats on rank 0 ([world][count]); op 0: all-reduce sum in rank order, op 1: all-gather
int host_collective(msk_ctx* ctx, const float* dsend, float* drecv, size_t count, int op, hipStream_t stream) {
  MSK_CHECK_HIP(ctx, hipStreamSynchronize(stream));
  const int W = ctx->world;
  std::vector<float> mine(count), out(op == 0 ? count : count * W);
  MSK_CHECK_HIP(ctx, hipMemcpy(mine.data(), dsend, count * sizeof(float), hipMemcpyDeviceToHost));
  const size_t out_bytes = out.size() * sizeof(float);
  const unsigned long long tag = (unsigned long long)count * 2 + (unsigned)op;  // travels ahead of every payload
  if (ctx->rank == 0) {
    std::vector<float> tmp(count);
    if (op == 0) out = mine;
    else memcpy(out.data(), mine.data(), count * sizeof(float));
    for (int r = 1; r < W; ++r) {
      unsigned long long peer_tag = 0;
      if (!recv_all(ctx->host_fds[r], &peer_tag, sizeof(peer_tag)))
        return msk_fail(ctx, __FILE__, __LINE__, "host transport", "peer closed during a collective");
      if (peer_tag != tag)   // ranks disagree on the collective sequence: fail loudly instead of exchanging garbage
        return msk_fail(ctx, __FILE__, __LINE__, "host transport", "collective mismatch between ranks (kind or element count)");
      if (!recv_all(ctx->host_fds[r], tmp.data(), count * sizeof(float)))
        return msk_fail(ctx, __FILE__, __LINE__, "host transport", "peer closed during a collective");
      if (op == 0) for (size_t i = 0; i < count; ++i) out[i] += tmp[i];
      else memcpy(out.data() + (size_t)r * count, tmp.data(), count * sizeof(float));
    }
    for (int r = 1; r < W; ++r)
      if (!send_all(ctx->host_fds[r], out.data(), out_bytes))
        return msk_fail(ctx, __FILE__, __LINE__, "host transport", "peer closed during a collective");
  } else {
    if (!send_all(ctx->host_fds[0], &tag, sizeof(tag)) || !send_all(ctx->host_fds[0], mine.data(), count * sizeof(float)) ||
        !recv_all(ctx->host_fds[0], out.data(), out_bytes))
      return msk_fail(ctx, __FILE__, __LINE__, "host transport", "rank 0 closed during a collective");
  }
  MSK_CHECK_HIP(ctx, hipMemcpy(drecv, out.data(), out_bytes, hipMemcpyHostToDevice));
  return 0;
}
int host_broadcast(msk_ctx* ctx, float* buf, size_t count, int root) {
  if (ctx->world == 1) return 0;
  MSK_REQUIRE(ctx, root == 0, "host transport broadcasts from rank 0 only");
  std::vector<float> h(count);
  MSK_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->rank == 0) {
    MSK_CHECK_HIP(ctx, hipMemcpy(h.data(), buf, count * sizeof(float), hipMemcpyDeviceToHost));
    for (int r = 1; r < ctx->world; ++r)
      if (!send_all(ctx->host_fds[r], h.data(), count * sizeof(float)))
        return msk_fail(ctx, __FILE__, __LINE__, "host transport", "peer closed during broadcast");
  } else {
    if (!recv_all(ctx->host_fds[0], h.data(), count * sizeof(float)))
      return msk_fail(ctx, __FILE__, __LINE__, "host transport", "rank 0 closed during broadcast");
    MSK_CHECK_HIP(ctx, hipMemcpy(buf, h.data(), count * sizeof(float), hipMemcpyHostToDevice));
  }
  return 0;
}
#else
int host_broadcast(msk_ctx* ctx, float*, size_t, int) {
  return msk_fail(ctx, __FILE__, __LINE__, "host transport", "not compiled into this library");
}
int host_connect(msk_ctx* ctx, int, int) {
  return msk_fail(ctx, __FILE__, __LINE__, "MSEGK_DP_TRANSPORT=host", "the host transport is a test facility: it is compiled into libmsegk_test.so only");
}
int host_collective(msk_ctx* ctx, const float*, float*, size_t, int, hipStream_t) {
  return msk_fail(ctx, __FILE__, __LINE__, "host transport", "not compiled into this library");
}
#endif

// Mode 1: a collective that the compute stream needs the result of.  The communication stream picks up the current tail
// of the compute stream, runs the collective, and the compute stream waits for it.
struct CommBridge {
  msk_ctx* ctx;
  hipStream_t run;   // the stream the collective is enqueued on
  bool bridged;
  explicit CommBridge(msk_ctx* c) : ctx(c), run(c->stream), bridged(false) {
    if (c->dp_mode == 3 && c->comm_stream != nullptr && !c->host_transport) {
      // mode 3: the collective runs on the compute stream; a gradient bucket still in flight on the communication stream
      // (same communicator) has to finish first -- one cross-stream wait per bucket, not per collective
      msk_dp_wait_impl(c);
      return;
    }
    if (c->dp_mode == 1 && c->comm_stream != nullptr && !c->host_transport) {
      hipEventRecord(c->ev_comm_main, c->stream);
      hipStreamWaitEvent(c->comm_stream, c->ev_comm_main, 0);
      run = c->comm_stream;
      bridged = true;
    }
  }
  ~CommBridge() {
    if (bridged) {
      hipEventRecord(ctx->ev_comm_back, ctx->comm_stream);
      hipStreamWaitEvent(ctx->stream, ctx->ev_comm_back, 0);
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

int msk_dp_rccl_version(int* version) {
  if (!version) return -1;
  *version = 0;
  return ncclGetVersion(version) == ncclSuccess ? 0 : -1;
}

int msk_dp_init(msk_ctx* ctx, const char* id128, int rank, int world) {
  MSK_REQUIRE(ctx, ctx->comm == nullptr, "communicator already initialised");
  MSK_REQUIRE(ctx, world >= 1 && rank >= 0 && rank < world, "bad rank/world");
  MSK_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  {
    const char* tr = getenv("MSEGK_DP_TRANSPORT");
    if (tr && strcmp(tr, "host") == 0) {
      ctx->rank = rank;
      ctx->world = world;
      if (world > 1 && host_connect(ctx, rank, world) != 0) return -1;
      ctx->host_transport = true;
      ctx->comm = (void*)ctx;   // non-null marker: "initialised"
      return 0;
    }
  }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  StdoutToStderr quiet;
  ncclComm_t comm;
  MSK_CHECK_NCCL(ctx, ncclCommInitRank(&comm, world, id, rank));
  ctx->comm = (void*)comm;
  ctx->rank = rank;
  ctx->world = world;
  // mode 2 only: second communicator for the gradient buckets (collective: every rank is inside msk_dp_init)
  {
    const char* me = getenv("MSEGK_DP_MODE");
    if (me && me[0] >= '0' && me[0] <= '3') ctx->dp_mode = me[0] - '0';
  }
  ncclComm_t comm_grad = nullptr;
  if (ctx->dp_mode == 2) {
    const ncclResult_t sr = ncclCommSplit(comm, 0, rank, &comm_grad, nullptr);
    if (sr != ncclSuccess || comm_grad == nullptr) {
      // no second communicator: back to the arrangement with the fewest moving parts (mode 0: one all-reduce after backward on
      // the compute stream) -- NOT to the single-communicator variants, which cost +15 ... +22 ms per step where they were
      // measured (header comment); the effective mode is what msk_get_option("dp_mode") and bench.py's dp.dp_mode report
      fprintf(stderr, "[msegk] rank %d: dp_mode 2 requested but ncclCommSplit failed (%s): falling back to dp_mode 0\n", rank,
              sr != ncclSuccess ? ncclGetErrorString(sr) : "null communicator");
      comm_grad = nullptr;
      ctx->dp_mode = 0;
    }
  }
  ctx->comm_grad = (void*)comm_grad;
  MSK_CHECK_HIP(ctx, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
  MSK_CHECK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_comm_main, hipEventDisableTiming));
  MSK_CHECK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_comm_side, hipEventDisableTiming));
  MSK_CHECK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_comm_done, hipEventDisableTiming));
  MSK_CHECK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_comm_back, hipEventDisableTiming));
  ctx->comm_pending = false;
  return 0;
}

int msk_dp_allreduce_sum(msk_ctx* ctx, float* buf, size_t count) {
  msk_weights_changed_impl(ctx, buf, count * sizeof(float));
  MSK_REQUIRE(ctx, ctx->comm != nullptr, "msk_dp_init not called");
  if (msk_join_side_impl(ctx) != 0) return -1;  // the gradient arena includes side-stream weight gradients
  msk_launch_scope ls(ctx, "rccl_allreduce");
  if (ctx->host_transport) return ctx->world > 1 ? host_collective(ctx, buf, buf, count, 0, ctx->stream) : 0;
  CommBridge br(ctx);
  MSK_CHECK_NCCL(ctx, ncclAllReduce(buf, buf, count, ncclFloat, ncclSum, (ncclComm_t)ctx->comm, br.run));
  return 0;
}

int msk_dp_allreduce_async(msk_ctx* ctx, float* buf, size_t count) {
  MSK_REQUIRE(ctx, ctx->comm != nullptr, "msk_dp_init not called");
  if (count == 0) return 0;
  if (ctx->host_transport || ctx->dp_mode == 0 || ctx->comm_stream == nullptr) return msk_dp_allreduce_sum(ctx, buf, count);
  // modes 1 and 3: the single communicator on the communication stream; mode 2: the second communicator
  ncclComm_t bucket_comm = (ncclComm_t)(ctx->dp_mode == 2 ? ctx->comm_grad : ctx->comm);
  // the bucket's gradients come from the compute stream (data-gradient chain, bias/BN/PReLU gradients) and from the
  // weight-gradient side stream: wait for the current tail of both, block neither
  MSK_CHECK_HIP(ctx, hipEventRecord(ctx->ev_comm_main, ctx->stream));
  MSK_CHECK_HIP(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->ev_comm_main, 0));
  if (ctx->side_dirty) {
    MSK_CHECK_HIP(ctx, hipEventRecord(ctx->ev_comm_side, ctx->side));
    MSK_CHECK_HIP(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->ev_comm_side, 0));
  }
  msk_weights_changed_impl(ctx, buf, count * sizeof(float));
  // HIP events on the COMMUNICATION stream (the launch scope's events sit on the compute stream): the bucket's own
  // duration incl. the wait for the slowest peer, reported by bench.py as dp.per_rank_collective_ms_per_step
  msk_pending_event pe;
  const bool prof = ctx->prof;
  if (prof) {
    pe.a = msk_prof_event(ctx);
    pe.b = msk_prof_event(ctx);
    pe.tag = "rccl_allreduce_bucket";
    hipEventRecord(pe.a, ctx->comm_stream);
  }
  MSK_CHECK_NCCL(ctx, ncclAllReduce(buf, buf, count, ncclFloat, ncclSum, bucket_comm, ctx->comm_stream));
  if (prof) {
    hipEventRecord(pe.b, ctx->comm_stream);
    ctx->prof_pending.push_back(pe);
  }
  ctx->comm_pending = true;
  return 0;
}

int msk_dp_wait(msk_ctx* ctx) { return msk_dp_wait_impl(ctx); }

int msk_dp_allreduce_stats(msk_ctx* ctx, float* buf, size_t count) {
  // per-channel BatchNorm-backward sums: produced on the main stream, so the weight-gradient side stream
  // keeps running (msk_dp_allreduce_sum would join it 24 times per step and undo the overlap)
  MSK_REQUIRE(ctx, ctx->comm != nullptr, "msk_dp_init not called");
  msk_launch_scope ls(ctx, "rccl_allreduce_stats");
  if (ctx->host_transport) return ctx->world > 1 ? host_collective(ctx, buf, buf, count, 0, ctx->stream) : 0;
  CommBridge br(ctx);
  MSK_CHECK_NCCL(ctx, ncclAllReduce(buf, buf, count, ncclFloat, ncclSum, (ncclComm_t)ctx->comm, br.run));
  return 0;
}

int msk_dp_allgather(msk_ctx* ctx, const float* send, float* recv, size_t count_per_rank) {
  MSK_REQUIRE(ctx, ctx->comm != nullptr, "msk_dp_init not called");
  msk_launch_scope ls(ctx, "rccl_allgather");
  if (ctx->host_transport) {
    if (ctx->world > 1) return host_collective(ctx, send, recv, count_per_rank, 1, ctx->stream);
    MSK_CHECK_HIP(ctx, hipMemcpyAsync(recv, send, count_per_rank * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
  }
  CommBridge br(ctx);
  MSK_CHECK_NCCL(ctx, ncclAllGather(send, recv, count_per_rank, ncclFloat, (ncclComm_t)ctx->comm, br.run));
  return 0;
}

int msk_dp_broadcast(msk_ctx* ctx, float* buf, size_t count, int root) {
  MSK_REQUIRE(ctx, ctx->comm != nullptr, "msk_dp_init not called");
  msk_weights_changed_impl(ctx, buf, count * sizeof(float));  // parameters are broadcast once at wrap time
  if (ctx->host_transport) return host_broadcast(ctx, buf, count, root);
  msk_launch_scope ls(ctx, "rccl_broadcast");
  CommBridge br(ctx);
  MSK_CHECK_NCCL(ctx, ncclBroadcast(buf, buf, count, ncclFloat, root, (ncclComm_t)ctx->comm, br.run));
  return 0;
}

int msk_dp_barrier(msk_ctx* ctx) {
  MSK_REQUIRE(ctx, ctx->comm != nullptr, "msk_dp_init not called");
  float* tok = (float*)msk_workspace(ctx, 256);
  if (!tok) return -1;
  if (ctx->host_transport) {
    if (ctx->world > 1 && host_collective(ctx, tok, tok, 1, 0, ctx->stream) != 0) return -1;
    MSK_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
  }
  {
    CommBridge br(ctx);
    MSK_CHECK_NCCL(ctx, ncclAllReduce(tok, tok, 1, ncclFloat, ncclSum, (ncclComm_t)ctx->comm, br.run));
  }
  MSK_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int msk_dp_destroy(msk_ctx* ctx) {
  if (ctx && ctx->host_transport) {
    for (int fd : ctx->host_fds)
      if (fd >= 0) close(fd);
    ctx->host_fds.clear();
    ctx->host_transport = false;
    ctx->comm = nullptr;
    ctx->world = 1;
    ctx->rank = 0;
    return 0;
  }
  if (ctx && ctx->comm_stream) {
    hipStreamSynchronize(ctx->comm_stream);
    if (ctx->comm_grad) ncclCommDestroy((ncclComm_t)ctx->comm_grad);
    ctx->comm_grad = nullptr;
    hipStreamDestroy(ctx->comm_stream);
    hipEventDestroy(ctx->ev_comm_main);
    hipEventDestroy(ctx->ev_comm_side);
    hipEventDestroy(ctx->ev_comm_done);
    hipEventDestroy(ctx->ev_comm_back);
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
