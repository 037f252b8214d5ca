// A 1-rank RCCL all-gather / all-reduce (a) on the compute stream, (b) on a second stream with an event hand-over each way.
//   hipcc --offload-arch=gfx950 tools/probes/xstream_rccl_probe.hip -o /tmp/xr -L/opt/rocm/lib -lrccl && /tmp/xr
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
__global__ void work(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
int main() {
  float *p, *s, *r; hipMalloc(&p, 64 << 20); hipMalloc(&s, 4096); hipMalloc(&r, 4096);
  hipStream_t A, B; hipStreamCreateWithFlags(&A, hipStreamNonBlocking); hipStreamCreateWithFlags(&B, hipStreamNonBlocking);
  hipEvent_t ea, eb, t0, t1; hipEventCreateWithFlags(&ea, hipEventDisableTiming); hipEventCreateWithFlags(&eb, hipEventDisableTiming);
  hipEventCreate(&t0); hipEventCreate(&t1);
  ncclUniqueId id; ncclGetUniqueId(&id); ncclComm_t comm; ncclCommInitRank(&comm, 1, id, 0);
  const int N = 200;
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipDeviceSynchronize();
      hipEventRecord(t0, A);
      for (int i = 0; i < N; ++i) {
        work<<<4096, 256, 0, A>>>(p, 1 << 20);
        if (mode == 0) ncclAllGather(s, r, 512, ncclFloat, comm, A);
        else if (mode == 1) ncclAllReduce(s, s, 512, ncclFloat, ncclSum, comm, A);
        else {
          hipEventRecord(ea, A); hipStreamWaitEvent(B, ea, 0);
          if (mode == 2) ncclAllGather(s, r, 512, ncclFloat, comm, B); else ncclAllReduce(s, s, 512, ncclFloat, ncclSum, comm, B);
          hipEventRecord(eb, B); hipStreamWaitEvent(A, eb, 0);
        }
        work<<<4096, 256, 0, A>>>(p, 1 << 20);
      }
      hipEventRecord(t1, A); hipEventSynchronize(t1);
      float ms; hipEventElapsedTime(&ms, t0, t1);
      if (rep) printf("mode %d (%s): %.2f us per iteration\n", mode, mode == 0 ? "allgather on A" : mode == 1 ? "allreduce on A" : mode == 2 ? "allgather on B + hand-over" : "allreduce on B + hand-over", ms * 1e3 / N);
    }
  }
  return 0;
}
