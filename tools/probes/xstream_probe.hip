// Cost of a cross-stream hand-over (event record on A -> B waits -> kernel on B -> event record on B -> A waits) against the
// same kernels on one stream.   hipcc --offload-arch=gfx950 tools/probes/xstream_probe.hip -o /tmp/x && /tmp/x
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void work(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
int main() {
  float* p; hipMalloc(&p, 64 << 20);
  hipStream_t A, B; hipStreamCreateWithFlags(&A, hipStreamNonBlocking); hipStreamCreateWithFlags(&B, hipStreamNonBlocking);
  hipEvent_t ea, eb, t0, t1; hipEventCreateWithFlags(&ea, hipEventDisableTiming); hipEventCreateWithFlags(&eb, hipEventDisableTiming);
  hipEventCreate(&t0); hipEventCreate(&t1);
  const int N = 200;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipDeviceSynchronize();
      hipEventRecord(t0, A);
      for (int i = 0; i < N; ++i) {
        work<<<4096, 256, 0, A>>>(p, 1 << 20);                       // ~ a few us of main-stream work
        if (mode == 0) { tiny<<<1, 64, 0, A>>>(p); }
        else {
          hipEventRecord(ea, A); hipStreamWaitEvent(B, ea, 0);
          tiny<<<1, 64, 0, B>>>(p);
          hipEventRecord(eb, B); hipStreamWaitEvent(A, eb, 0);
        }
        if (mode == 2) { /* second hand-over per iteration */
          hipEventRecord(ea, A); hipStreamWaitEvent(B, ea, 0);
          tiny<<<1, 64, 0, B>>>(p);
          hipEventRecord(eb, B); hipStreamWaitEvent(A, eb, 0);
        }
        work<<<4096, 256, 0, A>>>(p, 1 << 20);
      }
      hipEventRecord(t1, A); hipEventSynchronize(t1);
      float ms; hipEventElapsedTime(&ms, t0, t1);
      if (rep) printf("mode %d: %.2f us per iteration\n", mode, ms * 1e3 / N);
    }
  }
  return 0;
}
