// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds 16-bit values equal to their own element index;
// every lane supplies the address of 4 contiguous elements at (row = lane/4 [within 16-lane group: i/4], ...).
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/tr16_probe.hip -o /tmp/tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15;
  // lane supplies: group g base 1000*g?  use element index = g*256 + (i/4)*32 + (i%4)*4  (row pitch 32 elements)
  const int e = g * 256 + (i >> 2) * 32 + (i & 3) * 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + e));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", h[l * 4 + j]); printf("\n"); }
  return 0;
}
