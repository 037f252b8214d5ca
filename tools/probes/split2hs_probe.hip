// Probe of the scaled-low-piece operand format (msk_wbf.h): reconstructs x*w from the pieces the kernels use and prints
// the worst relative error.   hipcc --offload-arch=gfx950 -I medicalseg_amd/csrc -I include tools/probes/split2hs_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "msk_wbf.h"
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void probe(const float* x, const float* w, float* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned hi, lo, whi, wlo;
  wbf_split2hs_pair(x[i], x[i], hi, lo);
  wbf_split2h_pair(w[i], w[i], whi, wlo);
  uint4 hb = {whi, whi, whi, whi};
  uint4 bd = wbf_hi_down(hb);
  h2 H = __builtin_bit_cast(h2, hi), L = __builtin_bit_cast(h2, lo), WH = __builtin_bit_cast(h2, whi), WL = __builtin_bit_cast(h2, wlo),
     BD = __builtin_bit_cast(h2, bd.x);
  float p = (float)L.x * (float)BD.x + (float)H.x * (float)WL.x + (float)H.x * (float)WH.x;
  out[i] = p;
  out[n + i] = (float)L.x;
  out[2 * n + i] = (float)BD.x;
}
int main() {
  const int n = 4096;
  float *x, *w, *o;
  hipMallocManaged(&x, n * 4); hipMallocManaged(&w, n * 4); hipMallocManaged(&o, 3 * n * 4);
  for (int i = 0; i < n; ++i) { x[i] = (float)((i * 7919 % 20011) - 10000) * 0.37f * powf(2.f, -(i % 20)); w[i] = (float)((i * 104729 % 3001) - 1500) * 0.41f; }
  probe<<<n / 256, 256>>>(x, w, o, n);
  hipDeviceSynchronize();
  double worst = 0; int wi = 0;
  for (int i = 0; i < n; ++i) { double t = (double)x[i] * w[i]; if (t == 0) continue; double e = fabs(o[i] - t) / fabs(t); if (e > worst) { worst = e; wi = i; } }
  printf("worst rel err %.3e at x=%g w=%g got %g  (l'=%g, wdown=%g)\n", worst, x[wi], w[wi], o[wi], o[n + wi], o[2 * n + wi]);
  return 0;
}
