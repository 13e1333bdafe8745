// Shader clock during a small serial kernel: s_memtime (shader cycles) against s_memrealtime (100 MHz) around a dependent VALU chain,
// at 32 workgroups (what an FPS launch occupies) and at 2048.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long *out, int iters) {
  float x = threadIdx.x * 1e-3f;
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 64; ++t) x = __builtin_fmaf(x, 1.0001f, 0.5f);
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
  if (x == 12345.f) out[2] = 1;
}
int main() {
  unsigned long long *d, h[2];
  (void)hipMalloc(&d, 64);
  for (int blocks : {32, 32, 2048, 32}) {
    for (int iters : {1000, 20000}) {
      k<<<blocks, 256>>>(d, iters);
      (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      printf("blocks %4d iters %5d: %llu cycles, %.1f us -> %.2f GHz, %.2f cycles per dependent fma\n", blocks, iters, h[0], h[1] / 100.0,
             h[0] / (h[1] * 10.0), (double)h[0] / (64.0 * iters));
    }
  }
  return 0;
}
