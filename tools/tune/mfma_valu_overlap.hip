// Does VALU work hide under f32 MFMAs on gfx950?  A wave runs a chain of dependent v_mfma_f32_32x32x2f32 with V
// independent v_fma_f32 between consecutive MFMAs; time per MFMA vs V, at 1 and 2 waves per SIMD.  (Tuning aid, not
// part of the library.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int V>
__global__ void k(float *out, int iters) {
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  float v[8];
  for (int t = 0; t < 8; ++t) v[t] = a + t;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < V; ++t) v[t & 7] = __builtin_fmaf(v[t & 7], b, a);
    }
  }
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += acc[e];
  for (int t = 0; t < 8; ++t) s += v[t];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int V>
void run(float *out, int threads) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<V><<<256, threads>>>(out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  k<V><<<256, threads>>>(out, iters);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)iters * 8;
  printf("waves/SIMD=%d V=%2d: %.1f ns per MFMA per wave (%.0f cycles @2.4GHz); MFMA pipe %.0f%% busy if 64 cycles each\n",
         threads / 256, V, ms * 1e6 / mf, ms * 1e6 / mf * 2.4, 100.0 * 64 * (threads / 256) / (ms * 1e6 / mf * 2.4));
}
int main() {
  float *out; (void)hipMalloc(&out, 256 * 512 * 4);
  for (int threads : {256, 512}) {
    run<0>(out, threads); run<8>(out, threads); run<16>(out, threads); run<32>(out, threads); run<64>(out, threads);
  }
  return 0;
}
