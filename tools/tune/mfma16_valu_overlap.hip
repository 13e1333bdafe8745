// Do independent vector instructions hide under v_mfma_f32_16x16x32_f16 on gfx950?  (round 1's mfma_valu_overlap.hip asked it of a
// DEPENDENT chain of f32 MFMAs; V1's MFMAs are fp16, 16 cycles each, eight accumulators in rotation.)  A wave runs 32 MFMAs (eight
// accumulators, four deep) with V independent v_fma_f32 after each; time per MFMA against V, at 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int V>
__global__ void k(float *out, int iters) {
  f32x4 acc[8];
  for (int e = 0; e < 8; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 1e-3f + i); b[i] = (_Float16)(1.0f + i * 0.01f); }
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  float v[8];
  for (int t = 0; t < 8; ++t) v[t] = x + t;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m & 7], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < V; ++t) v[(m * V + t) & 7] = __builtin_fmaf(v[(m * V + t) & 7], y, x);
    }
  }
  float s = 0.f;
  for (int e = 0; e < 8; ++e) s += acc[e][0] + acc[e][1] + acc[e][2] + acc[e][3];
  for (int t = 0; t < 8; ++t) s += v[t];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int V>
void run(float *out, int threads) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<V><<<256, threads>>>(out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  k<V><<<256, threads>>>(out, iters);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)iters * 32;
  const double ns = ms * 1e6 / mf;  // per MFMA of ONE wave
  printf("waves/SIMD=%d V=%d: %.2f ns per (MFMA + %d FMA) per wave; per SIMD %.2f ns per MFMA slot\n", threads / 256, V, ns, V, ns / (threads / 256));
}
int main() {
  float *out; (void)hipMalloc(&out, 256 * 512 * 4);
  for (int th : {256, 512}) {
    run<0>(out, th); run<1>(out, th); run<2>(out, th); run<3>(out, th); run<4>(out, th); run<6>(out, th); run<8>(out, th);
  }
  return 0;
}
