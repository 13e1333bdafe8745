// Rate of v_mfma_f32_32x32x2f32 with NACC independent accumulator chains per wave (round-robin), 1 and 2 waves per SIMD,
// operands in registers (R) or one ds_read_b128 per four MFMAs (L).  Tuning aid, not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, bool LDS>
__global__ void k(float *out, int iters) {
  __shared__ float4 tile[64 * 17];
  f32x16 acc[NACC];
  for (int c = 0; c < NACC; ++c)
    for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int t = threadIdx.x; t < 64 * 17; t += blockDim.x) tile[t] = make_float4(a, b, a, b);
  __syncthreads();
  const float4 *p = tile + (threadIdx.x & 63);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      float4 av = make_float4(a, a, a, a);
      if (LDS) av = p[64 * (m & 15)];
#pragma unroll
      for (int c = 0; c < NACC; ++c) {
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b, acc[c], 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < NACC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b, acc[c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < NACC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b, acc[c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < NACC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b, acc[c], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int c = 0; c < NACC; ++c)
    for (int e = 0; e < 16; ++e) s += acc[c][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, bool LDS>
void run(float *out, int threads) {
  const int iters = 1000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<NACC, LDS><<<256, threads>>>(out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  k<NACC, LDS><<<256, threads>>>(out, iters);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)iters * 32 * NACC * (threads / 256);  // MFMAs per SIMD
  const double tf = mf * 4096.0 * 1024 / (ms * 1e-3) / 1e12;
  printf("%s chains=%d waves/SIMD=%d: %.1f cycles per MFMA on the pipe @2.4GHz, %.1f TFLOP/s\n", LDS ? "L" : "R", NACC,
         threads / 256, ms * 1e6 * 2.4 / mf, tf);
}
int main() {
  float *out; (void)hipMalloc(&out, 256 * 512 * 4);
  for (int threads : {256, 512}) {
    run<1, false>(out, threads); run<2, false>(out, threads); run<4, false>(out, threads);
    run<1, true>(out, threads); run<2, true>(out, threads); run<4, true>(out, threads);
  }
  return 0;
}
