// Rate of the bf16 MFMAs (16x16x32 and 32x32x16) on the whole chip: NACC independent accumulators per wave in rotation,
// 1 or 2 waves per SIMD, operands in registers (NB distinct B operands, random bf16 data), with the in-kernel clock
// (s_memtime / s_memrealtime).  Tuning aid behind DESIGN.md's V1 bf16x3 numbers; not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ unsigned long long g_stamp[2 * 256];
template <int NACC, bool BIG>
__global__ void k(const uint4 *in, float *out, int iters) {
  uint4 ua[6], ub[6];
  for (int i = 0; i < 6; ++i) { ua[i] = in[(threadIdx.x + 64 * i) & 1023]; ub[i] = in[(threadIdx.x + 64 * i + 512) & 1023]; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  if (BIG) {
    f32x16 acc[NACC];
    for (int c = 0; c < NACC; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int m = 0; m < 6; ++m)
#pragma unroll
        for (int c = 0; c < NACC; ++c)
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua[(m + c) % 6]), __builtin_bit_cast(bf16x8, ub[m]), acc[c], 0, 0, 0);
    }
    for (int c = 0; c < NACC; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
  } else {
    f32x4 acc[NACC];
    for (int c = 0; c < NACC; ++c) for (int e = 0; e < 4; ++e) acc[c][e] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int m = 0; m < 6; ++m)
#pragma unroll
        for (int c = 0; c < NACC; ++c)
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ua[(m + c) % 6]), __builtin_bit_cast(bf16x8, ub[m]), acc[c], 0, 0, 0);
    }
    for (int c = 0; c < NACC; ++c) for (int e = 0; e < 4; ++e) s += acc[c][e];
  }
  if (threadIdx.x == 0) { g_stamp[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0; g_stamp[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, bool BIG>
void run(const uint4 *in, float *out, int threads, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<NACC, BIG><<<256, threads>>>(in, out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  k<NACC, BIG><<<256, threads>>>(in, out, iters);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> st(512);
  (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamp), 512 * 8);
  std::vector<double> ghz, cyc;
  for (int i = 0; i < 256; ++i) { ghz.push_back((double)st[2 * i] / st[2 * i + 1] * 0.1); cyc.push_back((double)st[2 * i]); }
  std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
  const double mf = (double)iters * 6 * NACC * (threads / 256);  // MFMAs per SIMD
  const double flop = mf * (BIG ? 32768.0 : 16384.0) * 1024;
  printf("%s accs=%d waves/SIMD=%d iters=%d: %.2f us, %.1f s_memtime cycles per MFMA per SIMD, clock %.2f GHz, %.0f TFLOP/s\n",
         BIG ? "32x32x16" : "16x16x32", NACC, threads / 256, iters, ms * 1e3, cyc[128] / mf, ghz[128], flop / (ms * 1e-3) / 1e12);
}
int main() {
  std::vector<unsigned> h(4096);
  unsigned s = 1;
  for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (s & 0x807fffffu) | 0x3f000000u; v = (v & 0xffff0000u) | ((v >> 16) ^ 0x0123); }
  uint4 *in; float *out;
  (void)hipMalloc(&in, 4096 * 4); (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  for (int iters : {100, 2000}) {
    for (int threads : {256, 512}) {
      run<2, false>(in, out, threads, iters); run<4, false>(in, out, threads, iters); run<8, false>(in, out, threads, iters);
      run<1, true>(in, out, threads, iters); run<2, true>(in, out, threads, iters); run<4, true>(in, out, threads, iters);
    }
  }
  return 0;
}
