// Pure-store microbenchmarks: what store shape reaches the memset rate on MI355X?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// each wave writes CH consecutive KiB per step (CH dwordx4 instructions to consecutive 1 KiB chunks)
template <int CH, int NT>
__global__ __launch_bounds__(NT) void fill_wavechunk(f32x4 *p, size_t n4) {
  const size_t wave = ((size_t)blockIdx.x * NT + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const size_t nw = ((size_t)gridDim.x * NT) >> 6;
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t base = wave * 64 * CH; base < n4; base += nw * 64 * CH) {
#pragma unroll
    for (int c = 0; c < CH; ++c) p[base + c * 64 + lane] = v;
  }
}
// block-contiguous: each block owns a contiguous span, threads stride by NT
template <int UN, int NT>
__global__ __launch_bounds__(NT) void fill_blockspan(f32x4 *p, size_t n4) {
  const size_t per = n4 / gridDim.x;
  f32x4 *q = p + per * blockIdx.x;
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = threadIdx.x; i < per; i += (size_t)NT * UN) {
#pragma unroll
    for (int u = 0; u < UN; ++u) q[i + (size_t)u * NT] = v;
  }
}
template <int NT>
__global__ __launch_bounds__(NT) void fill_grid(f32x4 *p, size_t n4) {
  size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * NT;
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (; i < n4; i += stride) p[i] = v;
}
int main() {
  const size_t bytes = (size_t)32 * 1024 * 1024 * 4, n4 = bytes / 16;
  f32x4 *P; hipMalloc(&P, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto bench = [&](const char *name, auto launch) {
    for (int i = 0; i < 10; ++i) launch();
    std::vector<float> ts;
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipEventRecord(e0, 0);
      for (int i = 0; i < 100; ++i) launch();
      (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms * 10.f);
    }
    std::sort(ts.begin(), ts.end());
    printf("%-40s median %.2f us  %.0f GB/s\n", name, ts[2], bytes / ts[2] / 1e3);
  };
  bench("memset", [&] { (void)hipMemsetAsync(P, 0, bytes, 0); });
  bench("memsetD32", [&] { (void)hipMemsetD32Async((hipDeviceptr_t)P, 0x3f800000, bytes / 4, 0); });
#define G(NT, BL) bench("fill_grid<" #NT "> blocks=" #BL, [&] { fill_grid<NT><<<BL, NT>>>(P, n4); });
  G(256, 1024) G(256, 2048) G(256, 4096) G(256, 8192) G(256, 32768) G(512, 2048) G(1024, 1024) G(1024, 2048) G(1024, 8192) G(64, 32768) G(256,131072)
#define W(CH, NT, BL) bench("fill_wavechunk<" #CH "," #NT "> blocks=" #BL, [&] { fill_wavechunk<CH, NT><<<BL, NT>>>(P, n4); });
  W(4, 256, 2048) W(4, 256, 8192) W(8, 256, 2048) W(8, 256, 4096) W(16, 256, 2048) W(4, 64, 8192) W(4,64,32768) W(16, 64, 8192) W(2, 256, 8192) W(4,1024,2048)
#define S(UN, NT, BL) bench("fill_blockspan<" #UN "," #NT "> blocks=" #BL, [&] { fill_blockspan<UN, NT><<<BL, NT>>>(P, n4); });
  S(4, 256, 2048) S(4, 256, 8192) S(8, 256, 1024) S(4, 1024, 1024) S(4, 1024, 2048) S(8, 512, 2048) S(1, 256, 8192)
  return 0;
}
