// Stand-alone tuning harness for the materialising pairwise kernel (K1).  Not part of the library.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/tune/k1_tune.hip -o tools/tune/k1_tune
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store(float4 v, float4 *dst) { f32x4 t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<f32x4 *>(dst)); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ float gram(float x0, float x1, float x2, float rx, float y0, float y1, float y2, float ry) {
  float zz = (x0 * y0 + x1 * y1) + x2 * y2;
  return (rx + ry) - 2.0f * zz;
}

template <int ROWS, int NT, bool NTS>
__global__ __launch_bounds__(NT) void k1(const float *__restrict__ x, const float *__restrict__ y, float *__restrict__ P, int N, int M) {
  __shared__ float xs[ROWS * 4];
  const int b = blockIdx.z, i0 = blockIdx.y * ROWS, j0 = (blockIdx.x * NT + threadIdx.x) * 4;
  if (threadIdx.x < ROWS) {
    const float *xp = x + ((size_t)b * N + i0 + threadIdx.x) * 3;
    float a = xp[0], c = xp[1], d = xp[2];
    xs[threadIdx.x * 4] = a; xs[threadIdx.x * 4 + 1] = c; xs[threadIdx.x * 4 + 2] = d; xs[threadIdx.x * 4 + 3] = (a * a + c * c) + d * d;
  }
  __syncthreads();
  if (j0 >= M) return;
  const float4 *yp = reinterpret_cast<const float4 *>(y + ((size_t)b * M + j0) * 3);
  const float4 ya = yp[0], yb = yp[1], yc = yp[2];
  const float r0 = (ya.x * ya.x + ya.y * ya.y) + ya.z * ya.z, r1 = (ya.w * ya.w + yb.x * yb.x) + yb.y * yb.y;
  const float r2 = (yb.z * yb.z + yb.w * yb.w) + yc.x * yc.x, r3 = (yc.y * yc.y + yc.z * yc.z) + yc.w * yc.w;
  float *out = P + ((size_t)b * N + i0) * M + j0;
#pragma unroll 4
  for (int r = 0; r < ROWS; ++r) {
    const float4 xv = *reinterpret_cast<const float4 *>(&xs[r * 4]);
    float4 v;
    v.x = gram(xv.x, xv.y, xv.z, xv.w, ya.x, ya.y, ya.z, r0);
    v.y = gram(xv.x, xv.y, xv.z, xv.w, ya.w, yb.x, yb.y, r1);
    v.z = gram(xv.x, xv.y, xv.z, xv.w, yb.z, yb.w, yc.x, r2);
    v.w = gram(xv.x, xv.y, xv.z, xv.w, yc.y, yc.z, yc.w, r3);
    float4 *dst = reinterpret_cast<float4 *>(out + (size_t)r * M);
    if (NTS) nt_store(v, dst); else *dst = v;
  }
}

// persistent-ish: each block walks row tiles with a stride; y registers reloaded per cloud only
template <int ROWS, bool NTS>
__global__ __launch_bounds__(256) void k1p(const float *__restrict__ x, const float *__restrict__ y, float *__restrict__ P, int N, int M, int tiles_per_block) {
  __shared__ float xs[2][ROWS * 4];
  const int b = blockIdx.z, j0 = threadIdx.x * 4;
  const float4 *yp = reinterpret_cast<const float4 *>(y + ((size_t)b * M + j0) * 3);
  const float4 ya = yp[0], yb = yp[1], yc = yp[2];
  const float r0 = (ya.x * ya.x + ya.y * ya.y) + ya.z * ya.z, r1 = (ya.w * ya.w + yb.x * yb.x) + yb.y * yb.y;
  const float r2 = (yb.z * yb.z + yb.w * yb.w) + yc.x * yc.x, r3 = (yc.y * yc.y + yc.z * yc.z) + yc.w * yc.w;
  for (int t = 0; t < tiles_per_block; ++t) {
    const int i0 = (blockIdx.y * tiles_per_block + t) * ROWS;
    float *xb = xs[t & 1];
    if (threadIdx.x < ROWS) {
      const float *xp = x + ((size_t)b * N + i0 + threadIdx.x) * 3;
      float a = xp[0], c = xp[1], d = xp[2];
      xb[threadIdx.x * 4] = a; xb[threadIdx.x * 4 + 1] = c; xb[threadIdx.x * 4 + 2] = d; xb[threadIdx.x * 4 + 3] = (a * a + c * c) + d * d;
    }
    __syncthreads();
    float *out = P + ((size_t)b * N + i0) * M + j0;
#pragma unroll 4
    for (int r = 0; r < ROWS; ++r) {
      const float4 xv = *reinterpret_cast<const float4 *>(&xb[r * 4]);
      float4 v;
      v.x = gram(xv.x, xv.y, xv.z, xv.w, ya.x, ya.y, ya.z, r0);
      v.y = gram(xv.x, xv.y, xv.z, xv.w, ya.w, yb.x, yb.y, r1);
      v.z = gram(xv.x, xv.y, xv.z, xv.w, yb.z, yb.w, yc.x, r2);
      v.w = gram(xv.x, xv.y, xv.z, xv.w, yc.y, yc.z, yc.w, r3);
      float4 *dst = reinterpret_cast<float4 *>(out + (size_t)r * M);
      if (NTS) nt_store(v, dst); else *dst = v;
    }
  }
}

// wave-owns-row: every wave writes whole 4 KiB rows (M=1024) with 4 consecutive dwordx4 stores;
// each lane keeps 16 y points in registers.  grid (1, N/ROWS, B), block NT.
template <int ROWS, int NT>
__global__ __launch_bounds__(NT) void k1w(const float *__restrict__ x, const float *__restrict__ y, float *__restrict__ P, int N, int M) {
  __shared__ float xs[ROWS * 4];
  const int b = blockIdx.z, i0 = blockIdx.y * ROWS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < ROWS) {
    const float *xp = x + ((size_t)b * N + i0 + threadIdx.x) * 3;
    float a = xp[0], c = xp[1], d = xp[2];
    xs[threadIdx.x * 4] = a; xs[threadIdx.x * 4 + 1] = c; xs[threadIdx.x * 4 + 2] = d; xs[threadIdx.x * 4 + 3] = (a * a + c * c) + d * d;
  }
  float4 ya[4], yb[4], yc[4]; float r0[4], r1[4], r2[4], r3[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float4 *yp = reinterpret_cast<const float4 *>(y + ((size_t)b * M + c * 256 + lane * 4) * 3);
    ya[c] = yp[0]; yb[c] = yp[1]; yc[c] = yp[2];
    r0[c] = (ya[c].x * ya[c].x + ya[c].y * ya[c].y) + ya[c].z * ya[c].z; r1[c] = (ya[c].w * ya[c].w + yb[c].x * yb[c].x) + yb[c].y * yb[c].y;
    r2[c] = (yb[c].z * yb[c].z + yb[c].w * yb[c].w) + yc[c].x * yc[c].x; r3[c] = (yc[c].y * yc[c].y + yc[c].z * yc[c].z) + yc[c].w * yc[c].w;
  }
  __syncthreads();
  for (int r = wave; r < ROWS; r += NT / 64) {
    const float4 xv = *reinterpret_cast<const float4 *>(&xs[r * 4]);
    float *out = P + ((size_t)b * N + i0 + r) * M + lane * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float4 v;
      v.x = gram(xv.x, xv.y, xv.z, xv.w, ya[c].x, ya[c].y, ya[c].z, r0[c]);
      v.y = gram(xv.x, xv.y, xv.z, xv.w, ya[c].w, yb[c].x, yb[c].y, r1[c]);
      v.z = gram(xv.x, xv.y, xv.z, xv.w, yb[c].z, yb[c].w, yc[c].x, r2[c]);
      v.w = gram(xv.x, xv.y, xv.z, xv.w, yc[c].y, yc[c].z, yc[c].w, r3[c]);
      *reinterpret_cast<float4 *>(out + c * 256) = v;
    }
  }
}

__global__ void fill4(float4 *p, size_t n4) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
  const int B = 32, N = 1024, M = 1024;
  float *x, *y, *P;
  CK(hipMalloc(&x, (size_t)B * N * 3 * 4)); CK(hipMalloc(&y, (size_t)B * M * 3 * 4)); CK(hipMalloc(&P, (size_t)B * N * M * 4));
  std::vector<float> h((size_t)B * N * 3);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  CK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(y, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = (4.0 * N * M + 12.0 * (N + M)) * B;
  auto bench = [&](const char *name, auto launch) {
    for (int i = 0; i < 10; ++i) launch();
    std::vector<float> ts;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0, 0);
      for (int i = 0; i < 100; ++i) launch();
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms * 10.f);
    }
    std::sort(ts.begin(), ts.end());
    printf("%-34s median %.2f us  %.0f GB/s   (min %.2f us %.0f GB/s)\n", name, ts[2], bytes / ts[2] / 1e3, ts[0], bytes / ts[0] / 1e3);
  };
#define V(ROWS, NT, NTS) bench("k1<" #ROWS "," #NT "," #NTS ">", [&] { k1<ROWS, NT, NTS><<<dim3((M + NT * 4 - 1) / (NT * 4), N / ROWS, B), NT>>>(x, y, P, N, M); });
  V(16, 256, false) V(16, 256, true) V(8, 256, false) V(8, 256, true) V(32, 256, false) V(32, 256, true) V(64, 256, true) V(4, 256, true)
  V(16, 128, true) V(32, 128, true) V(16, 64, true)
#define VW(ROWS, NT) bench("k1w<" #ROWS "," #NT ">", [&] { k1w<ROWS, NT><<<dim3(1, N / ROWS, B), NT>>>(x, y, P, N, M); });
  VW(16, 256) VW(32, 256) VW(64, 256) VW(128, 256) VW(32, 512) VW(64, 512) VW(64, 1024) VW(128, 1024) VW(16, 64) VW(32, 64) VW(32, 128) VW(64, 128)
  V(32, 256, false) V(64, 256, false) V(128, 256, false)
#define VP(ROWS, NTS, TPB) bench("k1p<" #ROWS "," #NTS "> tpb=" #TPB, [&] { k1p<ROWS, NTS><<<dim3(1, N / ROWS / TPB, B), 256>>>(x, y, P, N, M, TPB); });
  VP(16, false, 4) VP(16, true, 4) VP(16, true, 8) VP(16, true, 16) VP(8, true, 8) VP(16, true, 64)
  bench("fill float4 (pure store)", [&] { fill4<<<2048, 256>>>((float4 *)P, (size_t)B * N * M / 4); });
  bench("fill float4 8192 blocks", [&] { fill4<<<8192, 256>>>((float4 *)P, (size_t)B * N * M / 4); });
  bench("hipMemsetAsync", [&] { hipMemsetAsync(P, 0, (size_t)B * N * M * 4, 0); });
  return 0;
}
