// Tuning harness for the fused NN-min kernel (K2).  Not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
  float dx = ax - bx, dy = ay - by, dz = az - bz;
  return (dx * dx + dy * dy) + dz * dz;
}
constexpr int RCH = 1024;
template <int Q, int NW, bool SCALAR>
__global__ __launch_bounds__(NW * 64) void nn_min3(const float *__restrict__ x, const float *__restrict__ y, int N, int M,
                                                   float *__restrict__ min_x, int *__restrict__ arg_x,
                                                   float *__restrict__ min_y, int *__restrict__ arg_y) {
  constexpr int QB = 64 * Q;
  __shared__ float4 sref[RCH];
  __shared__ float sval[NW][QB];
  __shared__ int sidx[NW][QB];
  const int b = blockIdx.z, dir = blockIdx.y;
  const float *qp = dir == 0 ? x : y, *rp = dir == 0 ? y : x;
  const int nq = dir == 0 ? N : M, nr = dir == 0 ? M : N;
  float *omin = dir == 0 ? min_x : min_y;
  int *oarg = dir == 0 ? arg_x : arg_y;
  const int q0 = blockIdx.x * QB;
  if (q0 >= nq) return;
  qp += (size_t)b * nq * 3; rp += (size_t)b * nr * 3;
  const int lane = threadIdx.x & 63;
  const int wave = SCALAR ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (threadIdx.x >> 6);
  float qx[Q], qy[Q], qz[Q], best[Q]; int bi[Q];
#pragma unroll
  for (int t = 0; t < Q; ++t) {
    int q = q0 + lane + 64 * t; q = q < nq ? q : nq - 1;
    qx[t] = qp[q * 3]; qy[t] = qp[q * 3 + 1]; qz[t] = qp[q * 3 + 2]; best[t] = __builtin_inff(); bi[t] = 0;
  }
  for (int c0 = 0; c0 < nr; c0 += RCH) {
    const int cnt = min(RCH, nr - c0);
    __syncthreads();
    for (int p = threadIdx.x; p < cnt; p += NW * 64) { const float *s = rp + (size_t)(c0 + p) * 3; sref[p] = make_float4(s[0], s[1], s[2], 0.f); }
    __syncthreads();
    const int per = RCH / NW;
    const int lo = wave * per, hi = min(lo + per, cnt);
#pragma unroll 4
    for (int p = lo; p < hi; ++p) {
      const float4 r = sref[p];
#pragma unroll
      for (int t = 0; t < Q; ++t) {
        const float d = sqdist3(qx[t], qy[t], qz[t], r.x, r.y, r.z);
        const bool lt = d < best[t];
        best[t] = lt ? d : best[t];
        bi[t] = lt ? c0 + p : bi[t];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < Q; ++t) { sval[wave][lane + 64 * t] = best[t]; sidx[wave][lane + 64 * t] = bi[t]; }
  __syncthreads();
  for (int e = threadIdx.x; e < QB; e += NW * 64) {
    const int q = q0 + e;
    if (q < nq) {
      float v = sval[0][e]; int ix = sidx[0][e];
#pragma unroll
      for (int w = 1; w < NW; ++w) { const float ov = sval[w][e]; const int oi = sidx[w][e]; const bool take = (ov < v) || (ov == v && oi < ix); v = take ? ov : v; ix = take ? oi : ix; }
      omin[(size_t)b * nq + q] = v; oarg[(size_t)b * nq + q] = ix;
    }
  }
}
int main() {
  const int B = 32, N = 1024, M = 1024;
  float *x, *y, *mx, *my; int *ax, *ay;
  hipMalloc(&x, B * N * 12); hipMalloc(&y, B * M * 12); hipMalloc(&mx, B * N * 4); hipMalloc(&my, B * M * 4); hipMalloc(&ax, B * N * 4); hipMalloc(&ay, B * M * 4);
  std::vector<float> h((size_t)B * N * 3);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 40503u + 7) % 1999) / 1000.f - 1.f;
  hipMemcpy(y, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<int> ref(B * N), got(B * N);
  auto bench = [&](const char *name, auto launch, bool first) {
    for (int i = 0; i < 5; ++i) launch();
    std::vector<float> ts;
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipEventRecord(e0, 0); for (int i = 0; i < 50; ++i) launch(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms * 20.f);
    }
    std::sort(ts.begin(), ts.end());
    hipMemcpy(got.data(), ax, B * N * 4, hipMemcpyDeviceToHost);
    if (first) ref = got;
    printf("%-22s median %.2f us  %s\n", name, ts[2], got == ref ? "same-argmins" : "MISMATCH");
  };
#define V(Q, NW, SC, first) bench("Q=" #Q " NW=" #NW " scalar=" #SC, [&] { nn_min3<Q, NW, SC><<<dim3((N + 64 * Q - 1) / (64 * Q), 2, B), NW * 64>>>(x, y, N, M, mx, ax, my, ay); }, first);
  V(2, 4, false, true) V(2, 4, true, false) V(4, 4, true, false) V(4, 8, true, false) V(2, 8, true, false) V(8, 8, true, false) V(1, 4, true, false) V(4, 16, true, false) V(2, 16, true, false) V(8, 16, true, false)
  return 0;
}
