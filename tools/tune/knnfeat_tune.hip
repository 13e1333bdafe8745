// Timing harness for hitadv_knn_features (not part of the library): build with -DKF_BUF=.. / -DKF_MODE=.. to see where
// the kernel's time goes (tools/tune/README in DESIGN.md section 4, K5).
#include "../../hit_adv_amd/csrc/knn.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char **argv) {
  const int B = 32, N = 1024, D = argc > 1 ? atoi(argv[1]) : 64, K = argc > 2 ? atoi(argv[2]) : 5;
  std::vector<float> hx((size_t)B * N * D), hxx((size_t)B * N);
  srand(1);
  for (auto &v : hx) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (size_t i = 0; i < hxx.size(); ++i) { float a = 0; for (int d = 0; d < D; ++d) a += hx[i * D + d] * hx[i * D + d]; hxx[i] = a; }
  float *x, *xx; int64_t *idx;
  hipMalloc(&x, hx.size() * 4); hipMalloc(&xx, hxx.size() * 4); hipMalloc(&idx, (size_t)B * N * K * 8);
  hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(xx, hxx.data(), hxx.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hitadv_knn_features(x, xx, B, N, D, K, idx, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  const int reps = 50;
  for (int i = 0; i < reps; ++i) hitadv_knn_features(x, xx, B, N, D, K, idx, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("D=%d K=%d SUB=%d MODE=%d: %.2f us\n", D, K, KF_SUB, KF_MODE, ms * 1000.f / reps);
  return 0;
}
