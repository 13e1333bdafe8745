// Is v_mfma_f32_32x32x16_f16 symmetric under transposition?  D1 = A B (+C) and D2 = B^T A^T (+C^T) hold the same products summed
// over the same k: are D1[m][n] and D2[n][m] the same BITS?   (round 5: rowmlp_stream_k takes its products transposed)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void probe(const _Float16 *A, const _Float16 *B, const float *C, float *D1, float *D2, int signedB) {
  // one wave: A [32][16], B [16][32] (given as Bt [32][16]), C [32][32]
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const size_t t = blockIdx.x;
  A += t * 512; B += t * 512; C += t * 1024; D1 += t * 1024; D2 += t * 1024;
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = A[r * 16 + 8 * h + i]; b[i] = B[r * 16 + 8 * h + i]; }
  f16v c1, c2;
  for (int e = 0; e < 16; ++e) {
    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
    c1[e] = C[row * 32 + r];   // C[m=row][n=r]
    c2[e] = C[r * 32 + row];   // C^T[n=row][m=r]
  }
  const f16v d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);  // D1[m][n], lane n = r
  const f16v d2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c2, 0, 0, 0);  // D2[n][m], lane m = r
  for (int e = 0; e < 16; ++e) {
    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
    D1[row * 32 + r] = d1[e];  // [m][n]
    D2[r * 32 + row] = d2[e];  // stored as [m][n] too
  }
}
int main() {
  const int T = 4096;
  _Float16 *hA = (_Float16 *)malloc(T * 512 * 2), *hB = (_Float16 *)malloc(T * 512 * 2);
  float *hC = (float *)malloc(T * 1024 * 4), *h1 = (float *)malloc(T * 1024 * 4), *h2 = (float *)malloc(T * 1024 * 4);
  _Float16 *dA, *dB; float *dC, *d1, *d2;
  hipMalloc(&dA, T * 512 * 2); hipMalloc(&dB, T * 512 * 2); hipMalloc(&dC, T * 1024 * 4); hipMalloc(&d1, T * 1024 * 4); hipMalloc(&d2, T * 1024 * 4);
  for (int variant = 0; variant < 4; ++variant) {  // 0: A >= 0, C = 0; 1: A signed, C = 0; 2: A >= 0, C random; 3: A signed, C random
    srand(7 + variant);
    for (int i = 0; i < T * 512; ++i) {
      float a = (float)rand() / RAND_MAX * 2.f - 1.f, b = (float)rand() / RAND_MAX * 2.f - 1.f;
      if (!(variant & 1)) a = a < 0 ? (rand() % 3 ? 0.f : -a) : a;
      hA[i] = (_Float16)(a * 3.f); hB[i] = (_Float16)(b * 0.4f);
    }
    for (int i = 0; i < T * 1024; ++i) hC[i] = (variant & 2) ? ((float)rand() / RAND_MAX * 2.f - 1.f) * 4.f : 0.f;
    hipMemcpy(dA, hA, T * 512 * 2, hipMemcpyHostToDevice); hipMemcpy(dB, hB, T * 512 * 2, hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, T * 1024 * 4, hipMemcpyHostToDevice);
    probe<<<T, 64>>>(dA, dB, dC, d1, d2, variant);
    hipMemcpy(h1, d1, T * 1024 * 4, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, T * 1024 * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (long i = 0; i < (long)T * 1024; ++i) bad += ((uint32_t *)h1)[i] != ((uint32_t *)h2)[i];
    printf("variant %d (A %s, C %s): %ld of %ld elements differ between A B and (B^T A^T)^T\n", variant, (variant & 1) ? "signed" : ">= 0",
           (variant & 2) ? "random" : "0", bad, (long)T * 1024);
  }
  return 0;
}
