// Backward of "shared linear layer -> max over points" (PointNet's 128->1024 layer + global max-pool).
//
// With y[b,n,j] = x[b,n,:].W[j,:] + bias[j] and g[b,j] = max_n y[b,n,j], the gradient w.r.t. x is
//     dX[b,n,:] = sum_{j : argmax[b,j] == n} dg[b,j] * W[j,:]
// i.e. only B*Cout rows of W are ever added, to at most Cout distinct points per cloud.  torch's
// autograd materialises the [B*N,Cout] gradient (zero fill + scatter), and runs a dense GEMM over it;
// this kernel does the sparse sum directly: one WAVE per destination point, the arg-max table of the
// cloud in LDS, matches found with ballot and consumed in ascending j (fixed order, no atomics).
#include "common.hpp"
#include "hitadv.h"

namespace hitadv {

constexpr int LM_MAXC = 8;    // Cin <= 64 * LM_MAXC
constexpr int LM_RPW = 8;     // destination rows per wave
constexpr int LM_ROWS = 4 * LM_RPW;

__global__ __launch_bounds__(256) void linear_max_bwd_k(const float *__restrict__ dg, const float *__restrict__ W,
                                                        const int64_t *__restrict__ idx, int N, int Cout, int Cin,
                                                        float *__restrict__ dX) {
  extern __shared__ int sidx[];  // Cout entries
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = threadIdx.x; j < Cout; j += 256) sidx[j] = (int)idx[(size_t)b * Cout + j];
  __syncthreads();
  const float *dgb = dg + (size_t)b * Cout;
  const int nc = (Cin + 63) >> 6;
  for (int r = 0; r < LM_RPW; ++r) {
    const int n = blockIdx.x * LM_ROWS + wave * LM_RPW + r;
    if (n >= N) break;  // wave-uniform
    float acc[LM_MAXC];
#pragma unroll
    for (int c = 0; c < LM_MAXC; ++c) acc[c] = 0.f;
    for (int j0 = 0; j0 < Cout; j0 += 64) {
      const int j = j0 + lane;
      unsigned long long m = __ballot(j < Cout && sidx[j] == n);
      while (m) {
        const int jj = j0 + __builtin_ctzll(m);
        m &= m - 1;
        const float g = dgb[jj];
        const float *wr = W + (size_t)jj * Cin;
#pragma unroll
        for (int c = 0; c < LM_MAXC; ++c)
          if (c < nc) {
            const int k = lane + 64 * c;
            if (k < Cin) acc[c] = fmaf(g, wr[k], acc[c]);
          }
      }
    }
    float *o = dX + ((size_t)b * N + n) * Cin;
#pragma unroll
    for (int c = 0; c < LM_MAXC; ++c)
      if (c < nc) {
        const int k = lane + 64 * c;
        if (k < Cin) o[k] = acc[c];
      }
  }
}

}  // namespace hitadv

using namespace hitadv;

extern "C" int hitadv_linear_max_bwd(const float *dg, const float *W, const int64_t *idx, int B, int N, int Cout,
                                     int Cin, float *dX, void *stream) {
  if (!dg || !W || !idx || !dX || B <= 0 || N <= 0 || Cout <= 0 || Cin <= 0 || Cin > 64 * LM_MAXC || Cout > 16384)
    return HITADV_E_ARG;
  dim3 grid((N + LM_ROWS - 1) / LM_ROWS, B);
  linear_max_bwd_k<<<grid, 256, (size_t)Cout * sizeof(int), (hipStream_t)stream>>>(dg, W, idx, N, Cout, Cin, dX);
  HITADV_LAUNCH_CHECK();
  return 0;
}
