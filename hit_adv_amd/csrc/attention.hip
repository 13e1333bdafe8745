// PCT's offset attention between its two batched products (model/pct_cls.py:127-131):
//
//     attention = softmax(energy, dim=-1)                                   rows of energy [B,N,N]
//     attention = attention / (1e-9 + attention.sum(dim=1, keepdim=True))   column renormalisation
//
// torch runs this as softmax + sum + add + div (four passes over the 8 MB tensor at B = 32, N = 256) and its backward as
// nine more (div / neg / mul / sum / expand-add ... + softmax backward): per PCT pass 4 layers x 13 element-wise launches of
// 5-14 us, a sixth of the pass (tools/pct_pass_ops.py: 612 us of at::native kernels in the 1374 us of the four layers).
// Here: two launches forward, two backward, every reduction in a fixed order (no atomics: same bits every run).
//
//   forward   S = softmax rows;  c_j = sum_i S[i,j] (ascending i);  A = S / (1e-9 + c_j);  A [B,N,N] and c [B,N] are kept
//   backward  given dA:  h_j = sum_i dA[i,j] A[i,j] (ascending i);  dS[i,j] = (dA[i,j] - h_j) / (1e-9 + c_j);
//             dE[i,j] = S[i,j] (dS[i,j] - sum_j' dS[i,j'] S[i,j']),  S = A (1e-9 + c_j)
//
// N a multiple of 64, N <= 1024 (PCT: 256).  One wave per row in the row kernels (lane = column mod 64), one workgroup per
// (cloud, 64 columns) in the column kernels: its eight waves take the rows i = w mod 8 and meet in LDS in wave order.
#include "common.hpp"
#include "hitadv.h"

namespace hitadv {

constexpr int OA_MAXQ = 16;       // N / 64 <= 16
constexpr float OA_EPS = 1e-9f;   // model/pct_cls.py:129

// wave-wide sum / max in a fixed butterfly order (every lane ends with the same value)
__device__ __forceinline__ float oa_wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, HITADV_WAVE);
  return v;
}
__device__ __forceinline__ float oa_wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, HITADV_WAVE));
  return v;
}

// S = softmax of every row of E [rows, N]; one wave per row
__global__ __launch_bounds__(256) void oa_softmax_rows_k(const float *__restrict__ E, float *__restrict__ S, long long rows, int N) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int Q = N >> 6;
  const float *e = E + row * N;
  float v[OA_MAXQ];
  float mx = -__builtin_inff();
#pragma unroll
  for (int q = 0; q < OA_MAXQ; ++q) {
    v[q] = q < Q ? e[64 * q + lane] : -__builtin_inff();
    mx = fmaxf(mx, v[q]);
  }
  mx = oa_wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int q = 0; q < OA_MAXQ; ++q) {
    v[q] = q < Q ? expf(v[q] - mx) : 0.f;
    sum += v[q];
  }
  sum = oa_wave_sum(sum);
  const float inv = 1.0f / sum;
  float *s = S + row * N;
#pragma unroll
  for (int q = 0; q < OA_MAXQ; ++q)
    if (q < Q) s[64 * q + lane] = v[q] * inv;
}

// MODE 0 (forward): c_j = sum_i S[i,j], then S[i,j] /= (eps + c_j) in place (S becomes A), c written.
// MODE 1 (backward): h_j = sum_i dA[i,j] A[i,j], h written (X = dA, Y = A).
template <int MODE>
__global__ __launch_bounds__(512) void oa_columns_k(float *__restrict__ X, const float *__restrict__ Y, float *__restrict__ out,
                                                    int N) {
  constexpr int NW = 8;  // waves per block: wave w takes the rows i = w mod 8
  __shared__ float part[NW][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, j = blockIdx.x * 64 + lane;
  float *x = X + (size_t)b * N * N + j;
  const float *y = MODE == 1 ? Y + (size_t)b * N * N + j : nullptr;
  float acc = 0.f;
  for (int i0 = wave; i0 < N; i0 += 8 * NW) {  // eight rows of this wave in flight, added in ascending i
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(i0 + NW * u, N - 1);
      t[u] = MODE == 1 ? x[(size_t)i * N] * y[(size_t)i * N] : x[(size_t)i * N];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += i0 + NW * u < N ? t[u] : 0.f;
  }
  part[wave][lane] = acc;
  __syncthreads();
  float total = part[0][lane];
#pragma unroll
  for (int w = 1; w < NW; ++w) total += part[w][lane];  // wave order
  if (wave == 0) out[(size_t)b * N + j] = total;
  if (MODE == 0) {
    const float inv = 1.0f / (OA_EPS + total);
    for (int i0 = wave; i0 < N; i0 += 8 * NW) {  // the same rows again (L2), eight in flight
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = x[(size_t)min(i0 + NW * u, N - 1) * N];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + NW * u < N) x[(size_t)(i0 + NW * u) * N] = t[u] * inv;
    }
  }
}

// dE rows from dA, A, c, h (see the header); one wave per row
__global__ __launch_bounds__(256) void oa_backward_rows_k(const float *__restrict__ dA, const float *__restrict__ A,
                                                          const float *__restrict__ c, const float *__restrict__ h,
                                                          float *__restrict__ dE, long long rows, int N) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int Q = N >> 6;
  const long long b = row / N;
  const float *da = dA + row * N, *a = A + row * N, *cc = c + b * N, *hh = h + b * N;
  float s[OA_MAXQ], ds[OA_MAXQ];
  float dot = 0.f;
#pragma unroll
  for (int q = 0; q < OA_MAXQ; ++q) {
    s[q] = ds[q] = 0.f;
    if (q < Q) {
      const int j = 64 * q + lane;
      const float den = OA_EPS + cc[j];
      s[q] = a[j] * den;
      ds[q] = (da[j] - hh[j]) / den;
      dot += ds[q] * s[q];
    }
  }
  dot = oa_wave_sum(dot);
  float *o = dE + row * N;
#pragma unroll
  for (int q = 0; q < OA_MAXQ; ++q)
    if (q < Q) o[64 * q + lane] = s[q] * (ds[q] - dot);
}

}  // namespace hitadv

using namespace hitadv;

extern "C" int hitadv_offset_attention_supported(int N) { return N >= 64 && N <= 64 * OA_MAXQ && (N & 63) == 0; }

extern "C" int hitadv_offset_attention_fwd(const float *E, int B, int N, float *A, float *colsum, void *stream) {
  if (!E || !A || !colsum || B <= 0 || !hitadv_offset_attention_supported(N)) return HITADV_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long long rows = (long long)B * N;
  oa_softmax_rows_k<<<(unsigned)((rows + 3) / 4), 256, 0, s>>>(E, A, rows, N);
  oa_columns_k<0><<<dim3(N / 64, B), 512, 0, s>>>(A, nullptr, colsum, N);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_offset_attention_bwd(const float *dA, const float *A, const float *colsum, int B, int N, float *h,
                                           float *dE, void *stream) {
  if (!dA || !A || !colsum || !h || !dE || B <= 0 || !hitadv_offset_attention_supported(N)) return HITADV_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long long rows = (long long)B * N;
  oa_columns_k<1><<<dim3(N / 64, B), 512, 0, s>>>(const_cast<float *>(dA), A, h, N);
  oa_backward_rows_k<<<(unsigned)((rows + 3) / 4), 256, 0, s>>>(dA, A, colsum, h, dE, rows, N);
  HITADV_LAUNCH_CHECK();
  return 0;
}
