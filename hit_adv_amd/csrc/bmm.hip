// Small batched fp32 matrix products for PCT's offset-attention blocks (model/pct_cls.py:111-139: energy = q k, x_r = v
// attention, and the four products of their backward pass per layer): 32 batches of 256 x 256 outputs over K = 64 or 256.
// rocBLAS runs each batch as ONE 256 x 256 tile -- 32 workgroups on a 256-CU chip, 41-65 us per call for 0.3-1 GFLOP, fourteen
// calls per PCT pass = 17 % of cfg5's kernel time (profiles/r03_cfg5_kernel_stats.csv).  Here: 64 x 64 tiles (512 workgroups for
// the same call), exact fp32 FMA chains on the f32 matrix cores (v_mfma_f32_32x32x2_f32: k ascending, two values per
// instruction), operands transposed on their way into LDS so that all four transpose combinations read global memory along its
// contiguous dimension.
//
//   C[b] (M x N, row-major, ldc = N) = op(A[b]) op(B[b]);  op(A) is M x K: A stored [M,K] (TA = 0) or [K,M] (TA = 1);
//   op(B) is K x N: B stored [K,N] (TB = 0) or [N,K] (TB = 1).  M, N multiples of 64, K a multiple of 32.
#include <stdlib.h>

#include "common.hpp"
#include "hitadv.h"

namespace hitadv {

typedef float f32x16_b __attribute__((ext_vector_type(16)));
constexpr int BMM_T = 64, BMM_KC = 32, BMM_LD = BMM_T + 4;  // tile edge, K chunk, LDS row length (floats)

// one operand tile, k-major in LDS: s[k][i], i = the tile's 64 rows (A) or columns (B).
// KMAJOR_SRC: the source is stored [k][i] (i contiguous): float4 along i.  Otherwise [i][k] (k contiguous): float4 along k,
// transposed by four scalar LDS writes.
template <bool KMAJOR_SRC>
__device__ __forceinline__ void bmm_fetch(const float *__restrict__ src, int ld, int i0, int k0, float4 (&r)[2]) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = threadIdx.x + 256 * u;  // 512 float4 per tile
    if (KMAJOR_SRC) {
      const int k = e >> 4, i4 = e & 15;
      r[u] = *reinterpret_cast<const float4 *>(src + (size_t)(k0 + k) * ld + i0 + 4 * i4);
    } else {
      const int i = e >> 3, k4 = e & 7;
      r[u] = *reinterpret_cast<const float4 *>(src + (size_t)(i0 + i) * ld + k0 + 4 * k4);
    }
  }
}
template <bool KMAJOR_SRC>
__device__ __forceinline__ void bmm_stash(float *s, const float4 (&r)[2]) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = threadIdx.x + 256 * u;
    if (KMAJOR_SRC) {
      const int k = e >> 4, i4 = e & 15;
      *reinterpret_cast<float4 *>(s + k * BMM_LD + 4 * i4) = r[u];
    } else {
      const int i = e >> 3, k4 = e & 7;
      s[(4 * k4 + 0) * BMM_LD + i] = r[u].x;
      s[(4 * k4 + 1) * BMM_LD + i] = r[u].y;
      s[(4 * k4 + 2) * BMM_LD + i] = r[u].z;
      s[(4 * k4 + 3) * BMM_LD + i] = r[u].w;
    }
  }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void bmm_f32_k(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C,
                                                 int M, int N, int K, long long sa, long long sb, long long sc) {
  __shared__ __attribute__((aligned(16))) float sA[2][BMM_KC * BMM_LD], sBm[2][BMM_KC * BMM_LD];
  const int b = blockIdx.z, m0 = blockIdx.y * BMM_T, n0 = blockIdx.x * BMM_T;
  A += (size_t)b * sa;
  B += (size_t)b * sb;
  C += (size_t)b * sc;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1, l32 = lane & 31, h = lane >> 5;
  const int lda = TA ? M : K, ldb = TB ? K : N;
  f32x16_b acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float4 ra[2], rb[2];
  bmm_fetch<TA>(A, lda, m0, 0, ra);   // A stored [K,M] when TA: k-major
  bmm_fetch<!TB>(B, ldb, n0, 0, rb);  // B stored [K,N] when !TB: k-major
  bmm_stash<TA>(sA[0], ra);
  bmm_stash<!TB>(sBm[0], rb);
  const int nk = K / BMM_KC;
  for (int ks = 0; ks < nk; ++ks) {
    __syncthreads();
    if (ks + 1 < nk) {
      bmm_fetch<TA>(A, lda, m0, (ks + 1) * BMM_KC, ra);
      bmm_fetch<!TB>(B, ldb, n0, (ks + 1) * BMM_KC, rb);
    }
    const float *pa = sA[ks & 1] + h * BMM_LD + 32 * wr + l32;
    const float *pb = sBm[ks & 1] + h * BMM_LD + 32 * wc + l32;
#pragma unroll
    for (int k = 0; k < BMM_KC; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[k * BMM_LD], pb[k * BMM_LD], acc, 0, 0, 0);
    if (ks + 1 < nk) {
      bmm_stash<TA>(sA[(ks + 1) & 1], ra);
      bmm_stash<!TB>(sBm[(ks + 1) & 1], rb);
    }
  }
  // element e of a lane: row (e & 3) + 8 (e >> 2) + 4 h, column l32 of the wave's 32 x 32 tile
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = m0 + 32 * wr + (e & 3) + 8 * (e >> 2) + 4 * h;
    C[(size_t)row * N + n0 + 32 * wc + l32] = acc[e];
  }
}

}  // namespace hitadv

using namespace hitadv;

extern "C" int hitadv_bmm_f32_supported(int M, int N, int K) {
  return (M > 0 && N > 0 && K > 0 && (M % BMM_T) == 0 && (N % BMM_T) == 0 && (K % BMM_KC) == 0) ? 1 : 0;
}

extern "C" int hitadv_bmm_f32(const float *A, const float *B, float *C, int batches, int M, int N, int K, int trans_a, int trans_b,
                              void *stream) {
  if (!A || !B || !C || batches <= 0 || batches > 65535 || !hitadv_bmm_f32_supported(M, N, K) || ((uintptr_t)A & 15) ||
      ((uintptr_t)B & 15))
    return HITADV_E_ARG;
  const dim3 grid((unsigned)(N / BMM_T), (unsigned)(M / BMM_T), (unsigned)batches);
  hipStream_t s = (hipStream_t)stream;
  const long long sa = (long long)M * K, sb = (long long)K * N, sc = (long long)M * N;
  if (trans_a) {
    if (trans_b) bmm_f32_k<true, true><<<grid, 256, 0, s>>>(A, B, C, M, N, K, sa, sb, sc);
    else bmm_f32_k<true, false><<<grid, 256, 0, s>>>(A, B, C, M, N, K, sa, sb, sc);
  } else {
    if (trans_b) bmm_f32_k<false, true><<<grid, 256, 0, s>>>(A, B, C, M, N, K, sa, sb, sc);
    else bmm_f32_k<false, false><<<grid, 256, 0, s>>>(A, B, C, M, N, K, sa, sb, sc);
  }
  HITADV_LAUNCH_CHECK();
  return 0;
}
