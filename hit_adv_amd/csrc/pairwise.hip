// Pairwise squared distances and fused nearest-neighbour reductions (Chamfer / Hausdorff).
//
//   K1  pairwise3_vec4      materialising P[B,N,M], D = 3.  HBM-store-bound: every lane writes one
//                           float4 per row, a wave writes 1 KiB contiguous; the 4 y points of a lane
//                           live in registers for the whole row tile, x rows come from LDS (broadcast).
//   K2  nn_min3             fused row/column minima, D = 3, no matrix.  fp32-VALU-bound: references
//                           staged in LDS as float4, each lane owns Q queries, the 8 waves of a block
//                           split the reference range and merge (value, index) pairs through LDS.
//   generic-D variants      correctness-first kernels for D != 3 (the [B,3,N] call of quirk Q1).
#include "common.hpp"
#include "hitadv.h"

namespace hitadv {

// ------------------------------------------------------------------------------------ K1
// x = the row point (i), y = the column point (j); the three forms are defined in common.hpp
template <int FORM>
__device__ __forceinline__ float pair_value(float x0, float x1, float x2, float rx, float y0, float y1,
                                            float y2, float ry) {
  return pair_dist<FORM>(x0, x1, x2, rx, y0, y1, y2, ry);
}

constexpr int K1_ROWS = 32;  // 6.31 TB/s vs 6.02 at 16 rows (tools/tune/k1_tune.hip on MI355X)

template <int FORM>
__global__ __launch_bounds__(256) void pairwise3_vec4(const float *__restrict__ x,
                                                      const float *__restrict__ y,
                                                      float *__restrict__ P, int N, int M) {
  __shared__ float xs[K1_ROWS * 4];
  const int b = blockIdx.z;
  const int i0 = blockIdx.y * K1_ROWS;
  const int j0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int rows = min(K1_ROWS, N - i0);
  if (threadIdx.x < rows) {
    const float *xp = x + ((size_t)b * N + i0 + threadIdx.x) * 3;
    float a = xp[0], c = xp[1], d = xp[2];
    xs[threadIdx.x * 4 + 0] = a;
    xs[threadIdx.x * 4 + 1] = c;
    xs[threadIdx.x * 4 + 2] = d;
    xs[threadIdx.x * 4 + 3] = sq_norm<FORM>(a, c, d);
  }
  __syncthreads();
  if (j0 >= M) return;
  const float4 *yp = reinterpret_cast<const float4 *>(y + ((size_t)b * M + j0) * 3);
  const float4 ya = yp[0], yb = yp[1], yc = yp[2];  // (x0 y0 z0 x1)(y1 z1 x2 y2)(z2 x3 y3 z3)
  float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
  if (FORM != HITADV_FORM_DIRECT) {
    r0 = sq_norm<FORM>(ya.x, ya.y, ya.z);
    r1 = sq_norm<FORM>(ya.w, yb.x, yb.y);
    r2 = sq_norm<FORM>(yb.z, yb.w, yc.x);
    r3 = sq_norm<FORM>(yc.y, yc.z, yc.w);
  }
  float *out = P + ((size_t)b * N + i0) * M + j0;
  auto row = [&](int r) {
    const float4 xv = *reinterpret_cast<const float4 *>(&xs[r * 4]);
    float4 v;
    v.x = pair_value<FORM>(xv.x, xv.y, xv.z, xv.w, ya.x, ya.y, ya.z, r0);
    v.y = pair_value<FORM>(xv.x, xv.y, xv.z, xv.w, ya.w, yb.x, yb.y, r1);
    v.z = pair_value<FORM>(xv.x, xv.y, xv.z, xv.w, yb.z, yb.w, yc.x, r2);
    v.w = pair_value<FORM>(xv.x, xv.y, xv.z, xv.w, yc.y, yc.z, yc.w, r3);
    *reinterpret_cast<float4 *>(out + (size_t)r * M) = v;
  };
  if (rows == K1_ROWS) {  // full tile: compile-time trip count, the scheduler can keep stores back to back
#pragma unroll 4
    for (int r = 0; r < K1_ROWS; ++r) row(r);
  } else {
    for (int r = 0; r < rows; ++r) row(r);
  }
}

// Any M / any alignment: one column per lane, dword stores.
template <int FORM>
__global__ __launch_bounds__(256) void pairwise3_scalar(const float *__restrict__ x,
                                                        const float *__restrict__ y,
                                                        float *__restrict__ P, int N, int M) {
  __shared__ float xs[K1_ROWS * 4];
  const int b = blockIdx.z;
  const int i0 = blockIdx.y * K1_ROWS;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int rows = min(K1_ROWS, N - i0);
  if (threadIdx.x < rows) {
    const float *xp = x + ((size_t)b * N + i0 + threadIdx.x) * 3;
    float a = xp[0], c = xp[1], d = xp[2];
    xs[threadIdx.x * 4 + 0] = a;
    xs[threadIdx.x * 4 + 1] = c;
    xs[threadIdx.x * 4 + 2] = d;
    xs[threadIdx.x * 4 + 3] = sq_norm<FORM>(a, c, d);
  }
  __syncthreads();
  if (j >= M) return;
  const float *yp = y + ((size_t)b * M + j) * 3;
  const float y0 = yp[0], y1 = yp[1], y2 = yp[2];
  const float ry = sq_norm<FORM>(y0, y1, y2);
  float *out = P + ((size_t)b * N + i0) * M + j;
  for (int r = 0; r < rows; ++r)
    out[(size_t)r * M] = pair_value<FORM>(xs[r * 4], xs[r * 4 + 1], xs[r * 4 + 2], xs[r * 4 + 3], y0, y1, y2, ry);
}

// Generic D: one wave per output element, lanes stride the feature axis, direct form.
__global__ __launch_bounds__(256) void pairwise_generic(const float *__restrict__ x,
                                                        const float *__restrict__ y,
                                                        float *__restrict__ P, int B, int N, int M, int D) {
  const long long e = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long total = (long long)B * N * M;
  if (e >= total) return;
  const int lane = threadIdx.x & 63;
  const int j = (int)(e % M);
  const int i = (int)((e / M) % N);
  const int b = (int)(e / ((long long)M * N));
  const float *xp = x + ((size_t)b * N + i) * D;
  const float *yp = y + ((size_t)b * M + j) * D;
  float acc = 0.f;
  for (int d = lane; d < D; d += 64) {
    float t = xp[d] - yp[d];
    acc = acc + t * t;
  }
  acc = wave_sum(acc);
  if (lane == 0) P[e] = acc;
}

// ------------------------------------------------------------------------------------ K2
constexpr int K2_Q = 4;            // queries per lane
constexpr int K2_NW = 8;           // waves per block; they split each staged reference chunk
constexpr int K2_RCH = 1024;       // references staged per LDS chunk
constexpr int K2_QB = 64 * K2_Q;   // queries per block
// (Q, NW) = (4, 8) and -fno-slp-vectorize: 15.7 us at B=32, 1024x1024 on MI355X against 24.4 us for
// (2, 4) with hipcc's default SLP packing (tools/tune/k2_tune.hip): 4.3 Tpair/s = 47 T lane-op/s, which
// is the plain-VALU issue ceiling measured by tools/tune/valu_rate.hip (40-48 T lane-op/s).

template <int FORM>
__global__ __launch_bounds__(K2_NW * 64) void nn_min3(const float *__restrict__ x, const float *__restrict__ y,
                                                      int N, int M, float *__restrict__ min_x,
                                                      int32_t *__restrict__ arg_x, float *__restrict__ min_y,
                                                      int32_t *__restrict__ arg_y) {
  __shared__ float4 sref[K2_RCH];
  __shared__ float sval[K2_NW][K2_QB];
  __shared__ int sidx[K2_NW][K2_QB];
  const int b = blockIdx.z;
  const int dir = blockIdx.y;
  const float *qp = dir == 0 ? x : y;
  const float *rp = dir == 0 ? y : x;
  const int nq = dir == 0 ? N : M;
  const int nr = dir == 0 ? M : N;
  float *omin = dir == 0 ? min_x : min_y;
  int32_t *oarg = dir == 0 ? arg_x : arg_y;
  const int q0 = blockIdx.x * K2_QB;
  if (q0 >= nq || omin == nullptr) return;  // block-uniform
  qp += (size_t)b * nq * 3;
  rp += (size_t)b * nr * 3;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // keeps the scan bounds in SGPRs

  float qx[K2_Q], qy[K2_Q], qz[K2_Q], qr[K2_Q], best[K2_Q];
  int bi[K2_Q];
#pragma unroll
  for (int t = 0; t < K2_Q; ++t) {
    int q = q0 + lane + 64 * t;
    q = q < nq ? q : nq - 1;
    qx[t] = qp[q * 3 + 0];
    qy[t] = qp[q * 3 + 1];
    qz[t] = qp[q * 3 + 2];
    qr[t] = sq_norm<FORM>(qx[t], qy[t], qz[t]);  // forms 0 and 1 are symmetric in (query, reference)
    best[t] = __builtin_inff();
    bi[t] = 0;
  }

  for (int c0 = 0; c0 < nr; c0 += K2_RCH) {
    const int cnt = min(K2_RCH, nr - c0);
    __syncthreads();
    for (int p = threadIdx.x; p < cnt; p += K2_NW * 64) {
      const float *s = rp + (size_t)(c0 + p) * 3;
      sref[p] = make_float4(s[0], s[1], s[2], sq_norm<FORM>(s[0], s[1], s[2]));
    }
    __syncthreads();
    const int per = K2_RCH / K2_NW;
    const int lo = wave * per, hi = min(lo + per, cnt);
#pragma unroll 4
    for (int p = lo; p < hi; ++p) {
      const float4 r = sref[p];
#pragma unroll
      for (int t = 0; t < K2_Q; ++t) {
        const float d = pair_dist<FORM>(qx[t], qy[t], qz[t], qr[t], r.x, r.y, r.z, r.w);
        const bool lt = d < best[t];
        best[t] = lt ? d : best[t];
        bi[t] = lt ? c0 + p : bi[t];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < K2_Q; ++t) {
    sval[wave][lane + 64 * t] = best[t];
    sidx[wave][lane + 64 * t] = bi[t];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < K2_QB; e += K2_NW * 64) {
    const int q = q0 + e;
    if (q < nq) {
      float v = sval[0][e];
      int ix = sidx[0][e];
#pragma unroll
      for (int w = 1; w < K2_NW; ++w) {
        const float ov = sval[w][e];
        const int oi = sidx[w][e];
        const bool take = (ov < v) || (ov == v && oi < ix);
        v = take ? ov : v;
        ix = take ? oi : ix;
      }
      omin[(size_t)b * nq + q] = v;
      oarg[(size_t)b * nq + q] = ix;
    }
  }
}

// Row / column minima of a materialised non-negative P[B,N,M] (generic-D path): one wave per output.
__global__ __launch_bounds__(256) void minreduce_matrix(const float *__restrict__ P, int B, int N, int M,
                                                        float *__restrict__ min_x, int32_t *__restrict__ arg_x,
                                                        float *__restrict__ min_y, int32_t *__restrict__ arg_y) {
  const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long long nrow = (long long)B * N, ncol = (long long)B * M;
  if (w < nrow) {
    if (min_x == nullptr) return;
    const float *row = P + w * M;
    unsigned long long key = ~0ull;
    for (int j = lane; j < M; j += 64) {
      unsigned long long k = ((unsigned long long)fbits(row[j]) << 32) | (uint32_t)j;
      key = k < key ? k : key;
    }
    key = wave_min_u64(key);
    if (lane == 0) {
      min_x[w] = __uint_as_float((uint32_t)(key >> 32));
      arg_x[w] = (int32_t)(key & 0xffffffffu);
    }
  } else if (w < nrow + ncol) {
    if (min_y == nullptr) return;
    const long long c = w - nrow;
    const int b = (int)(c / M), j = (int)(c % M);
    const float *col = P + (size_t)b * N * M + j;
    unsigned long long key = ~0ull;
    for (int i = lane; i < N; i += 64) {
      unsigned long long k = ((unsigned long long)fbits(col[(size_t)i * M]) << 32) | (uint32_t)i;
      key = k < key ? k : key;
    }
    key = wave_min_u64(key);
    if (lane == 0) {
      min_y[c] = __uint_as_float((uint32_t)(key >> 32));
      arg_y[c] = (int32_t)(key & 0xffffffffu);
    }
  }
}

// ------------------------------------------------------------------------------------ backward
// grad_q[i] = 2 g_own[i] (q_i - r_arg_own[i]) + sum_{j: arg_other[j]==i} 2 g_other[j] (q_i - r_j)
// One lane per query; the other side's (arg, g, point) triples stream through LDS.  Fixed order.
constexpr int BW_CH = 1024;
__global__ __launch_bounds__(256) void nn_min_bwd3(const float *__restrict__ x, const float *__restrict__ y,
                                                   const int32_t *__restrict__ arg_x,
                                                   const int32_t *__restrict__ arg_y,
                                                   const float *__restrict__ g_min_x,
                                                   const float *__restrict__ g_min_y, int N, int M,
                                                   float *__restrict__ grad_x, float *__restrict__ grad_y) {
  __shared__ float4 sref[BW_CH];
  __shared__ __attribute__((aligned(16))) int sarg[BW_CH];
  const int b = blockIdx.z, dir = blockIdx.y;
  const float *qp = dir == 0 ? x : y;
  const float *rp = dir == 0 ? y : x;
  const int nq = dir == 0 ? N : M, nr = dir == 0 ? M : N;
  const int32_t *a_own = dir == 0 ? arg_x : arg_y;
  const int32_t *a_oth = dir == 0 ? arg_y : arg_x;
  const float *g_own = dir == 0 ? g_min_x : g_min_y;
  const float *g_oth = dir == 0 ? g_min_y : g_min_x;
  float *gout = dir == 0 ? grad_x : grad_y;
  if (gout == nullptr || blockIdx.x * 256 >= nq) return;
  qp += (size_t)b * nq * 3;
  rp += (size_t)b * nr * 3;
  const int q = blockIdx.x * 256 + threadIdx.x;
  const bool live = q < nq;
  const int qq = live ? q : nq - 1;
  const float qx = qp[qq * 3], qy = qp[qq * 3 + 1], qz = qp[qq * 3 + 2];
  float ax = 0.f, ay = 0.f, az = 0.f;
  if (g_own != nullptr && a_own != nullptr) {
    const int a = a_own[(size_t)b * nq + qq];
    const float g2 = 2.0f * g_own[(size_t)b * nq + qq];
    ax = g2 * (qx - rp[a * 3]);
    ay = g2 * (qy - rp[a * 3 + 1]);
    az = g2 * (qz - rp[a * 3 + 2]);
  }
  if (g_oth != nullptr && a_oth != nullptr) {
    for (int c0 = 0; c0 < nr; c0 += BW_CH) {
      const int cnt = min(BW_CH, nr - c0);
      __syncthreads();
      for (int p = threadIdx.x; p < cnt; p += 256) {
        const float *s = rp + (size_t)(c0 + p) * 3;
        sref[p] = make_float4(s[0], s[1], s[2], 2.0f * g_oth[(size_t)b * nr + c0 + p]);
        sarg[p] = a_oth[(size_t)b * nr + c0 + p];
      }
      for (int p = cnt + threadIdx.x; p < ((cnt + 3) & ~3); p += 256) sarg[p] = -1;  // pad the last int4
      __syncthreads();
      // four arg entries per broadcast LDS read; matches are rare (each source point hits one query)
      for (int p = 0; p < cnt; p += 4) {
        const int4 a = *reinterpret_cast<const int4 *>(&sarg[p]);
        if ((a.x == q) | (a.y == q) | (a.z == q) | (a.w == q)) {
          const int av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (av[u] == q && p + u < cnt) {
              const float4 r = sref[p + u];
              ax = ax + r.w * (qx - r.x);
              ay = ay + r.w * (qy - r.y);
              az = az + r.w * (qz - r.z);
            }
        }
      }
    }
  }
  if (live) {
    float *o = gout + ((size_t)b * nq + q) * 3;
    o[0] = ax;
    o[1] = ay;
    o[2] = az;
  }
}

// Generic D: one block per query row, lanes stride the feature axis.
__global__ __launch_bounds__(256) void nn_min_bwd_generic(
    const float *__restrict__ x, const float *__restrict__ y, const int32_t *__restrict__ arg_x,
    const int32_t *__restrict__ arg_y, const float *__restrict__ g_min_x, const float *__restrict__ g_min_y,
    int N, int M, int D, float *__restrict__ grad_x, float *__restrict__ grad_y) {
  const int b = blockIdx.z, dir = blockIdx.y, q = blockIdx.x;
  const float *qp = dir == 0 ? x : y;
  const float *rp = dir == 0 ? y : x;
  const int nq = dir == 0 ? N : M, nr = dir == 0 ? M : N;
  const int32_t *a_own = dir == 0 ? arg_x : arg_y;
  const int32_t *a_oth = dir == 0 ? arg_y : arg_x;
  const float *g_own = dir == 0 ? g_min_x : g_min_y;
  const float *g_oth = dir == 0 ? g_min_y : g_min_x;
  float *gout = dir == 0 ? grad_x : grad_y;
  if (gout == nullptr || q >= nq) return;
  qp += ((size_t)b * nq + q) * D;
  rp += (size_t)b * nr * D;
  gout += ((size_t)b * nq + q) * D;
  for (int d = threadIdx.x; d < D; d += 256) {
    const float qv = qp[d];
    float acc = 0.f;
    if (g_own != nullptr && a_own != nullptr) {
      const int a = a_own[(size_t)b * nq + q];
      acc = 2.0f * g_own[(size_t)b * nq + q] * (qv - rp[(size_t)a * D + d]);
    }
    if (g_oth != nullptr && a_oth != nullptr) {
      for (int j = 0; j < nr; ++j)
        if (a_oth[(size_t)b * nr + j] == q)
          acc = acc + 2.0f * g_oth[(size_t)b * nr + j] * (qv - rp[(size_t)j * D + d]);
    }
    gout[d] = acc;
  }
}

}  // namespace hitadv

using namespace hitadv;

extern "C" int hitadv_pairwise_sqdist(const float *x, const float *y, float *P, int B, int N, int M, int D,
                                      int form, void *stream) {
  if (!x || !y || !P || B <= 0 || N <= 0 || M <= 0 || D <= 0) return HITADV_E_ARG;
  if (form != HITADV_FORM_DIRECT && form != HITADV_FORM_GRAM && form != HITADV_FORM_GRAM_KNN) return HITADV_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (D == 3) {
    const bool vec = (M % 4 == 0) && (((uintptr_t)y & 15) == 0) && (((uintptr_t)P & 15) == 0);
    if (vec) {
      dim3 grid((M + 1023) / 1024, (N + K1_ROWS - 1) / K1_ROWS, B);
      if (form == HITADV_FORM_DIRECT)
        pairwise3_vec4<HITADV_FORM_DIRECT><<<grid, 256, 0, s>>>(x, y, P, N, M);
      else if (form == HITADV_FORM_GRAM)
        pairwise3_vec4<HITADV_FORM_GRAM><<<grid, 256, 0, s>>>(x, y, P, N, M);
      else
        pairwise3_vec4<HITADV_FORM_GRAM_KNN><<<grid, 256, 0, s>>>(x, y, P, N, M);
    } else {
      dim3 grid((M + 255) / 256, (N + K1_ROWS - 1) / K1_ROWS, B);
      if (form == HITADV_FORM_DIRECT)
        pairwise3_scalar<HITADV_FORM_DIRECT><<<grid, 256, 0, s>>>(x, y, P, N, M);
      else if (form == HITADV_FORM_GRAM)
        pairwise3_scalar<HITADV_FORM_GRAM><<<grid, 256, 0, s>>>(x, y, P, N, M);
      else
        pairwise3_scalar<HITADV_FORM_GRAM_KNN><<<grid, 256, 0, s>>>(x, y, P, N, M);
    }
  } else {
    const long long total = (long long)B * N * M;
    pairwise_generic<<<(unsigned)((total + 3) / 4), 256, 0, s>>>(x, y, P, B, N, M, D);
  }
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_nn_min(const float *x, const float *y, int B, int N, int M, int D, int form, float *min_x,
                             int32_t *arg_x, float *min_y, int32_t *arg_y, float *scratch, void *stream) {
  if (!x || !y || B <= 0 || N <= 0 || M <= 0 || D <= 0) return HITADV_E_ARG;
  if (form != HITADV_FORM_DIRECT && !(form == HITADV_FORM_GRAM && D == 3)) return HITADV_E_ARG;
  if ((min_x == nullptr) != (arg_x == nullptr) || (min_y == nullptr) != (arg_y == nullptr)) return HITADV_E_ARG;
  if (!min_x && !min_y) return HITADV_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (D == 3) {
    const int nmax = N > M ? N : M;
    dim3 grid((nmax + K2_QB - 1) / K2_QB, 2, B);
    if (form == HITADV_FORM_DIRECT)
      nn_min3<HITADV_FORM_DIRECT><<<grid, K2_NW * 64, 0, s>>>(x, y, N, M, min_x, arg_x, min_y, arg_y);
    else
      nn_min3<HITADV_FORM_GRAM><<<grid, K2_NW * 64, 0, s>>>(x, y, N, M, min_x, arg_x, min_y, arg_y);
  } else {
    if (!scratch) return HITADV_E_ARG;
    const long long total = (long long)B * N * M;
    pairwise_generic<<<(unsigned)((total + 3) / 4), 256, 0, s>>>(x, y, scratch, B, N, M, D);
    const long long waves = (long long)B * (N + M);
    minreduce_matrix<<<(unsigned)((waves + 3) / 4), 256, 0, s>>>(scratch, B, N, M, min_x, arg_x, min_y, arg_y);
  }
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_nn_min_bwd(const float *x, const float *y, const int32_t *arg_x, const int32_t *arg_y,
                                 const float *g_min_x, const float *g_min_y, int B, int N, int M, int D,
                                 float *grad_x, float *grad_y, void *stream) {
  if (!x || !y || B <= 0 || N <= 0 || M <= 0 || D <= 0 || (!grad_x && !grad_y)) return HITADV_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int nmax = N > M ? N : M;
  if (D == 3) {
    dim3 grid((nmax + 255) / 256, 2, B);
    nn_min_bwd3<<<grid, 256, 0, s>>>(x, y, arg_x, arg_y, g_min_x, g_min_y, N, M, grad_x, grad_y);
  } else {
    dim3 grid(nmax, 2, B);
    nn_min_bwd_generic<<<grid, 256, 0, s>>>(x, y, arg_x, arg_y, g_min_x, g_min_y, N, M, D, grad_x, grad_y);
  }
  HITADV_LAUNCH_CHECK();
  return 0;
}
