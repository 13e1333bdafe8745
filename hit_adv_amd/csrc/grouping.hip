// ball_query, group_points, three_nn, three_interpolate and their gradients
// (pointnet2_ops natives, re-designed for wave64; gradients are owner-computes, no atomics).
#include "common.hpp"
#include "hitadv.h"

namespace hitadv {

// One WAVE per query: 64 candidate points per step, hits compacted in ascending index order with
// ballot + prefix popcount, so the "first nsample hits in index order" rule of
// ball_query_gpu.cu:9-44 is kept without a serial scan.  Empty ball -> zeros, short ball -> padded
// with the first hit.
template <bool INCLUSIVE, typename IdxT>
__global__ __launch_bounds__(256) void ball_query_k(int n, int m, float radius2, int nsample,
                                                    const float *__restrict__ new_xyz,
                                                    const float *__restrict__ xyz, IdxT *__restrict__ idx,
                                                    long long nquery, int empty_value) {
  const long long qid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qid >= nquery) return;
  const int lane = threadIdx.x & 63;
  const int b = (int)(qid / m);
  const float *q = new_xyz + qid * 3;
  const float qx = q[0], qy = q[1], qz = q[2];
  const float *P = xyz + (size_t)b * n * 3;
  IdxT *out = idx + qid * nsample;
  int cnt = 0, first = empty_value;
  for (int k0 = 0; k0 < n && cnt < nsample; k0 += 64) {
    const int k = k0 + lane;
    bool hit = false;
    if (k < n) {
      const float d2 = sqdist3(qx, qy, qz, P[k * 3], P[k * 3 + 1], P[k * 3 + 2]);
      hit = INCLUSIVE ? (d2 <= radius2) : (d2 < radius2);
    }
    const unsigned long long mask = __ballot(hit);
    if (mask) {
      if (cnt == 0) first = k0 + __builtin_ctzll(mask);
      const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
      if (hit && pos < nsample) out[pos] = (IdxT)k;
      cnt += __popcll(mask);
    }
  }
  cnt = cnt < nsample ? cnt : nsample;
  for (int l = cnt + lane; l < nsample; l += 64) out[l] = (IdxT)first;
}

__global__ __launch_bounds__(256) void group_points_k(int c, int n, int npoints, int nsample,
                                                      const float *__restrict__ points,
                                                      const int32_t *__restrict__ idx, float *__restrict__ out,
                                                      long long total) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const long long per = (long long)npoints * nsample;
  const long long jk = e % per;
  const long long bc = e / per;
  const int b = (int)(bc / c);
  out[e] = points[bc * n + idx[(size_t)b * per + jk]];
}

__global__ __launch_bounds__(256) void group_points_grad_k(int c, int n, int npoints, int nsample,
                                                           const float *__restrict__ grad_out,
                                                           const int32_t *__restrict__ idx,
                                                           float *__restrict__ grad_points) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int per = npoints * nsample;
  float *gp = grad_points + (size_t)b * c * n + i;
  for (int l = 0; l < c; ++l) gp[(size_t)l * n] = 0.f;
  const int32_t *ip = idx + (size_t)b * per;
  const float *go = grad_out + (size_t)b * c * per;
  for (int e = 0; e < per; ++e)
    if (ip[e] == i)
      for (int l = 0; l < c; ++l) gp[(size_t)l * n] += go[(size_t)l * per + e];
}

constexpr int TN_CH = 1024;
__global__ __launch_bounds__(256) void three_nn_k(int n, int m, const float *__restrict__ unknown,
                                                  const float *__restrict__ known, float *__restrict__ dist2,
                                                  int32_t *__restrict__ idx) {
  __shared__ float4 sk[TN_CH];
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const bool live = j < n;
  const float *u = unknown + ((size_t)b * n + (live ? j : n - 1)) * 3;
  const float ux = u[0], uy = u[1], uz = u[2];
  known += (size_t)b * m * 3;
  float b1 = __builtin_inff(), b2 = b1, b3 = b1;
  int i1 = 0, i2 = 0, i3 = 0;
  for (int c0 = 0; c0 < m; c0 += TN_CH) {
    const int cnt = min(TN_CH, m - c0);
    __syncthreads();
    for (int r = threadIdx.x; r < cnt; r += 256) {
      const float *s = known + (size_t)(c0 + r) * 3;
      sk[r] = make_float4(s[0], s[1], s[2], 0.f);
    }
    __syncthreads();
    for (int r = 0; r < cnt; ++r) {
      const float4 v = sk[r];
      const float d = sqdist3(ux, uy, uz, v.x, v.y, v.z);
      const int k = c0 + r;
      if (d < b1) {
        b3 = b2; i3 = i2;
        b2 = b1; i2 = i1;
        b1 = d; i1 = k;
      } else if (d < b2) {
        b3 = b2; i3 = i2;
        b2 = d; i2 = k;
      } else if (d < b3) {
        b3 = d; i3 = k;
      }
    }
  }
  if (live) {
    const size_t o = ((size_t)b * n + j) * 3;
    dist2[o] = b1; dist2[o + 1] = b2; dist2[o + 2] = b3;
    idx[o] = i1; idx[o + 1] = i2; idx[o + 2] = i3;
  }
}

__global__ __launch_bounds__(256) void three_interpolate_k(int c, int m, int n,
                                                           const float *__restrict__ points,
                                                           const int32_t *__restrict__ idx,
                                                           const float *__restrict__ weight,
                                                           float *__restrict__ out, long long total) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int j = (int)(e % n);
  const long long bc = e / n;
  const int b = (int)(bc / c);
  const size_t o = ((size_t)b * n + j) * 3;
  const float *p = points + bc * m;
  out[e] = (p[idx[o]] * weight[o] + p[idx[o + 1]] * weight[o + 1]) + p[idx[o + 2]] * weight[o + 2];
}

__global__ __launch_bounds__(256) void three_interpolate_grad_k(int c, int n, int m,
                                                                const float *__restrict__ grad_out,
                                                                const int32_t *__restrict__ idx,
                                                                const float *__restrict__ weight,
                                                                float *__restrict__ grad_points) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  float *gp = grad_points + (size_t)b * c * m + i;
  for (int l = 0; l < c; ++l) gp[(size_t)l * m] = 0.f;
  const int32_t *ip = idx + (size_t)b * n * 3;
  const float *wp = weight + (size_t)b * n * 3;
  const float *go = grad_out + (size_t)b * c * n;
  for (int e = 0; e < n * 3; ++e)
    if (ip[e] == i) {
      const float w = wp[e];
      const int j = e / 3;
      for (int l = 0; l < c; ++l) gp[(size_t)l * m] += go[(size_t)l * n + j] * w;
    }
}

}  // namespace hitadv

using namespace hitadv;

extern "C" int hitadv_query_ball_point(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                                       const float *xyz, int32_t *idx, void *stream) {
  if (!new_xyz || !xyz || !idx || b <= 0 || n <= 0 || m <= 0 || nsample <= 0) return HITADV_E_ARG;
  const long long nq = (long long)b * m;
  ball_query_k<false, int32_t><<<(unsigned)((nq + 3) / 4), 256, 0, (hipStream_t)stream>>>(
      n, m, radius * radius, nsample, new_xyz, xyz, idx, nq, 0);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_query_ball_point_inclusive(int b, int n, int m, float radius, int nsample,
                                                 const float *new_xyz, const float *xyz, int64_t *idx, void *stream) {
  if (!new_xyz || !xyz || !idx || b <= 0 || n <= 0 || m <= 0 || nsample <= 0) return HITADV_E_ARG;
  const long long nq = (long long)b * m;
  ball_query_k<true, int64_t><<<(unsigned)((nq + 3) / 4), 256, 0, (hipStream_t)stream>>>(
      n, m, radius * radius, nsample, new_xyz, xyz, idx, nq, n);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                                   const int32_t *idx, float *out, void *stream) {
  if (!points || !idx || !out || b <= 0 || c <= 0 || n <= 0 || npoints <= 0 || nsample <= 0) return HITADV_E_ARG;
  const long long total = (long long)b * c * npoints * nsample;
  group_points_k<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(c, n, npoints, nsample, points,
                                                                                   idx, out, total);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                        const int32_t *idx, float *grad_points, void *stream) {
  if (!grad_out || !idx || !grad_points || b <= 0 || c <= 0 || n <= 0 || npoints <= 0 || nsample <= 0)
    return HITADV_E_ARG;
  dim3 grid((n + 255) / 256, b);
  group_points_grad_k<<<grid, 256, 0, (hipStream_t)stream>>>(c, n, npoints, nsample, grad_out, idx, grad_points);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                               int32_t *idx, void *stream) {
  if (!unknown || !known || !dist2 || !idx || b <= 0 || n <= 0 || m <= 0) return HITADV_E_ARG;
  dim3 grid((n + 255) / 256, b);
  three_nn_k<<<grid, 256, 0, (hipStream_t)stream>>>(n, m, unknown, known, dist2, idx);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_three_interpolate(int b, int c, int m, int n, const float *points, const int32_t *idx,
                                        const float *weight, float *out, void *stream) {
  if (!points || !idx || !weight || !out || b <= 0 || c <= 0 || m <= 0 || n <= 0) return HITADV_E_ARG;
  const long long total = (long long)b * c * n;
  three_interpolate_k<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(c, m, n, points, idx,
                                                                                        weight, out, total);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                             const int32_t *idx, const float *weight, float *grad_points,
                                             void *stream) {
  if (!grad_out || !idx || !weight || !grad_points || b <= 0 || c <= 0 || n <= 0 || m <= 0) return HITADV_E_ARG;
  dim3 grid((m + 255) / 256, b);
  three_interpolate_grad_k<<<grid, 256, 0, (hipStream_t)stream>>>(c, n, m, grad_out, idx, weight, grad_points);
  HITADV_LAUNCH_CHECK();
  return 0;
}
