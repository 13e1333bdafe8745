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

// ---------------------------------------------------------------------------------------------------
// EdgeConv's neighbour reduction (model/dgcnn_cls.py:16-43 + conv/bn/LeakyReLU/max of :93-112), after the algebraic
// split  W [x_j - x_i ; x_i] = Wa x_j + (Wb - Wa) x_i :   out[i,c] = lrelu( V[i,c] + max_{j in nbr(i)} U[j,c] ).
// The k-times larger edge tensor [B,2C,N,k] never exists; the per-edge 1x1 convolution became two per-POINT GEMMs
// (U, V) and this gather-max.  A lane owns 4 consecutive channels of one point (float4 rows of U: coalesced).
namespace hitadv {

__global__ __launch_bounds__(256) void edge_max_fwd_k(const float *__restrict__ U, const float *__restrict__ V,
                                                      const int64_t *__restrict__ idx, int N, int C, int k, float slope,
                                                      float *__restrict__ out, int32_t *__restrict__ arg,
                                                      long long total4) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;  // (b, i, c4)
  if (e >= total4) return;
  const int c4n = C >> 2;
  const int c4 = (int)(e % c4n);
  const long long bi = e / c4n;  // b * N + i
  const long long b = bi / N;
  const int64_t *nb = idx + bi * k;
  float4 best = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff());
  int4 bj = make_int4(0, 0, 0, 0);
  for (int t = 0; t < k; ++t) {
    const int j = (int)nb[t];
    const float4 u = *reinterpret_cast<const float4 *>(U + ((size_t)(b * N + j)) * C + 4 * c4);
    if (u.x > best.x) { best.x = u.x; bj.x = j; }
    if (u.y > best.y) { best.y = u.y; bj.y = j; }
    if (u.z > best.z) { best.z = u.z; bj.z = j; }
    if (u.w > best.w) { best.w = u.w; bj.w = j; }
  }
  const float4 v = *reinterpret_cast<const float4 *>(V + (size_t)bi * C + 4 * c4);
  float4 o = make_float4(best.x + v.x, best.y + v.y, best.z + v.z, best.w + v.w);
  o.x = o.x > 0.f ? o.x : o.x * slope; o.y = o.y > 0.f ? o.y : o.y * slope;
  o.z = o.z > 0.f ? o.z : o.z * slope; o.w = o.w > 0.f ? o.w : o.w * slope;
  *reinterpret_cast<float4 *>(out + (size_t)bi * C + 4 * c4) = o;
  *reinterpret_cast<int4 *>(arg + (size_t)bi * C + 4 * c4) = bj;
}

// dV[i,c] = dout[i,c] * lrelu'(out[i,c]);  dU[arg[i,c], c] += dV[i,c]  (dU zero-filled by the caller's launch order:
// this kernel is preceded by a memset on the same stream).  The scatter uses float atomics: a point can be the winning
// neighbour of many others, and which of them are there is only known through the forward's arg table.
__global__ __launch_bounds__(256) void edge_max_bwd_k(const float *__restrict__ dout, const float *__restrict__ out,
                                                      const int32_t *__restrict__ arg, int N, int C, float slope,
                                                      float *__restrict__ dU, float *__restrict__ dV, long long total) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;  // (b, i, c)
  if (e >= total) return;
  const int c = (int)(e % C);
  const long long b = e / ((long long)N * C);
  const float g = dout[e] * (out[e] > 0.f ? 1.0f : slope);
  dV[e] = g;
  atomicAdd(dU + ((size_t)(b * N + arg[e])) * C + c, g);
}

}  // namespace hitadv

extern "C" int hitadv_edge_max_fwd(const float *U, const float *V, const int64_t *idx, int B, int N, int C, int k,
                                   float slope, float *out, int32_t *arg, void *stream) {
  if (!U || !V || !idx || !out || !arg || B <= 0 || N <= 0 || C <= 0 || (C & 3) || k <= 0 ||
      (((uintptr_t)U | (uintptr_t)V | (uintptr_t)out | (uintptr_t)arg) & 15))
    return HITADV_E_ARG;
  const long long total4 = (long long)B * N * (C >> 2);
  hitadv::edge_max_fwd_k<<<(unsigned)((total4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(U, V, idx, N, C, k, slope, out,
                                                                                         arg, total4);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_edge_max_bwd(const float *dout, const float *out, const int32_t *arg, int B, int N, int C,
                                   float slope, float *dU, float *dV, void *stream) {
  if (!dout || !out || !arg || !dU || !dV || B <= 0 || N <= 0 || C <= 0) return HITADV_E_ARG;
  const long long total = (long long)B * N * C;
  hipError_t e = hipMemsetAsync(dU, 0, (size_t)total * sizeof(float), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  hitadv::edge_max_bwd_k<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(dout, out, arg, N, C, slope, dU,
                                                                                        dV, total);
  HITADV_LAUNCH_CHECK();
  return 0;
}
