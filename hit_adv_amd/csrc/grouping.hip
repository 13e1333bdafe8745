// ball_query, group_points, three_nn, three_interpolate and their gradients
// (pointnet2_ops natives, re-designed for wave64; gradients are owner-computes, no atomics).
#include "common.hpp"
#include "hitadv.h"

namespace hitadv {

// One WAVE per query: 64 candidate points per step, hits compacted in ascending index order with
// ballot + prefix popcount, so the "first nsample hits in index order" rule of
// ball_query_gpu.cu:9-44 is kept without a serial scan.  Empty ball -> zeros, short ball -> padded
// with the first hit.
// FORM: how the squared distance is evaluated (common.hpp::pair_dist): 0 = direct (the CUDA extension), 3 = the victims'
// Gram-form square_distance(new_xyz, xyz) in torch's fp32 arithmetic (model/pointnet2_utils.py:19-41, :87-107).
template <bool INCLUSIVE, typename IdxT, int FORM = 0>
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
  const float rq = sq_norm<FORM>(qx, qy, qz);
  const float *P = xyz + (size_t)b * n * 3;
  IdxT *out = idx + qid * nsample;
  int cnt = 0, first = empty_value;
  for (int k0 = 0; k0 < n && cnt < nsample; k0 += 64) {
    const int k = k0 + lane;
    bool hit = false;
    if (k < n) {
      const float x = P[k * 3], y = P[k * 3 + 1], z = P[k * 3 + 2];
      const float d2 = pair_dist<FORM>(qx, qy, qz, rq, x, y, z, sq_norm<FORM>(x, y, z));
      hit = INCLUSIVE ? !(d2 > radius2) : (d2 < radius2);
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

extern "C" int hitadv_query_ball_point_victim(int b, int n, int m, float radius2, int nsample, int form,
                                              const float *new_xyz, const float *xyz, int64_t *idx, void *stream) {
  if (!new_xyz || !xyz || !idx || b <= 0 || n <= 0 || m <= 0 || nsample <= 0) return HITADV_E_ARG;
  if (form != HITADV_FORM_DIRECT && form != HITADV_FORM_SQUARE_DISTANCE) return HITADV_E_ARG;
  const long long nq = (long long)b * m;
  const unsigned grid = (unsigned)((nq + 3) / 4);
  if (form == HITADV_FORM_DIRECT)
    ball_query_k<true, int64_t, 0><<<grid, 256, 0, (hipStream_t)stream>>>(n, m, radius2, nsample, new_xyz, xyz, idx, nq, n);
  else
    ball_query_k<true, int64_t, 3><<<grid, 256, 0, (hipStream_t)stream>>>(n, m, radius2, nsample, new_xyz, xyz, idx, nq, n);
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
                                                      int ld, const int64_t *__restrict__ idx, int N, int C, int k,
                                                      float slope, float *__restrict__ out, int32_t *__restrict__ arg,
                                                      long long total4) {
  long long blk = blockIdx.x;  // blocks of one cloud on one XCD: its U rows are gathered k times, from that L2
  if ((gridDim.x & 7) == 0) blk = (blk & 7) * (gridDim.x >> 3) + (blk >> 3);
  const long long e = blk * 256 + threadIdx.x;  // (b, i, c4)
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
    const float4 u = *reinterpret_cast<const float4 *>(U + ((size_t)(b * N + j)) * ld + 4 * c4);
    if (u.x > best.x) { best.x = u.x; bj.x = j; }
    if (u.y > best.y) { best.y = u.y; bj.y = j; }
    if (u.z > best.z) { best.z = u.z; bj.z = j; }
    if (u.w > best.w) { best.w = u.w; bj.w = j; }
  }
  const float4 v = *reinterpret_cast<const float4 *>(V + (size_t)bi * ld + 4 * c4);
  float4 o = make_float4(best.x + v.x, best.y + v.y, best.z + v.z, best.w + v.w);
  o.x = o.x > 0.f ? o.x : o.x * slope; o.y = o.y > 0.f ? o.y : o.y * slope;
  o.z = o.z > 0.f ? o.z : o.z * slope; o.w = o.w > 0.f ? o.w : o.w * slope;
  *reinterpret_cast<float4 *>(out + (size_t)bi * C + 4 * c4) = o;
  *reinterpret_cast<int4 *>(arg + (size_t)bi * C + 4 * c4) = bj;
}

// Backward.  dV[i,c] = dout[i,c] * lrelu'(out[i,c]);  dU[j,c] = sum over the points i that list j as a neighbour AND whose
// winner for channel c was j (arg[i,c] == j) of dV[i,c].  A scatter with float atomics would do it in one pass, but the
// order of the additions -- and with it the low bits of the result -- would change from run to run.  Instead the
// neighbour table is turned around first (who lists j?), and every (j, c) then GATHERS its terms in ascending i:
// same bits every run, no zero-fill, no atomics on floats.
//
// reverse_graph_k (S neighbour lists of k entries pointing at N targets; S == N for EdgeConv): a block owns J
// consecutive target points j of one cloud.  It walks the cloud's S*k edges once: edges
// into its range set bit i of row j in an LDS bit matrix and bump j's edge count (integer LDS atomics: order-free);
// edges into LOWER ranges are only counted -- that count is where this block's share of the list array begins, so no
// block waits for another.  A row's set bits, read in ascending order, are the in-list.  A neighbour list that repeats
// an entry yields one bit but two counts: segments may end with unused slots, hence span[j] = (begin, end) rather than
// a compact CSR.  The gather tests arg[i,c] == j once per listed i, which is what the de-duplication wants.
#define RG_THREADS 256
#define RG_BITMAP_WORDS 15360  // 60 KB: two blocks per CU

__global__ __launch_bounds__(RG_THREADS) void reverse_graph_k(const int64_t *__restrict__ idx, int S, int N, int k, int W,
                                                              int J, int2 *__restrict__ span,
                                                              int32_t *__restrict__ col) {  // S lists -> N targets
  extern __shared__ uint32_t rg_sm[];
  uint32_t *bm = rg_sm;                               // J rows of W words (+1 pad: a thread per row walks its words)
  int32_t *cnt = (int32_t *)(rg_sm + RG_BITMAP_WORDS);  // J edge counts, then their exclusive prefix
  int32_t *red = cnt + RG_THREADS;                    // block reduction / scan workspace (2 x RG_THREADS)
  const int b = blockIdx.y, tid = threadIdx.x;
  const int j0 = blockIdx.x * J, rows = min(J, N - j0);
  const int E = S * k, RS = W + 1;
  const int64_t *nb = idx + (size_t)b * E;
  for (int w = tid; w < rows * RS; w += RG_THREADS) bm[w] = 0u;
  cnt[tid] = 0;
  __syncthreads();
  int below = 0;
  auto edge = [&](int e, long long j) {
    if (j < 0 || j >= N) return;  // the reference does not bound-check idx either; such an edge feeds nobody
    if (j < j0) {
      ++below;
    } else if (j < j0 + rows) {
      const int i = e / k;
      atomicOr(&bm[(int)(j - j0) * RS + (i >> 5)], 1u << (i & 31));
      atomicAdd(&cnt[(int)(j - j0)], 1);
    }
  };
  int e = tid;
  for (; e + 7 * RG_THREADS < E; e += 8 * RG_THREADS) {  // eight table entries in flight per lane
    long long j[8];
    for (int u = 0; u < 8; ++u) j[u] = nb[e + u * RG_THREADS];
    for (int u = 0; u < 8; ++u) edge(e + u * RG_THREADS, j[u]);
  }
  for (; e < E; e += RG_THREADS) edge(e, nb[e]);
  // block sum of `below`, and the exclusive scan of cnt[0..J) (J <= RG_THREADS: one entry per thread)
  int32_t *src = red, *dst = red + RG_THREADS;
  src[tid] = below;
  __syncthreads();
  for (int d = RG_THREADS >> 1; d > 0; d >>= 1) {
    if (tid < d) src[tid] += src[tid + d];
    __syncthreads();
  }
  const int base = src[0];
  __syncthreads();
  const int mine = cnt[tid];
  src[tid] = mine;
  __syncthreads();
  for (int d = 1; d < RG_THREADS; d <<= 1) {
    dst[tid] = src[tid] + (tid >= d ? src[tid - d] : 0);
    __syncthreads();
    int32_t *t = src; src = dst; dst = t;
  }
  if (tid < rows) {
    const int begin = base + src[tid] - mine;
    int pos = begin;
    int32_t *cl = col + (size_t)b * E;
    for (int w = 0; w < W; ++w) {
      uint32_t bits = bm[tid * RS + w];
      while (bits) {
        cl[pos++] = (w << 5) + __builtin_ctz(bits);
        bits &= bits - 1;
      }
    }
    span[(size_t)b * N + j0 + tid] = make_int2(begin, pos);
  }
}

// One lane per (b, j, 4 channels), like the forward.  Blocks are renumbered so that the blocks of one cloud run on one
// XCD: every (i, c) row of arg / dout / out is visited once per neighbour that lists... k times in all, and those
// re-reads should hit that XCD's L2.
__global__ __launch_bounds__(256) void edge_max_bwd_k(const float *__restrict__ dout, const float *__restrict__ out,
                                                      const int32_t *__restrict__ arg, const int2 *__restrict__ span,
                                                      const int32_t *__restrict__ col, int N, int C, int E, float slope,
                                                      float *__restrict__ dU, float *__restrict__ dV, int ldg,
                                                      long long total4) {
  long long blk = blockIdx.x;
  if ((gridDim.x & 7) == 0) blk = (blk & 7) * (gridDim.x >> 3) + (blk >> 3);
  const long long e = blk * 256 + threadIdx.x;  // (b, j, c4)
  if (e >= total4) return;
  const int c4n = C >> 2;
  const int c4 = (int)(e % c4n);
  const long long bj = e / c4n;
  const long long b = bj / N;
  const int j = (int)(bj - b * N);
  const float4 *d4 = reinterpret_cast<const float4 *>(dout), *o4 = reinterpret_cast<const float4 *>(out);
  const int4 *a4 = reinterpret_cast<const int4 *>(arg);
  {
    const float4 d = d4[e], o = o4[e];
    *reinterpret_cast<float4 *>(dV + (size_t)bj * ldg + 4 * c4) = make_float4(d.x * (o.x > 0.f ? 1.0f : slope), d.y * (o.y > 0.f ? 1.0f : slope),
                                                    d.z * (o.z > 0.f ? 1.0f : slope), d.w * (o.w > 0.f ? 1.0f : slope));
  }
  const int2 seg = span[bj];
  const int32_t *cl = col + b * E;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  auto take = [&](long long src, const int4 &a) {
    if (a.x == j || a.y == j || a.z == j || a.w == j) {
      const float4 d = d4[src], o = o4[src];
      if (a.x == j) acc.x += d.x * (o.x > 0.f ? 1.0f : slope);
      if (a.y == j) acc.y += d.y * (o.y > 0.f ? 1.0f : slope);
      if (a.z == j) acc.z += d.z * (o.z > 0.f ? 1.0f : slope);
      if (a.w == j) acc.w += d.w * (o.w > 0.f ? 1.0f : slope);
    }
  };
  int t = seg.x;
  const int end = seg.y;
  for (; t + 4 <= end; t += 4) {  // four listed points at a time: their arg rows are fetched together
    long long src[4];
    int4 a[4];
    for (int u = 0; u < 4; ++u) src[u] = (b * N + cl[t + u]) * c4n + c4;
    for (int u = 0; u < 4; ++u) a[u] = a4[src[u]];
    for (int u = 0; u < 4; ++u) take(src[u], a[u]);
  }
  for (; t < end; ++t) {
    const long long src = (b * N + cl[t]) * c4n + c4;
    take(src, a4[src]);
  }
  *reinterpret_cast<float4 *>(dU + (size_t)bj * ldg + 4 * c4) = acc;
}

}  // namespace hitadv

// ---------------------------------------------------------------------------------------------------
// DGCNN's global pooling (model/dgcnn_cls.py:101-106): LeakyReLU of the embedding layer's pre-activation Z [B,N,C], then
// max and mean over the points, concatenated -> [B,2C].  One pass over Z (the activation tensor itself is never
// written): a block owns 64 channels of one cloud, 16 lanes x float4 across the channels, 16 row groups down the points;
// the row groups are combined in fixed order, so the mean is the same bits every run.  arg = first point attaining the max.
namespace hitadv {

__global__ __launch_bounds__(256) void lrelu_pool_fwd_k(const float *__restrict__ Z, int N, int C, float slope,
                                                        float *__restrict__ out, int32_t *__restrict__ arg) {
  __shared__ float s_sum[16][64], s_best[16][64];
  __shared__ int s_arg[16][64];
  const int lane = threadIdx.x & 15, rg = threadIdx.x >> 4, b = blockIdx.y;
  const int c = blockIdx.x * 64 + lane * 4;
  const float4 *z4 = reinterpret_cast<const float4 *>(Z + (size_t)b * N * C + c);
  const int stride4 = C >> 2;
  float sum[4] = {0.f, 0.f, 0.f, 0.f}, best[4];
  int bi[4] = {0, 0, 0, 0};
  for (int q = 0; q < 4; ++q) best[q] = -__builtin_inff();
  auto take = [&](const float4 &z, int i) {
    const float v[4] = {z.x, z.y, z.z, z.w};
    for (int q = 0; q < 4; ++q) {
      sum[q] += v[q] > 0.f ? v[q] : v[q] * slope;
      if (v[q] > best[q]) { best[q] = v[q]; bi[q] = i; }
    }
  };
  int i = rg;
  for (; i + 112 < N; i += 128) {  // eight rows in flight per lane; consumed in row order, so the sums keep their order
    float4 z[8];
    for (int u = 0; u < 8; ++u) z[u] = z4[(size_t)(i + 16 * u) * stride4];
    for (int u = 0; u < 8; ++u) take(z[u], i + 16 * u);
  }
  for (; i < N; i += 16) take(z4[(size_t)i * stride4], i);
  for (int q = 0; q < 4; ++q) {
    s_sum[rg][lane * 4 + q] = sum[q];
    s_best[rg][lane * 4 + q] = best[q];
    s_arg[rg][lane * 4 + q] = bi[q];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int t = threadIdx.x;
    float tot = 0.f, bst = -__builtin_inff();
    int ba = 0;
    for (int g = 0; g < 16; ++g) {
      tot += s_sum[g][t];
      const float v = s_best[g][t];
      if (v > bst || (v == bst && s_arg[g][t] < ba)) { bst = v; ba = s_arg[g][t]; }
    }
    const int cc = blockIdx.x * 64 + t;
    out[(size_t)b * 2 * C + cc] = bst > 0.f ? bst : bst * slope;
    out[(size_t)b * 2 * C + C + cc] = tot / (float)N;
    arg[(size_t)b * C + cc] = ba;
  }
}

// dZ[b,i,c] = lrelu'(Z[b,i,c]) * ( g[b,C+c] / N  +  (i == arg[b,c]) * g[b,c] )
__global__ __launch_bounds__(256) void lrelu_pool_bwd_k(const float *__restrict__ Z, const float *__restrict__ g,
                                                        const int32_t *__restrict__ arg, int N, int C, float slope,
                                                        float *__restrict__ dZ, long long total4) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;  // (b, i, c4)
  if (e >= total4) return;
  const int c4n = C >> 2;
  const int c4 = (int)(e % c4n);
  const long long bi = e / c4n;
  const long long b = bi / N;
  const int i = (int)(bi - b * N);
  const float4 z = reinterpret_cast<const float4 *>(Z)[e];
  const float4 gmax = *reinterpret_cast<const float4 *>(g + b * 2 * C + 4 * c4);
  const float4 gmean = *reinterpret_cast<const float4 *>(g + b * 2 * C + C + 4 * c4);
  const int4 a = *reinterpret_cast<const int4 *>(arg + b * C + 4 * c4);
  const float inv = 1.0f / (float)N;
  float4 d;
  d.x = (gmean.x * inv + (a.x == i ? gmax.x : 0.f)) * (z.x > 0.f ? 1.0f : slope);
  d.y = (gmean.y * inv + (a.y == i ? gmax.y : 0.f)) * (z.y > 0.f ? 1.0f : slope);
  d.z = (gmean.z * inv + (a.z == i ? gmax.z : 0.f)) * (z.z > 0.f ? 1.0f : slope);
  d.w = (gmean.w * inv + (a.w == i ? gmax.w : 0.f)) * (z.w > 0.f ? 1.0f : slope);
  reinterpret_cast<float4 *>(dZ)[e] = d;
}

}  // namespace hitadv

extern "C" int hitadv_lrelu_pool_fwd(const float *Z, int B, int N, int C, float slope, float *out, int32_t *arg,
                                     void *stream) {
  if (!Z || !out || !arg || B <= 0 || N <= 0 || C <= 0 || (C & 63) || ((uintptr_t)Z & 15)) return HITADV_E_ARG;
  hitadv::lrelu_pool_fwd_k<<<dim3(C / 64, B), 256, 0, (hipStream_t)stream>>>(Z, N, C, slope, out, arg);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_lrelu_pool_bwd(const float *Z, const float *g, const int32_t *arg, int B, int N, int C, float slope,
                                     float *dZ, void *stream) {
  if (!Z || !g || !arg || !dZ || B <= 0 || N <= 0 || C <= 0 || (C & 63) ||
      (((uintptr_t)Z | (uintptr_t)dZ | (uintptr_t)g | (uintptr_t)arg) & 15))
    return HITADV_E_ARG;
  const long long total4 = (long long)B * N * (C >> 2);
  hitadv::lrelu_pool_bwd_k<<<(unsigned)((total4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(Z, g, arg, N, C, slope, dZ,
                                                                                          total4);
  HITADV_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// First shared layer of a "sample and group" block (PointNet++ set abstraction, pointnet2_utils.py:161-205; PCT
// Local_op, pct_utils.py:98-140) without the grouped tensor.  The reference gathers [B,S,ns,Cin] neighbour features,
// subtracts the centre, concatenates and runs a 1x1 convolution over S*ns rows; the convolution is linear, so
//   W [x_j - c_i ; c_i] + t  =  Wa x_j + ((Wb - Wa) c_i + t)  =  U[j] + V[i]
// with U = one GEMM over the N points and V = one over the S centres (ns times fewer rows, no gather, no concat):
//   H[b,i,s,:] = relu(U[b, idx[b,i,s], :] + V[b,i,:]).
// Backward: the ReLU mask is recomputed from U + V (H is not read), dV[i] = sum_s dH[i,s]*mask, and dU[j] GATHERS its
// contributions over the reversed neighbour table (reverse_graph_k) in ascending (i, s): no atomics, same bits each run.
namespace hitadv {

__global__ __launch_bounds__(256) void group_add_relu_fwd_k(const float *__restrict__ U, const float *__restrict__ V,
                                                            const int64_t *__restrict__ idx, int N, int S, int ns, int C,
                                                            float *__restrict__ H, long long total4, int xcd_order,
                                                            int blocks_per_cloud) {
  // Workgroup ids go round the eight XCDs (id mod 8), each with an L2 of its own: in launch order every cloud's blocks land
  // on all eight and U[b] is fetched eight times (PMC, cfg4 shape: 190 MB of reads for 50 MB of operands).  xcd_order 1: the
  // ids of one XCD take a contiguous eighth of the work; 2: XCD x takes the clouds b = x mod 8 (blocks_per_cloud > 0), so
  // that the eight XCDs write into eight CONSECUTIVE clouds at any time.  0: launch order -- the one shipped: both XCD-local
  // orders bring the traffic to 1.00x algorithmic and the kernel from 94.9 / 74.4 us to 101-102 / 96-101 us (cfg4 / cfg5 shape;
  // docs/kernels/round4.md section 6): it is bound by its writes, which launch order spreads over adjacent addresses.
  long long blk = blockIdx.x;
  if (xcd_order == 1 && (gridDim.x & 7u) == 0u) {
    blk = (long long)(blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  } else if (xcd_order == 2 && blocks_per_cloud > 0) {
    const long long q = blockIdx.x >> 3, x = blockIdx.x & 7u;  // q-th block of XCD x
    blk = (q % blocks_per_cloud) + (long long)blocks_per_cloud * (8 * (q / blocks_per_cloud) + x);
  }
  const long long e = blk * 256 + threadIdx.x;  // (b, i, s, c4)
  if (e >= total4) return;
  const int c4n = C >> 2;
  const int c4 = (int)(e % c4n);
  const long long bis = e / c4n;  // (b * S + i) * ns + s
  const long long bi = bis / ns;
  const long long b = bi / S;
  const long long j = idx[bis];
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (j >= 0 && j < N) {
    const float4 u = *reinterpret_cast<const float4 *>(U + ((size_t)(b * N + j)) * C + 4 * c4);
    const float4 v = *reinterpret_cast<const float4 *>(V + (size_t)bi * C + 4 * c4);
    o = make_float4(fmaxf(u.x + v.x, 0.f), fmaxf(u.y + v.y, 0.f), fmaxf(u.z + v.z, 0.f), fmaxf(u.w + v.w, 0.f));
  }
  reinterpret_cast<float4 *>(H)[e] = o;
}

// dV[b,i,c] = sum_s dH[b,i,s,c] * [U[b,idx[b,i,s],c] + V[b,i,c] > 0]   (one lane per (b, i, 4 channels), s ascending)
__global__ __launch_bounds__(256) void group_add_relu_dv_k(const float *__restrict__ dH, const float *__restrict__ U,
                                                           const float *__restrict__ V, const int64_t *__restrict__ idx,
                                                           int N, int S, int ns, int C, float *__restrict__ dV,
                                                           long long total4) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;  // (b, i, c4)
  if (e >= total4) return;
  const int c4n = C >> 2;
  const int c4 = (int)(e % c4n);
  const long long bi = e / c4n;
  const long long b = bi / S;
  const float4 v = *reinterpret_cast<const float4 *>(V + (size_t)bi * C + 4 * c4);
  const int64_t *nb = idx + bi * ns;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int t = 0; t < ns; ++t) {
    const long long j = nb[t];
    if (j < 0 || j >= N) continue;
    const float4 u = *reinterpret_cast<const float4 *>(U + ((size_t)(b * N + j)) * C + 4 * c4);
    const float4 d = *reinterpret_cast<const float4 *>(dH + ((size_t)(bi * ns + t)) * C + 4 * c4);
    acc.x += u.x + v.x > 0.f ? d.x : 0.f;
    acc.y += u.y + v.y > 0.f ? d.y : 0.f;
    acc.z += u.z + v.z > 0.f ? d.z : 0.f;
    acc.w += u.w + v.w > 0.f ? d.w : 0.f;
  }
  *reinterpret_cast<float4 *>(dV + (size_t)bi * C + 4 * c4) = acc;
}

// dU[b,j,c] = sum over the lists i that contain j, over the slots s with idx[b,i,s] == j (a ball query pads a list by
// repeating an entry), of dH[b,i,s,c] * mask.  G lanes (G = C/4 rounded up to a power of two, at most 64) serve one
// target point: they read list i G entries at a time, ballot the matching slots, and each lane (4 channels) adds the rows
// in ascending (i, s).  A wave therefore holds 64/G targets (4 at C = 64).
template <int G>
__global__ __launch_bounds__(64) void group_add_relu_du_k(const float *__restrict__ dH, const float *__restrict__ U,
                                                          const float *__restrict__ V, const int64_t *__restrict__ idx,
                                                          const int2 *__restrict__ span, const int32_t *__restrict__ col,
                                                          int N, int S, int ns, int C, long long targets,
                                                          float *__restrict__ dU) {
  constexpr int TPW = 64 / G;  // targets per wave
  long long blk = blockIdx.x;  // the blocks of one cloud on one XCD (its dH rows are read once, U / V / idx from L2)
  if ((gridDim.x & 7) == 0) blk = (blk & 7) * (gridDim.x >> 3) + (blk >> 3);
  const int lane = threadIdx.x, grp = lane / G, gl = lane % G;
  const long long tgt = blk * TPW + grp;  // (b, j)
  if (tgt >= targets) return;
  const long long b = tgt / N;
  const int j = (int)(tgt - b * N);
  const int2 seg = span[tgt];
  const int32_t *cl = col + b * (long long)S * ns;
  const int c4n = C >> 2;
  const unsigned long long gmask = G == 64 ? ~0ull : ((1ull << G) - 1ull);
  for (int base = 0; base < c4n; base += G) {  // C <= 256: one trip; every lane of the group stays in for the ballots
    const int c4 = base + gl;
    const bool live = c4 < c4n;
    const int cc = live ? c4 : 0;
    const float4 u = *reinterpret_cast<const float4 *>(U + ((size_t)(b * N + j)) * C + 4 * cc);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // Four lists at a time: a list costs a chain of three dependent loads (its number, its entries, the gradient row of the
    // matching slot), and the kernel is bound by those latencies.  The chains of four lists run side by side -- the list
    // numbers, then their V rows and entries, then the rows of their FIRST matching slots -- while the additions keep their
    // order (ascending list, ascending slot; the further slots of a list -- a ball query pads by repeating an entry -- are
    // fetched where they are added).
    constexpr int LB = 4, NCH = 64 / G;  // ns <= 64: at most 64 / G chunks of G entries per list
    for (int t = seg.x; t < seg.y; t += LB) {
      int li[LB];
#pragma unroll
      for (int q = 0; q < LB; ++q) li[q] = t + q < seg.y ? cl[t + q] : -1;
      long long bi[LB];
      float4 v[LB];
      long long mine[LB][NCH];
#pragma unroll
      for (int q = 0; q < LB; ++q) {
        bi[q] = b * S + max(li[q], 0);
        v[q] = *reinterpret_cast<const float4 *>(V + (size_t)bi[q] * C + 4 * cc);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
          mine[q][ch] = (li[q] >= 0 && ch * G + gl < ns) ? idx[bi[q] * ns + ch * G + gl] : -1;  // -1 matches no target
      }
      unsigned long long hit[LB][NCH];
      int first[LB];
      float4 d0[LB];
#pragma unroll
      for (int q = 0; q < LB; ++q) {
        first[q] = -1;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
          hit[q][ch] = (__ballot(mine[q][ch] == j) >> (grp * G)) & gmask;  // (every lane of the group is here)
          if (first[q] < 0 && hit[q][ch]) {
            first[q] = ch * G + __builtin_ctzll(hit[q][ch]);
            hit[q][ch] &= hit[q][ch] - 1;
          }
        }
        d0[q] = *reinterpret_cast<const float4 *>(dH + ((size_t)(bi[q] * ns + max(first[q], 0))) * C + 4 * cc);
      }
#pragma unroll
      for (int q = 0; q < LB; ++q) {
        if (first[q] < 0) continue;  // group-uniform
        const bool mx = u.x + v[q].x > 0.f, my = u.y + v[q].y > 0.f, mz = u.z + v[q].z > 0.f, mw = u.w + v[q].w > 0.f;
        acc.x += mx ? d0[q].x : 0.f;
        acc.y += my ? d0[q].y : 0.f;
        acc.z += mz ? d0[q].z : 0.f;
        acc.w += mw ? d0[q].w : 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
          unsigned long long rest = hit[q][ch];
          while (rest) {
            const int sl = ch * G + __builtin_ctzll(rest);
            rest &= rest - 1;
            const float4 d = *reinterpret_cast<const float4 *>(dH + ((size_t)(bi[q] * ns + sl)) * C + 4 * cc);
            acc.x += mx ? d.x : 0.f;
            acc.y += my ? d.y : 0.f;
            acc.z += mz ? d.z : 0.f;
            acc.w += mw ? d.w : 0.f;
          }
        }
      }
    }
    if (live) *reinterpret_cast<float4 *>(dU + ((size_t)(b * N + j)) * C + 4 * c4) = acc;
  }
}

}  // namespace hitadv

extern "C" int hitadv_group_add_relu_fwd(const float *U, const float *V, const int64_t *idx, int B, int N, int S, int ns,
                                         int C, float *H, void *stream) {
  if (!U || !V || !idx || !H || B <= 0 || N <= 0 || S <= 0 || ns <= 0 || C <= 0 || (C & 3) ||
      (((uintptr_t)U | (uintptr_t)V | (uintptr_t)H) & 15))
    return HITADV_E_ARG;
  const long long total4 = (long long)B * S * ns * (C >> 2);
  static const int want = [] { const char *e = getenv("HITADV_GAR_XCD"); return e ? atoi(e) : 0; }();  // tuning knob, see the kernel
  const long long per_cloud4 = (long long)S * ns * (C >> 2);
  int order = want, bpc = 0;
  if (order == 2) {
    if ((per_cloud4 & 255) == 0 && (B & 7) == 0) bpc = (int)(per_cloud4 >> 8);
    else order = 0;
  }
  hitadv::group_add_relu_fwd_k<<<(unsigned)((total4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(U, V, idx, N, S, ns, C, H,
                                                                                              total4, order, bpc);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t hitadv_group_add_relu_bwd_scratch_ints(int B, int N, int S, int ns) {
  if (B <= 0 || N <= 0 || S <= 0 || ns <= 0) return HITADV_E_ARG;
  return (int64_t)B * (2 * (int64_t)N + (int64_t)S * ns);
}

extern "C" int hitadv_group_add_relu_bwd(const float *dH, const float *U, const float *V, const int64_t *idx, int B, int N,
                                         int S, int ns, int C, float *dU, float *dV, int32_t *scratch, void *stream) {
  if (!dH || !U || !V || !idx || !dU || !dV || !scratch || B <= 0 || N <= 0 || S <= 0 || ns <= 0 || ns > 64 || C <= 0 ||
      (C & 3) || (long long)S * ns > 0x7fffffffLL || (long long)B * N > 0x7fffffffLL ||
      (((uintptr_t)dH | (uintptr_t)U | (uintptr_t)V | (uintptr_t)dU | (uintptr_t)dV) & 15) || ((uintptr_t)scratch & 7))
    return HITADV_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long long tv = (long long)B * S * (C >> 2);
  hitadv::group_add_relu_dv_k<<<(unsigned)((tv + 255) / 256), 256, 0, s>>>(dH, U, V, idx, N, S, ns, C, dV, tv);
  HITADV_LAUNCH_CHECK();
  const int W = (S + 31) / 32;
  if (W + 1 > RG_BITMAP_WORDS) return HITADV_E_ARG;
  const int J = min(RG_THREADS, RG_BITMAP_WORDS / (W + 1));
  int2 *span = reinterpret_cast<int2 *>(scratch);
  int32_t *col = scratch + (size_t)B * 2 * N;
  const size_t shm = (size_t)(RG_BITMAP_WORDS + 3 * RG_THREADS) * 4;
  HITADV_RAISE_LDS((&hitadv::reverse_graph_k), (int)shm);
  hitadv::reverse_graph_k<<<dim3((N + J - 1) / J, B), RG_THREADS, shm, s>>>(idx, S, N, ns, W, J, span, col);
  HITADV_LAUNCH_CHECK();
  const long long targets = (long long)B * N;
  const int c4n = C >> 2;
  if (c4n <= 16)
    hitadv::group_add_relu_du_k<16><<<(unsigned)((targets + 3) / 4), 64, 0, s>>>(dH, U, V, idx, span, col, N, S, ns, C, targets, dU);
  else if (c4n <= 32)
    hitadv::group_add_relu_du_k<32><<<(unsigned)((targets + 1) / 2), 64, 0, s>>>(dH, U, V, idx, span, col, N, S, ns, C, targets, dU);
  else
    hitadv::group_add_relu_du_k<64><<<(unsigned)targets, 64, 0, s>>>(dH, U, V, idx, span, col, N, S, ns, C, targets, dU);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_edge_max_fwd(const float *U, const float *V, int ld, const int64_t *idx, int B, int N, int C, int k,
                                   float slope, float *out, int32_t *arg, void *stream) {
  if (!U || !V || !idx || !out || !arg || B <= 0 || N <= 0 || C <= 0 || (C & 3) || k <= 0 || ld < C || (ld & 3) ||
      (((uintptr_t)U | (uintptr_t)V | (uintptr_t)out | (uintptr_t)arg) & 15))
    return HITADV_E_ARG;
  const long long total4 = (long long)B * N * (C >> 2);
  hitadv::edge_max_fwd_k<<<(unsigned)((total4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(U, V, ld, idx, N, C, k, slope,
                                                                                         out, arg, total4);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t hitadv_edge_max_bwd_scratch_ints(int B, int N, int k) {
  if (B <= 0 || N <= 0 || k <= 0) return HITADV_E_ARG;
  return (int64_t)B * (2 * (int64_t)N + (int64_t)N * k);
}

extern "C" int hitadv_edge_max_bwd(const float *dout, const float *out, const int32_t *arg, const int64_t *idx, int B,
                                   int N, int C, int k, float slope, float *dU, float *dV, int ldg, int32_t *scratch,
                                   void *stream) {
  if (!dout || !out || !arg || !idx || !dU || !dV || !scratch || B <= 0 || N <= 0 || C <= 0 || (C & 3) || k <= 0 ||
      ldg < C || (ldg & 3) ||
      (long long)N * k > 0x7fffffffLL ||
      (((uintptr_t)dout | (uintptr_t)out | (uintptr_t)arg | (uintptr_t)dU | (uintptr_t)dV) & 15) ||
      ((uintptr_t)scratch & 7))
    return HITADV_E_ARG;
  const int W = (N + 31) / 32;
  if (W + 1 > RG_BITMAP_WORDS) return HITADV_E_ARG;  // N > 491,488: one bit row no longer fits
  const int J = min(RG_THREADS, RG_BITMAP_WORDS / (W + 1));
  int2 *span = reinterpret_cast<int2 *>(scratch);
  int32_t *col = scratch + (size_t)B * 2 * N;
  const size_t shm = (size_t)(RG_BITMAP_WORDS + 3 * RG_THREADS) * 4;
  HITADV_RAISE_LDS((&hitadv::reverse_graph_k), (int)shm);
  hitadv::reverse_graph_k<<<dim3((N + J - 1) / J, B), RG_THREADS, shm, (hipStream_t)stream>>>(idx, N, N, k, W, J, span,
                                                                                          col);
  HITADV_LAUNCH_CHECK();
  const long long total4 = (long long)B * N * (C >> 2);
  hitadv::edge_max_bwd_k<<<(unsigned)((total4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      dout, out, arg, span, col, N, C, N * k, slope, dU, dV, ldg, total4);
  HITADV_LAUNCH_CHECK();
  return 0;
}
