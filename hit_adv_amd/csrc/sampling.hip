// Farthest point sampling (two semantics) and gather_points (+ deterministic grad).
//
//   K5  fps<PT, EXT>   one workgroup per cloud, points and running distances in VGPRs
//                      (PT points per lane), cloud mirrored in LDS as float4 for the winner's
//                      coordinates.  Per step: PT distance updates, a 64-bit (distance-bits,
//                      tie-key) max via two DPP reductions, one barrier, 4-slot merge.  Latency-bound
//                      serial chain of m steps.
//        MODE 0 : ShapeAttack/HiT_ADV.py:489-510 = model/pointnet2_utils.py:63-84 semantics (given start, direct-form
//                 squared distances, running distance 1e10, lowest index on ties)
//        MODE 1 : sampling_gpu.cu:69-173 semantics (start 0, |p|^2 <= 1e-3 skipped, the
//                 thread-slot tie order of the reference's shared-memory tree)
//        MODE 2 : PCT's sampler, util/other_utils.py:254-272: given start, distances by get_dists (:237-251: sqrt of
//                 the clamped Gram form in torch's own fp32 arithmetic, common.hpp::pct_dist), running distance 1e5
#include "common.hpp"
#include "hitadv.h"

namespace hitadv {

__device__ __forceinline__ uint32_t bitrev_n(uint32_t v, int bits) { return __brev(v) >> (32 - bits); }

template <int PT, int MODE, typename IdxT>
__global__ __launch_bounds__(256) void fps(const float *__restrict__ xyz, const int64_t *__restrict__ start,
                                           int N, int m, int ref_bs, int ref_bits, int use_lds,
                                           IdxT *__restrict__ idx) {
  extern __shared__ float4 spts[];  // N entries when use_lds
  __shared__ unsigned long long slot[2][4];
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  xyz += (size_t)b * N * 3;
  idx += (size_t)b * m;
  constexpr bool EXT = MODE == 1, PCT = MODE == 2;
  float px[PT], py[PT], pz[PT], run[PT], rp[PT];
  uint32_t tb[PT];
#pragma unroll
  for (int u = 0; u < PT; ++u) {
    const int k = threadIdx.x + 256 * u;
    const bool in = k < N;
    const int kk = in ? k : 0;
    px[u] = xyz[kk * 3];
    py[u] = xyz[kk * 3 + 1];
    pz[u] = xyz[kk * 3 + 2];
    run[u] = PCT ? 1e5f : 1e10f;
    rp[u] = (px[u] * px[u] + py[u] * py[u]) + pz[u] * pz[u];
    bool ok = in;
    uint32_t tie = (uint32_t)k;
    if (EXT) {
      const float mag = (px[u] * px[u] + py[u] * py[u]) + pz[u] * pz[u];
      ok = ok && !((double)mag <= 1e-3);
      const uint32_t s = (uint32_t)k & (uint32_t)(ref_bs - 1);
      tie = (ref_bits ? (bitrev_n(s, ref_bits) << 16) : 0u) | ((uint32_t)k >> ref_bits);
    }
    tb[u] = ok ? 0xFFFFFFFFu - tie : 0u;  // 0 marks "never a candidate"
    if (use_lds && in) spts[k] = make_float4(px[u], py[u], pz[u], rp[u]);
  }
  int far = EXT ? 0 : (int)start[b];
  __syncthreads();
  const int steps = EXT ? m - 1 : m;
  if (EXT && threadIdx.x == 0) idx[0] = 0;
  for (int j = 0; j < steps; ++j) {
    if (!EXT && threadIdx.x == 0) idx[j] = (IdxT)far;
    float cx, cy, cz, rc;
    if (use_lds) {
      const float4 c = spts[far];
      cx = c.x; cy = c.y; cz = c.z; rc = c.w;
    } else {
      cx = xyz[far * 3]; cy = xyz[far * 3 + 1]; cz = xyz[far * 3 + 2];
      rc = (cx * cx + cy * cy) + cz * cz;
    }
    unsigned long long best = 0ull;
#pragma unroll
    for (int u = 0; u < PT; ++u) {
      const float d = PCT ? pct_dist(cx, cy, cz, rc, px[u], py[u], pz[u], rp[u]) : sqdist3(px[u], py[u], pz[u], cx, cy, cz);
      if (EXT) {
        if (tb[u] != 0u) run[u] = d < run[u] ? d : run[u];
      } else {
        run[u] = d < run[u] ? d : run[u];
      }
      const unsigned long long key = tb[u] ? (((unsigned long long)fbits(run[u]) << 32) | tb[u]) : 0ull;
      best = key > best ? key : best;
    }
    best = wave_max_u64_dpp(best);
    if (lane == 0) slot[j & 1][wave] = best;
    __syncthreads();
    unsigned long long w = slot[j & 1][0];
#pragma unroll
    for (int t = 1; t < 4; ++t) {
      const unsigned long long o = slot[j & 1][t];
      w = o > w ? o : w;
    }
    if (w == 0ull) {
      far = 0;  // no candidate anywhere: the reference's (best=-1, besti=0) fall-through
    } else {
      const uint32_t tie = 0xFFFFFFFFu - (uint32_t)(w & 0xffffffffu);
      if (EXT) {
        const uint32_t s = ref_bits ? bitrev_n(tie >> 16, ref_bits) : 0u;
        far = (int)(((tie & 0xffffu) << ref_bits) | s);
      } else {
        far = (int)tie;
      }
    }
    if (EXT && threadIdx.x == 0) idx[j + 1] = (IdxT)far;
  }
}

__global__ __launch_bounds__(256) void gather_points_k(int c, int n, int npoints,
                                                       const float *__restrict__ points,
                                                       const int32_t *__restrict__ idx,
                                                       float *__restrict__ out, long long total) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int j = (int)(e % npoints);
  const long long bc = e / npoints;
  const int b = (int)(bc / c);
  out[e] = points[bc * n + idx[(size_t)b * npoints + j]];
}

// grad_points[b,:,i] = sum_{j: idx[b,j]==i} grad_out[b,:,j]; lane (b,i) is the only writer of column i.
__global__ __launch_bounds__(256) void gather_points_grad_k(int c, int n, int npoints,
                                                            const float *__restrict__ grad_out,
                                                            const int32_t *__restrict__ idx,
                                                            float *__restrict__ grad_points) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float *gp = grad_points + (size_t)b * c * n + i;
  for (int l = 0; l < c; ++l) gp[(size_t)l * n] = 0.f;
  const int32_t *ip = idx + (size_t)b * npoints;
  const float *go = grad_out + (size_t)b * c * npoints;
  for (int j = 0; j < npoints; ++j)
    if (ip[j] == i)
      for (int l = 0; l < c; ++l) gp[(size_t)l * n] += go[(size_t)l * npoints + j];
}

template <int MODE, typename IdxT>
static int launch_fps(const float *xyz, const int64_t *start, int B, int N, int m, IdxT *idx, hipStream_t s) {
  int ref_bs = 1, ref_bits = 0;
  if (MODE == 1) {  // include/cuda_utils.h:15-18 opt_n_threads: clamp(2^floor(log2 n), 1, 512)
    while (ref_bs * 2 <= N && ref_bs < 512) {
      ref_bs *= 2;
      ++ref_bits;
    }
  }
  const int use_lds = N <= 4096;
  const size_t shm = use_lds ? (size_t)N * sizeof(float4) : 0;
#define HITADV_FPS_CASE(PT)                                                                       \
  if (N <= 256 * PT) {                                                                            \
    fps<PT, MODE, IdxT><<<B, 256, shm, s>>>(xyz, start, N, m, ref_bs, ref_bits, use_lds, idx);     \
    return 0;                                                                                     \
  }
  HITADV_FPS_CASE(1)
  HITADV_FPS_CASE(2)
  HITADV_FPS_CASE(4)
  HITADV_FPS_CASE(8)
  HITADV_FPS_CASE(16)
  HITADV_FPS_CASE(32)
  HITADV_FPS_CASE(64)
#undef HITADV_FPS_CASE
  return HITADV_E_ARG;  // N > 16384 not supported
}

}  // namespace hitadv

using namespace hitadv;

extern "C" int hitadv_fps_from_start(const float *xyz, const int64_t *start, int B, int N, int m,
                                     int64_t *idx, void *stream) {
  if (!xyz || !start || !idx || B <= 0 || N <= 0 || m <= 0) return HITADV_E_ARG;
  int rc = launch_fps<0, int64_t>(xyz, start, B, N, m, idx, (hipStream_t)stream);
  if (rc) return rc;
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_fps_pct(const float *xyz, const int64_t *start, int B, int N, int m, int64_t *idx, void *stream) {
  if (!xyz || !start || !idx || B <= 0 || N <= 0 || m <= 0) return HITADV_E_ARG;
  int rc = launch_fps<2, int64_t>(xyz, start, B, N, m, idx, (hipStream_t)stream);
  if (rc) return rc;
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp,
                                              int32_t *idxs, void *stream) {
  (void)temp;  // running distances live in registers; the scratch tensor of the reference is unused
  if (!dataset || !idxs || b <= 0 || n <= 0) return HITADV_E_ARG;
  if (m <= 0) return 0;  // sampling_gpu.cu:73
  int rc = launch_fps<1, int32_t>(dataset, nullptr, b, n, m, idxs, (hipStream_t)stream);
  if (rc) return rc;
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_gather_points(int b, int c, int n, int npoints, const float *points, const int32_t *idx,
                                    float *out, void *stream) {
  if (!points || !idx || !out || b <= 0 || c <= 0 || n <= 0 || npoints <= 0) return HITADV_E_ARG;
  const long long total = (long long)b * c * npoints;
  gather_points_k<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(c, n, npoints, points, idx,
                                                                                    out, total);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                                         const int32_t *idx, float *grad_points, void *stream) {
  if (!grad_out || !idx || !grad_points || b <= 0 || c <= 0 || n <= 0 || npoints <= 0) return HITADV_E_ARG;
  dim3 grid((n + 255) / 256, b);
  gather_points_grad_k<<<grid, 256, 0, (hipStream_t)stream>>>(c, n, npoints, grad_out, idx, grad_points);
  HITADV_LAUNCH_CHECK();
  return 0;
}
