// Device-side bodies of the regularisers' forward pass (regulariser.hip), shared with the merged launch of iteration.hip.
#pragma once
#include "common.hpp"

namespace hitadv {

constexpr int RG_NPART = 16;  // floats per cloud: sp2, s1ms2, dot, rr, nn, d00..d22
constexpr int RG_NSCAL = 8;   // scalars: coef, inv_np, inv_ns, dist, scaled, -, -, -

__device__ __forceinline__ float block_sum(float v, float *sm) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__device__ inline void reg_finalise_body(const float *part, const float *__restrict__ scale_const, int B, int C, float cd_w,
                                  float ker_w, float hide_w, float *__restrict__ per_cloud, float *__restrict__ scal,
                                  float *__restrict__ dist_out, float *__restrict__ scaled_out, float *sm);

// `fin` != nullptr: the last block to arrive (ticket = scal[7], zeroed once by the caller) combines the clouds in the same
// launch -- what reg_finalise does as a second launch otherwise.
struct RegFin {
  const float *scale_const;
  int B;
  float cd_w, ker_w, hide_w;
  float *per_cloud, *scal, *dist_out, *scaled_out;
};

// The per-cloud pass of one block (b = the cloud); called from reg_partials (regulariser.hip) and from the merged launch of
// iteration.hip.
__device__ __forceinline__ void reg_partials_body(const float *__restrict__ P, const float *__restrict__ sigma,
                                                  const float *__restrict__ adv, const float *__restrict__ ori,
                                                  const float *__restrict__ hide_ref, int N, int C, float min_s,
                                                  float inv_range, float *part, RegFin fin, int fused, const int b) {
  __shared__ float sm[4];
  __shared__ float sw[4][14];
  __shared__ int s_last;
  // Loads are requested in batches ahead of their use (up to four strided slots per thread at a time; a slot past the end
  // reads a clamped address and enters the sums multiplied by an exact 0): a load inside a loop of unknown trip count is a
  // global round trip per trip.  The sums themselves keep the order of the plain strided loops.
  float a = 0.f;
  for (int e0 = 0; e0 < C * 3; e0 += 1024) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = P[(size_t)b * C * 3 + min(e0 + (int)threadIdx.x + 256 * u, C * 3 - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) a = fmaf(v[u] * (e0 + (int)threadIdx.x + 256 * u < C * 3 ? 1.0f : 0.0f), v[u], a);
  }
  float s2 = 0.f, dot = 0.f, rr = 0.f, nn = 0.f;
  for (int e0 = 0; e0 < C; e0 += 1024) {
    float sv[4], rv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t o = (size_t)b * C + min(e0 + (int)threadIdx.x + 256 * u, C - 1);
      sv[u] = sigma[o];
      rv[u] = hide_ref[o];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float w = e0 + (int)threadIdx.x + 256 * u < C ? 1.0f : 0.0f;
      const float t = (1.0f - sv[u]) * w;
      s2 = fmaf(t, t, s2);
      const float n = ((sv[u] - min_s) * inv_range) * w;
      const float r = rv[u] * w;
      dot = fmaf(r, n, dot);
      rr = fmaf(r, r, rr);
      nn = fmaf(n, n, nn);
    }
  }
  float d[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) d[k] = 0.f;
  const float *ap = adv + (size_t)b * 3 * N, *op = ori + (size_t)b * 3 * N;
  for (int n0 = 0; n0 < N; n0 += 1024) {
    float av[4][3], ov[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int n = min(n0 + (int)threadIdx.x + 256 * u, N - 1);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        av[u][c] = ap[c * N + n];
        ov[u][c] = op[c * N + n];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float w = n0 + (int)threadIdx.x + 256 * u < N ? 1.0f : 0.0f;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const float t = (av[u][i] - ov[u][j]) * w;
          d[3 * i + j] = fmaf(t, t, d[3 * i + j]);
        }
    }
  }
  // the fourteen block sums in one pass: a wave sum each, one barrier, then the four waves in block_sum's order
  float out[14] = {a, s2, dot, rr, nn, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8]};
  {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 14; ++k) {
      const float v = wave_sum(out[k]);
      if (lane == 0) sw[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 14) {
      const int k = threadIdx.x;
      __hip_atomic_store(&part[(size_t)b * RG_NPART + k], (sw[0][k] + sw[1][k]) + (sw[2][k] + sw[3][k]), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (!fused) return;
  int *ticket = reinterpret_cast<int *>(fin.scal + 7);
  if (!handoff_last_arriver(ticket, 0, fin.B, &s_last)) return;
  reg_finalise_body(part, fin.scale_const, fin.B, C, fin.cd_w, fin.ker_w, fin.hide_w, fin.per_cloud, fin.scal, fin.dist_out,
                    fin.scaled_out, sm);
  if (threadIdx.x == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One block: combine the clouds.  per_cloud[b] = {cos_b, inv(|r||n|), inv(|n|^2), arg0, arg1, arg2}
__device__ inline void reg_finalise_body(const float *part, const float *__restrict__ scale_const, int B, int C, float cd_w,
                                  float ker_w, float hide_w, float *__restrict__ per_cloud, float *__restrict__ scal,
                                  float *__restrict__ dist_out, float *__restrict__ scaled_out, float *sm) {
  float sp = 0.f, ss = 0.f, sq = 0.f, sc = 0.f, sk = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    float p[14];
#pragma unroll
    for (int k = 0; k < 14; ++k)  // written by other blocks of this launch in the fused form: read past the L1
      p[k] = __hip_atomic_load(&part[(size_t)b * RG_NPART + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sp += p[0];
    ss += p[1];
    const float nr = fmaxf(__builtin_sqrtf(p[3]), 1e-8f), nn = fmaxf(__builtin_sqrtf(p[4]), 1e-8f);
    const float cosv = p[2] / (nr * nn);
    sc += cosv;
    float q = 0.f;
    float *pc = per_cloud + (size_t)b * 8;
#pragma unroll
    for (int i = 0; i < 3; ++i) {  // preds = rows of adv; nearest ori row, lowest index on ties
      float best = p[5 + 3 * i];
      int arg = 0;
      if (p[5 + 3 * i + 1] < best) { best = p[5 + 3 * i + 1]; arg = 1; }
      if (p[5 + 3 * i + 2] < best) { best = p[5 + 3 * i + 2]; arg = 2; }
      q += best;
      pc[3 + i] = (float)arg;
    }
    sq += q / 3.0f;
    pc[0] = cosv;
    pc[1] = 1.0f / (nr * nn);
    pc[2] = 1.0f / (nn * nn);
    sk += scale_const[b];
  }
  {  // the five block sums in one pass (block_sum's order: a wave sum each, then (w0 + w1) + (w2 + w3))
    __shared__ float sw5[4][5];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float v5[5] = {sp, ss, sq, sc, sk};
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const float v = wave_sum(v5[k]);
      if (lane == 0) sw5[wave][k] = v;
    }
    __syncthreads();
    sp = (sw5[0][0] + sw5[1][0]) + (sw5[2][0] + sw5[3][0]);
    ss = (sw5[0][1] + sw5[1][1]) + (sw5[2][1] + sw5[3][1]);
    sq = (sw5[0][2] + sw5[1][2]) + (sw5[2][2] + sw5[3][2]);
    sc = (sw5[0][3] + sw5[1][3]) + (sw5[2][3] + sw5[3][3]);
    sk = (sw5[0][4] + sw5[1][4]) + (sw5[2][4] + sw5[3][4]);
  }
  if (threadIdx.x == 0) {
    const float np = __builtin_sqrtf(sp), ns = __builtin_sqrtf(ss);
    float dist = 0.f;
    if (cd_w != 0.f) dist += cd_w * (sq / (float)B);
    if (ker_w != 0.f) dist += ker_w * ((np + ns) / (float)C);
    if (hide_w != 0.f) dist += hide_w * (sc / (float)B);
    const float coef = sk / (float)B;
    scal[0] = coef;
    scal[1] = np > 0.f ? 1.0f / np : 0.f;
    scal[2] = ns > 0.f ? 1.0f / ns : 0.f;
    scal[3] = dist;
    scal[4] = coef * dist;
    *dist_out = dist;
    *scaled_out = coef * dist;
  }
}

// The backward pass's three terms (reg_backward, regulariser.hip), for the kernels that evaluate them on the fly instead
// of reading them from a launch of their own: the same expressions in the same order, so the same bits.
struct RegGrad {
  const float *per_cloud;  // [B,8] of the forward pass; nullptr = no regulariser
  const float *scal;       // [8]
  const float *hide_ref;   // [B,C]
  float cd_w, ker_w, hide_w, min_s, inv_range;
  int B;
};
// d loss / d adv[b,row,n]: a = adv[b,row,n], o = ori[b,arg(row),n] with arg = the nearest ori row of the forward pass
__device__ __forceinline__ float reg_grad_adv(const RegGrad &g, float a, float o) {
  const float k = 1.0f * g.scal[0];
  return g.cd_w != 0.f ? k * (g.cd_w / (3.0f * (float)g.B)) * 2.0f * (a - o) : 0.f;
}
__device__ __forceinline__ float reg_grad_perturb(const RegGrad &g, int C, float p) {
  const float k = 1.0f * g.scal[0];
  return g.ker_w != 0.f ? k * (g.ker_w / (float)C) * g.scal[1] * p : 0.f;
}
__device__ __forceinline__ float reg_grad_sigma(const RegGrad &g, int C, int b, float s, float href) {
  const float k = 1.0f * g.scal[0];
  float v = 0.f;
  if (g.ker_w != 0.f) v += (g.ker_w / (float)C) * g.scal[2] * (s - 1.0f);
  if (g.hide_w != 0.f) {
    const float *pc = g.per_cloud + (size_t)b * 8;
    const float n = (s - g.min_s) * g.inv_range;
    v += (g.hide_w / (float)g.B) * g.inv_range * (href * pc[1] - pc[0] * n * pc[2]);
  }
  return k * v;
}

}  // namespace hitadv
