// On-device attack bookkeeping: best-result tracking and the Adam step for (perturb, sigma).
// These replace the per-iteration device->host copies and Python loops of
// ShapeAttack/HiT_ADV.py:186-217 and the torch.optim.Adam instance of :142-145, so that a whole
// inner iteration can be captured into one hipGraph.
#include "common.hpp"
#include "hitadv.h"

namespace hitadv {

__device__ __forceinline__ uint32_t ordered_bits(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void best_update_k(
    const float *__restrict__ logits, const int64_t *__restrict__ label, const float *__restrict__ perturb,
    const float *__restrict__ sigma, const float *__restrict__ adv, int num_class, int N, int C,
    float *__restrict__ bestdist, int64_t *__restrict__ bestscore, float *__restrict__ o_bestdist,
    int64_t *__restrict__ o_bestscore, float *__restrict__ o_bestattack, int64_t *__restrict__ pred_out,
    float *__restrict__ dist_val_out, int32_t *__restrict__ iter_counter) {
  __shared__ float s1[4], s2[4];
  __shared__ int s_copy;
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // transformation_loss(batch_avg=False): (|P_b|_F + |1 - sigma_b|_2) / C
  float a1 = 0.f, a2 = 0.f;
  const float *pp = perturb + (size_t)b * C * 3;
  for (int e = threadIdx.x; e < C * 3; e += 256) a1 = fmaf(pp[e], pp[e], a1);
  const float *sp = sigma + (size_t)b * C;
  for (int e = threadIdx.x; e < C; e += 256) {
    const float t = 1.0f - sp[e];
    a2 = fmaf(t, t, a2);
  }
  a1 = wave_sum(a1);
  a2 = wave_sum(a2);
  if (lane == 0) {
    s1[wave] = a1;
    s2[wave] = a2;
  }
  // argmax over the logits, lowest index on ties (wave 0)
  unsigned long long key = 0ull;
  if (wave == 0) {
    for (int c = lane; c < num_class; c += 64) {
      const unsigned long long k =
          ((unsigned long long)ordered_bits(logits[(size_t)b * num_class + c]) << 32) | (0xFFFFFFFFu - (uint32_t)c);
      key = k > key ? k : key;
    }
    key = wave_max_u64(key);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float d = (__builtin_sqrtf((s1[0] + s1[1]) + (s1[2] + s1[3])) +
                     __builtin_sqrtf((s2[0] + s2[1]) + (s2[2] + s2[3]))) / (float)C;
    const int64_t pred = (int64_t)(0xFFFFFFFFu - (uint32_t)(key & 0xffffffffu));
    pred_out[b] = pred;
    dist_val_out[b] = d;
    int copy = 0;
    if (pred != label[b]) {
      if (d < bestdist[b]) {
        bestdist[b] = d;
        bestscore[b] = pred;
      }
      if (d < o_bestdist[b]) {
        o_bestdist[b] = d;
        o_bestscore[b] = pred;
        copy = 1;
      }
    }
    s_copy = copy;
    if (b == 0 && iter_counter != nullptr) *iter_counter += 1;
  }
  __syncthreads();
  if (s_copy) {
    const float *src = adv + (size_t)b * 3 * N;
    float *dst = o_bestattack + (size_t)b * 3 * N;
    for (int e = threadIdx.x; e < 3 * N; e += 256) dst[e] = src[e];
  }
}

__global__ __launch_bounds__(256) void adam_k(float *__restrict__ p0, const float *__restrict__ g0,
                                              float *__restrict__ m0, float *__restrict__ v0, long long n0,
                                              float lr0, float *__restrict__ p1, const float *__restrict__ g1,
                                              float *__restrict__ m1, float *__restrict__ v1, long long n1,
                                              float lr1, const int32_t *__restrict__ step) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n0 + n1) return;
  const bool second = e >= n0;
  const long long i = second ? e - n0 : e;
  float *p = second ? p1 : p0;
  const float *g = second ? g1 : g0;
  float *m = second ? m1 : m0;
  float *v = second ? v1 : v0;
  const double lr = second ? (double)lr1 : (double)lr0;
  const AdamCoef k = adam_coef(*step + 1, lr);  // csrc/arith.hpp: torch.optim.Adam's step, operation for operation
  float mi = m[i], vi = v[i];
  const float q = adam_update(p[i], g[i], mi, vi, k);
  m[i] = mi;
  v[i] = vi;
  p[i] = q;
}

__global__ void bump_k(int32_t *c) { *c += 1; }

// Adam on (perturb, sigma) with the gradient given as a sum of two terms (the deformation's and the regulariser's),
// followed by the projection the reference applies at the top of the NEXT iteration (HiT_ADV.py:157-158:
// perturb.clamp(-budget, budget), sigma.clamp(min_sigm, max_sigm)) -- same parameter values at every forward pass.
// `step` holds the 1-based step number (the caller's per-iteration kernel bumped it already).
__global__ __launch_bounds__(256) void adam2_k(float *__restrict__ p0, const float *__restrict__ g0,
                                               const float *__restrict__ h0, float *__restrict__ m0,
                                               float *__restrict__ v0, long long n0, float lr0, float lo0, float hi0,
                                               float *__restrict__ p1, const float *__restrict__ g1,
                                               const float *__restrict__ h1, float *__restrict__ m1,
                                               float *__restrict__ v1, long long n1, float lr1, float lo1, float hi1,
                                               const int32_t *__restrict__ step) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n0 + n1) return;
  const bool second = e >= n0;
  const long long i = second ? e - n0 : e;
  float *p = second ? p1 : p0;
  const float *g = second ? g1 : g0;
  const float *hh = second ? h1 : h0;
  float *m = second ? m1 : m0;
  float *v = second ? v1 : v0;
  const double lr = second ? (double)lr1 : (double)lr0;
  const float lo = second ? lo1 : lo0, hi = second ? hi1 : hi0;
  const AdamCoef k = adam_coef(*step, lr);
  const float gi = hh ? g[i] + hh[i] : g[i];
  float mi = m[i], vi = v[i];
  float q = adam_update(p[i], gi, mi, vi, k);
  m[i] = mi;
  v[i] = vi;
  if (lo <= hi) q = q < lo ? lo : (q > hi ? hi : q);
  p[i] = q;
}

// Adversarial losses of util/adv_utils.py on logits[B,K] with their gradient, one launch:
//   kind 0  UntargetedLogitsAdvLoss (:50-67)   mean_b max(z_t - max_{j != t} z_j + kappa, 0)
//   kind 1  LogitsAdvLoss           (:18-35)   mean_b max(max_{j != t} z_j - z_t + kappa, 0)
//   kind 2  CrossEntropyAdvLoss     (:77-85)   mean_b (logsumexp(z) - z_t)
// "max over the others" is the reference's max((1-onehot)*z - onehot*10000): the true class takes part with -10000.
// One wave per cloud; the mean is taken in cloud order by wave 0 of the single block.
__global__ __launch_bounds__(1024) void adv_loss_k(int kind, const float *__restrict__ logits,
                                                   const int64_t *__restrict__ target, int B, int K, float kappa,
                                                   float *__restrict__ loss, float *__restrict__ dlogits) {
  extern __shared__ float per[];  // B
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int b = wave; b < B; b += nw) {
    const float *z = logits + (size_t)b * K;
    float *d = dlogits + (size_t)b * K;
    const int t = (int)target[b];
    const float invB = 1.0f / (float)B;
    if (kind == 2) {
      float mx = -__builtin_inff();
      for (int j = lane; j < K; j += 64) mx = fmaxf(mx, z[j]);
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, HITADV_WAVE));
      float se = 0.f;
      for (int j = lane; j < K; j += 64) se += __expf(z[j] - mx);
      se = wave_sum(se);
      const float lse = mx + __logf(se);
      for (int j = lane; j < K; j += 64) d[j] = (__expf(z[j] - lse) - (j == t ? 1.0f : 0.f)) * invB;
      if (lane == 0) per[b] = lse - z[t];
    } else {
      unsigned long long key = 0ull;
      for (int j = lane; j < K; j += 64) {
        const float v = j == t ? -10000.f : z[j];
        const unsigned long long k = ((unsigned long long)ordered_bits(v) << 32) | (0xFFFFFFFFu - (uint32_t)j);
        key = k > key ? k : key;
      }
      key = wave_max_u64(key);
      const int o = (int)(0xFFFFFFFFu - (uint32_t)(key & 0xffffffffu));
      const float other = o == t ? -10000.f : z[o];
      const float margin = kind == 0 ? (z[t] - other) + kappa : (other - z[t]) + kappa;
      const bool on = margin >= 0.f;  // torch's clamp(min=0) passes the gradient at the boundary
      const float s = on ? (kind == 0 ? invB : -invB) : 0.f;
      // (not two classes per packed instruction: the loop vectoriser's v_cndmask_b32 -> v_pk_add_f32 is the instruction pair
      // tests/test_isa_guards.py keeps out of every product kernel -- docs/kernels/round6.md section 1; same bits either way)
#pragma clang loop vectorize(disable) interleave(disable)
      for (int j = lane; j < K; j += 64) d[j] = (j == t ? s : 0.f) - ((j == o && o != t) ? s : 0.f);
      if (lane == 0) per[b] = margin > 0.f ? margin : 0.f;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += per[b];
    loss[0] = a / (float)B;
  }
}

}  // namespace hitadv

using namespace hitadv;

extern "C" const char *hitadv_version(void) { return "hitadv-hip 0.1 (gfx950)"; }

extern "C" int hitadv_best_update(const float *logits, const int64_t *label, const float *perturb,
                                  const float *sigma, const float *adv, int B, int num_class, int N, int C,
                                  float *bestdist, int64_t *bestscore, float *o_bestdist, int64_t *o_bestscore,
                                  float *o_bestattack, int64_t *pred_out, float *dist_val_out,
                                  int32_t *iter_counter, void *stream) {
  if (!logits || !label || !perturb || !sigma || !adv || !bestdist || !bestscore || !o_bestdist || !o_bestscore ||
      !o_bestattack || !pred_out || !dist_val_out || B <= 0 || num_class <= 0 || N <= 0 || C <= 0)
    return HITADV_E_ARG;
  best_update_k<<<B, 256, 0, (hipStream_t)stream>>>(logits, label, perturb, sigma, adv, num_class, N, C, bestdist,
                                                    bestscore, o_bestdist, o_bestscore, o_bestattack, pred_out,
                                                    dist_val_out, iter_counter);
  HITADV_LAUNCH_CHECK();
  return 0;
}

// byte copy as a kernel: 16-byte words where both addresses allow, single bytes for the rest
__global__ __launch_bounds__(256) void copy_k(uint4 *__restrict__ dst, const uint4 *__restrict__ src, long long words,
                                              unsigned char *__restrict__ dtail, const unsigned char *__restrict__ stail, int tail) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < words; i += stride) dst[i] = src[i];
  if (blockIdx.x == 0 && (int)threadIdx.x < tail) dtail[threadIdx.x] = stail[threadIdx.x];
}

__global__ __launch_bounds__(256) void copy_bytes_k(unsigned char *__restrict__ dst, const unsigned char *__restrict__ src,
                                                    long long n) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

// [B,C,N] <-> [B,N,C] for a few channels (a cloud's coordinates: C = 3): one thread per point, the channel-major side read / written
// coalesced over the points.  torch's strided copy kernel moves the 1.5 MB of a 64 x 2048 cloud in 24 us (65 GB/s), twice per victim
// pass and level (`xyz.permute(0, 2, 1).contiguous()` and its backward).
template <bool TO_POINTS_MAJOR>
__global__ __launch_bounds__(256) void transpose_small_k(const float *__restrict__ src, float *__restrict__ dst, int C, int N,
                                                         long long total) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;  // (b, n)
  if (e >= total) return;
  const long long b = e / N;
  const int n = (int)(e - b * N);
  for (int c = 0; c < C; ++c) {
    const size_t cm = ((size_t)b * C + c) * N + n, pm = (size_t)e * C + c;
    if (TO_POINTS_MAJOR) dst[pm] = src[cm];
    else dst[cm] = src[pm];
  }
}

extern "C" int hitadv_transpose_small(const float *src, float *dst, int B, int C, int N, int to_points_major, void *stream) {
  if (!src || !dst || B <= 0 || C <= 0 || C > 16 || N <= 0) return HITADV_E_ARG;
  const long long total = (long long)B * N;
  if (to_points_major)
    transpose_small_k<true><<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(src, dst, C, N, total);
  else
    transpose_small_k<false><<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(src, dst, C, N, total);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_copy(void *dst, const void *src, int64_t nbytes, void *stream) {
  if (nbytes < 0 || (nbytes > 0 && (!dst || !src))) return HITADV_E_ARG;
  if (nbytes == 0 || dst == src) return 0;
  hipStream_t s = (hipStream_t)stream;
  if ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0) {
    const long long words = nbytes / 16;
    const int tail = (int)(nbytes - words * 16);
    const long long blocks = words > 0 ? (words + 255) / 256 : 1;
    copy_k<<<(unsigned)(blocks < 4096 ? blocks : 4096), 256, 0, s>>>(
        reinterpret_cast<uint4 *>(dst), reinterpret_cast<const uint4 *>(src), words,
        reinterpret_cast<unsigned char *>(dst) + words * 16, reinterpret_cast<const unsigned char *>(src) + words * 16, tail);
  } else {
    const long long blocks = (nbytes + 255) / 256;
    copy_bytes_k<<<(unsigned)(blocks < 4096 ? blocks : 4096), 256, 0, s>>>(reinterpret_cast<unsigned char *>(dst),
                                                                          reinterpret_cast<const unsigned char *>(src), nbytes);
  }
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_adam_step(float *perturb, const float *g_perturb, float *m_perturb, float *v_perturb,
                                int64_t n_perturb, float lr_perturb, float *sigma, const float *g_sigma,
                                float *m_sigma, float *v_sigma, int64_t n_sigma, float lr_sigma, int32_t *step,
                                void *stream) {
  if (!perturb || !g_perturb || !m_perturb || !v_perturb || !step || n_perturb <= 0 || n_sigma < 0)
    return HITADV_E_ARG;
  if (n_sigma > 0 && (!sigma || !g_sigma || !m_sigma || !v_sigma)) return HITADV_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long long total = n_perturb + n_sigma;
  adam_k<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(perturb, g_perturb, m_perturb, v_perturb, n_perturb,
                                                         lr_perturb, sigma, g_sigma, m_sigma, v_sigma, n_sigma,
                                                         lr_sigma, step);
  bump_k<<<1, 1, 0, s>>>(step);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_adam_step_sum(float *perturb, const float *g_perturb, const float *g_perturb2, float *m_perturb,
                                    float *v_perturb, int64_t n_perturb, float lr_perturb, float lo_perturb,
                                    float hi_perturb, float *sigma, const float *g_sigma, const float *g_sigma2,
                                    float *m_sigma, float *v_sigma, int64_t n_sigma, float lr_sigma, float lo_sigma,
                                    float hi_sigma, const int32_t *step, void *stream) {
  if (!perturb || !g_perturb || !m_perturb || !v_perturb || !step || n_perturb <= 0 || n_sigma < 0)
    return HITADV_E_ARG;
  if (n_sigma > 0 && (!sigma || !g_sigma || !m_sigma || !v_sigma)) return HITADV_E_ARG;
  const long long total = n_perturb + n_sigma;
  adam2_k<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      perturb, g_perturb, g_perturb2, m_perturb, v_perturb, n_perturb, lr_perturb, lo_perturb, hi_perturb, sigma,
      g_sigma, g_sigma2, m_sigma, v_sigma, n_sigma, lr_sigma, lo_sigma, hi_sigma, step);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_adv_loss(int kind, const float *logits, const int64_t *target, int B, int K, float kappa,
                               float *loss, float *dlogits, void *stream) {
  if (kind < 0 || kind > 2 || !logits || !target || !loss || !dlogits || B <= 0 || K <= 0 || B > 8192)
    return HITADV_E_ARG;
  const int threads = B >= 16 ? 1024 : 64 * B;
  adv_loss_k<<<1, threads, (size_t)B * sizeof(float), (hipStream_t)stream>>>(kind, logits, target, B, K, kappa, loss,
                                                                             dlogits);
  HITADV_LAUNCH_CHECK();
  return 0;
}
