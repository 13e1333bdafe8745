// Fewer, fatter launches around the victim inside one HiT-ADV iteration (ShapeAttack/HiT_ADV.py:156-246).  At one attack in
// flight the iteration is a chain of dependent launches of ~4-7 us each; these entry points merge neighbours of that chain
// whose data dependence allows it, without changing a single result bit:
//   hitadv_iteration_head       best-result tracking (hitadv_best_update) + adversarial loss and d loss / d logits
//                               (hitadv_adv_loss): both consume only the logits; one block per cloud, the batch mean of
//                               the loss is taken in cloud order by the last block to arrive
//   hitadv_deform_bwd_partials  the deformation backward WITHOUT its reduce launch ...
//   hitadv_adam_step_partials   ... whose fixed-order slab sum happens inside the Adam kernel that consumes it
#include "common.hpp"
#include "hitadv.h"
#include "regulariser_body.hpp"
#include "adam_body.hpp"

namespace hitadv {

__device__ __forceinline__ uint32_t ordered_bits_i(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// One block (256 threads) per cloud.  Waves 0..3 share the transformation-loss sums; wave 0 finds the prediction, wave 1
// the adversarial loss row.  scratch: per[B] floats, then one int ticket (zeroed once by the caller).
struct HeadArgs {
  const float *logits;
  const int64_t *label;
  const float *perturb, *sigma, *adv;
  int B, num_class, N, C;
  float *bestdist;
  int64_t *bestscore;
  float *o_bestdist;
  int64_t *o_bestscore;
  float *o_bestattack;
  int64_t *pred_out;
  float *dist_val_out;
  int32_t *iter_counter;
  int kind;
  float kappa;
  float *loss, *dlogits, *per;
  int *ticket;
  // optional: the logits themselves, logits[b,:] = feat[b,:feat_dim] @ Wlog[feat_dim,num_class] + blog, evaluated here (the
  // classifier's last layer, model/feature_models.py:91) instead of by a launch of its own; written to logits_out
  const float *feat, *Wlog, *blog;
  int feat_dim;
  float *logits_out;
};

__device__ __forceinline__ void iteration_head_body(const HeadArgs &a, const int b) {
  const float *__restrict__ logits = a.logits, *__restrict__ perturb = a.perturb, *__restrict__ sigma = a.sigma,
                           *__restrict__ adv = a.adv;
  const int64_t *__restrict__ label = a.label;
  const int B = a.B, num_class = a.num_class, N = a.N, C = a.C, kind = a.kind;
  const float kappa = a.kappa;
  float *__restrict__ bestdist = a.bestdist, *__restrict__ o_bestdist = a.o_bestdist, *__restrict__ o_bestattack = a.o_bestattack,
                      *__restrict__ dist_val_out = a.dist_val_out, *__restrict__ loss = a.loss, *__restrict__ dlogits = a.dlogits;
  int64_t *__restrict__ bestscore = a.bestscore, *__restrict__ o_bestscore = a.o_bestscore, *__restrict__ pred_out = a.pred_out;
  int32_t *__restrict__ iter_counter = a.iter_counter;
  float *per = a.per;
  int *ticket = a.ticket;
  __shared__ float s1[4], s2[4];
  __shared__ int s_copy, s_last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // transformation_loss(batch_avg=False): (|P_b|_F + |1 - sigma_b|_2) / C   -- as best_update_k
  // Everything the block reads is requested before anything is used: a load inside a loop whose trip count the compiler
  // does not know, or under a condition, is a dependent global round trip of its own (this kernel had about ten of them
  // in a row).  Out-of-range slots read a clamped address and enter the sums multiplied by an exact 0.
  float a1 = 0.f, a2 = 0.f;
  const float *pp = perturb + (size_t)b * C * 3;
  const float *sp = sigma + (size_t)b * C;
  const float *z = logits + (size_t)b * num_class;
  float z_lane;                                                // waves 0 and 1: the row, when it fits a wave
  if (a.feat != nullptr) {  // block-uniform; requires num_class <= 64 and feat_dim <= 256 (checked by the host)
    // wave w takes features 64 w .. 64 w + 63, lane c accumulates class c over them in ascending order (the feature is
    // broadcast from the lane that loaded it), then the four waves are added in order and the bias last
    __shared__ float s_z[4][64];
    const int k = 64 * wave + lane;
    const float f = k < a.feat_dim ? a.feat[(size_t)b * a.feat_dim + k] : 0.f;
    const int cc = min(lane, num_class - 1);
    float wcol[64];
#pragma unroll
    for (int t = 0; t < 64; ++t) wcol[t] = a.Wlog[(size_t)min(64 * wave + t, a.feat_dim - 1) * num_class + cc];
    const float bl = a.blog[cc];
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 64; ++t) acc = fmaf(__shfl(f, t, HITADV_WAVE), wcol[t], acc);  // f is 0 past feat_dim
    s_z[wave][lane] = acc;
    __syncthreads();
    z_lane = ((s_z[0][lane] + s_z[1][lane]) + (s_z[2][lane] + s_z[3][lane])) + bl;
    if (wave == 0 && lane < num_class) a.logits_out[(size_t)b * num_class + lane] = z_lane;
  } else {
    z_lane = z[min(lane, num_class - 1)];
  }
  const int64_t lab = label[b];
  const float bd0 = bestdist[b], obd0 = o_bestdist[b];
  for (int e0 = 0; e0 < C * 3; e0 += 1024) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = pp[min(e0 + (int)threadIdx.x + 256 * u, C * 3 - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float w = e0 + (int)threadIdx.x + 256 * u < C * 3 ? 1.0f : 0.0f;
      a1 = fmaf(v[u] * w, v[u], a1);  // the same fmaf chain in the same order as a plain strided loop
    }
  }
  for (int e0 = 0; e0 < C; e0 += 1024) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = sp[min(e0 + (int)threadIdx.x + 256 * u, C - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float w = e0 + (int)threadIdx.x + 256 * u < C ? 1.0f : 0.0f;
      const float t = (1.0f - v[u]) * w;
      a2 = fmaf(t, t, a2);
    }
  }
  a1 = wave_sum(a1);
  a2 = wave_sum(a2);
  if (lane == 0) {
    s1[wave] = a1;
    s2[wave] = a2;
  }
  const bool row_in_wave = num_class <= 64;  // block-uniform
  unsigned long long key = 0ull;
  if (wave == 0) {  // argmax over the logits, lowest index on ties
    for (int c = lane; c < num_class; c += 64) {
      const float zc = row_in_wave ? z_lane : z[c];
      const unsigned long long k = ((unsigned long long)ordered_bits_i(zc) << 32) | (0xFFFFFFFFu - (uint32_t)c);
      key = k > key ? k : key;
    }
    key = wave_max_u64(key);
  } else if (wave == 1) {  // adversarial loss row of this cloud -- as adv_loss_k
    float *d = dlogits + (size_t)b * num_class;
    const int t = (int)lab;
    const float invB = 1.0f / (float)B;
    // z[j] for the lane's own j comes from the register; z[o] / z[t] by a lane read of it
    auto zat = [&](int j) { return row_in_wave ? __shfl(z_lane, j, HITADV_WAVE) : z[j]; };
    auto zown = [&](int j) { return row_in_wave ? z_lane : z[j]; };
    float mine;
    if (kind == 2) {
      float mx = -__builtin_inff();
      for (int j = lane; j < num_class; j += 64) mx = fmaxf(mx, zown(j));
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, HITADV_WAVE));
      float se = 0.f;
      for (int j = lane; j < num_class; j += 64) se += __expf(zown(j) - mx);
      se = wave_sum(se);
      const float lse = mx + __logf(se);
      for (int j = lane; j < num_class; j += 64) d[j] = (__expf(zown(j) - lse) - (j == t ? 1.0f : 0.f)) * invB;
      mine = lse - zat(t);
    } else {
      unsigned long long k2 = 0ull;
      for (int j = lane; j < num_class; j += 64) {
        const float v = j == t ? -10000.f : zown(j);
        const unsigned long long k = ((unsigned long long)ordered_bits_i(v) << 32) | (0xFFFFFFFFu - (uint32_t)j);
        k2 = k > k2 ? k : k2;
      }
      k2 = wave_max_u64(k2);
      const int o = (int)(0xFFFFFFFFu - (uint32_t)(k2 & 0xffffffffu));
      const float zo = zat(o), zt = zat(t);
      const float other = o == t ? -10000.f : zo;
      const float margin = kind == 0 ? (zt - other) + kappa : (other - zt) + kappa;
      const bool on = margin >= 0.f;  // torch's clamp(min=0) passes the gradient at the boundary
      const float s = on ? (kind == 0 ? invB : -invB) : 0.f;
      // (not two classes per packed instruction: the loop vectoriser's v_cndmask_b32 -> v_pk_add_f32 is the instruction pair
      // tests/test_isa_guards.py keeps out of every product kernel -- docs/kernels/round6.md section 1; same bits either way)
#pragma clang loop vectorize(disable) interleave(disable)
      for (int j = lane; j < num_class; j += 64) d[j] = (j == t ? s : 0.f) - ((j == o && o != t) ? s : 0.f);
      mine = margin > 0.f ? margin : 0.f;
    }
    if (lane == 0) __hip_atomic_store(&per[b], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float dv = (__builtin_sqrtf((s1[0] + s1[1]) + (s1[2] + s1[3])) +
                      __builtin_sqrtf((s2[0] + s2[1]) + (s2[2] + s2[3]))) / (float)C;
    const int64_t pred = (int64_t)(0xFFFFFFFFu - (uint32_t)(key & 0xffffffffu));
    pred_out[b] = pred;
    dist_val_out[b] = dv;
    int copy = 0;
    if (pred != lab) {
      if (dv < bd0) {
        bestdist[b] = dv;
        bestscore[b] = pred;
      }
      if (dv < obd0) {
        o_bestdist[b] = dv;
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
  // batch mean of the loss, in cloud order, by the last block to arrive (hand-off protocol: common.hpp)
  if (!handoff_last_arriver(ticket, 0, B, &s_last)) return;
  if (threadIdx.x < 64) {  // wave 0: sixty-four clouds' values per round trip, added by lane 0 in cloud order
    float a = 0.f;
    for (int q0 = 0; q0 < B; q0 += 64) {
      const float v = __hip_atomic_load(&per[min(q0 + lane, B - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int n = min(64, B - q0);
      for (int q = 0; q < n; ++q) a += __shfl(v, q, HITADV_WAVE);
    }
    if (threadIdx.x == 0) {
      loss[0] = a / (float)B;
      __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ __launch_bounds__(256) void iteration_head_k(HeadArgs a) { iteration_head_body(a, blockIdx.x); }

// iteration_head and the regularisers' forward pass in ONE launch: both are one block per cloud with a last-arriver
// reduction, and the regularisers need nothing the victim produces (perturb, sigma and the deformed cloud only) -- blocks
// 0 .. B-1 run the head, blocks B .. 2B-1 the regularisers' per-cloud pass.  Same code, same bits, one launch less.
struct RegArgs {
  const float *ori, *hide_ref;
  float min_s, inv_range;
  float *part;
  RegFin fin;
};
// G attacks stacked: every per-cloud argument is the group-0 pointer of a buffer that holds the G groups one after the other
// (B clouds each), every per-group scalar / scratch likewise with its own stride.  blockIdx.y = the group: the arguments are
// moved to that group's rows and the unchanged per-group bodies run on them -- the bits of G separate launches.
constexpr int RG_NSCAL_I = RG_NSCAL;
__device__ __forceinline__ void shift_group(HeadArgs &a, RegArgs &g, const int grp) {
  if (grp == 0) return;
  const size_t cl = (size_t)grp * a.B;  // clouds in front of this group
  a.logits += cl * a.num_class;
  a.label += cl;
  a.perturb += cl * a.C * 3;
  a.sigma += cl * a.C;
  a.adv += cl * 3 * a.N;
  a.bestdist += cl;
  a.bestscore += cl;
  a.o_bestdist += cl;
  a.o_bestscore += cl;
  a.o_bestattack += cl * 3 * a.N;
  a.pred_out += cl;
  a.dist_val_out += cl;
  if (a.iter_counter != nullptr) a.iter_counter += grp;
  a.loss += grp;
  a.dlogits += cl * a.num_class;
  a.per += (size_t)grp * (a.B + 4);  // hitadv_iteration_head_scratch_floats
  a.ticket = reinterpret_cast<int *>(a.per + a.B);
  if (a.feat != nullptr) {
    a.feat += cl * a.feat_dim;
    a.logits_out += cl * a.num_class;
  }
  const size_t rs = (size_t)grp * ((size_t)a.B * (RG_NPART + 8) + RG_NSCAL_I);  // hitadv_regulariser_scratch_floats
  g.ori += cl * 3 * a.N;
  g.hide_ref += cl * a.C;
  g.part += rs;
  g.fin.scale_const += cl;
  g.fin.per_cloud += rs;
  g.fin.scal += rs;
  g.fin.dist_out += grp;
  g.fin.scaled_out += grp;
}

__global__ __launch_bounds__(256) void iteration_head_reg_k(HeadArgs a, RegArgs g) {
  shift_group(a, g, blockIdx.y);
  if ((int)blockIdx.x < a.B)
    iteration_head_body(a, blockIdx.x);
  else
    reg_partials_body(a.perturb, a.sigma, a.adv, g.ori, g.hide_ref, a.N, a.C, g.min_s, g.inv_range, g.part, g.fin, 1,
                      (int)blockIdx.x - a.B);
}

// Adam on (perturb [B,C,3], sigma [B,C]) with the deformation's gradient still in its per-slab partials: adam_body.hpp.
// One thread per (cloud, centre).
__global__ __launch_bounds__(256) void adam_partials_k(AdamArgs a) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= a.B * a.C) return;
  if (blockIdx.y != 0) {  // stacked groups (see shift_group): this group's rows of every buffer
    const int grp = blockIdx.y;
    const size_t cl = (size_t)grp * a.B, cc = cl * a.C;
    a.P += cc * 3; a.S += cc;
    a.partials += cl * a.nslab * 4 * a.C;
    a.mP += cc * 3; a.vP += cc * 3; a.mS += cc; a.vS += cc;
    a.step += grp;
    if (a.rg.per_cloud != nullptr) {
      const size_t rs = (size_t)grp * ((size_t)a.B * (RG_NPART + 8) + RG_NSCAL_I);
      a.rg.per_cloud += rs;
      a.rg.scal += rs;
      a.rg.hide_ref += cc;
    }
  }
  adam_partials_body<false>(a, e / a.C, e % a.C);
}

}  // namespace hitadv

using namespace hitadv;

extern "C" int64_t hitadv_iteration_head_scratch_floats(int B) { return B > 0 ? (int64_t)B + 4 : 0; }

extern "C" int hitadv_iteration_head(const float *logits, const int64_t *label, const float *perturb, const float *sigma,
                                     const float *adv, int B, int num_class, int N, int C, float *bestdist,
                                     int64_t *bestscore, float *o_bestdist, int64_t *o_bestscore, float *o_bestattack,
                                     int64_t *pred_out, float *dist_val_out, int32_t *iter_counter, int kind, float kappa,
                                     float *loss, float *dlogits, float *scratch, void *stream) {
  if (!logits || !label || !perturb || !sigma || !adv || !bestdist || !bestscore || !o_bestdist || !o_bestscore ||
      !o_bestattack || !pred_out || !dist_val_out || !loss || !dlogits || !scratch || kind < 0 || kind > 2 || B <= 0 ||
      num_class <= 0 || N <= 0 || C <= 0)
    return HITADV_E_ARG;
  const HeadArgs a{logits, label, perturb, sigma, adv, B, num_class, N, C, bestdist, bestscore, o_bestdist, o_bestscore,
                   o_bestattack, pred_out, dist_val_out, iter_counter, kind, kappa, loss, dlogits, scratch,
                   reinterpret_cast<int *>(scratch + B), nullptr, nullptr, nullptr, 0, nullptr};
  iteration_head_k<<<B, 256, 0, (hipStream_t)stream>>>(a);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_iteration_head_reg(const float *logits, const int64_t *label, const float *perturb, const float *sigma,
                                         const float *adv, int B, int num_class, int N, int C, float *bestdist,
                                         int64_t *bestscore, float *o_bestdist, int64_t *o_bestscore, float *o_bestattack,
                                         int64_t *pred_out, float *dist_val_out, int32_t *iter_counter, int kind,
                                         float kappa, float *loss, float *dlogits, float *scratch, const float *ori,
                                         const float *hide_ref, const float *scale_const, float cd_w, float ker_w,
                                         float hide_w, float min_sigm, float max_sigm, float *reg_scratch, float *dist_loss,
                                         float *scaled_loss, const float *feat, const float *Wlog, const float *blog,
                                         int feat_dim, void *stream) {
  if (feat != nullptr && (!Wlog || !blog || feat_dim <= 0 || feat_dim > 256 || num_class > 64)) return HITADV_E_ARG;
  if (!logits || !label || !perturb || !sigma || !adv || !bestdist || !bestscore || !o_bestdist || !o_bestscore ||
      !o_bestattack || !pred_out || !dist_val_out || !loss || !dlogits || !scratch || kind < 0 || kind > 2 || B <= 0 ||
      num_class <= 0 || N <= 0 || C <= 0 || !ori || !hide_ref || !scale_const || !reg_scratch || !dist_loss || !scaled_loss)
    return HITADV_E_ARG;
  const HeadArgs a{logits, label, perturb, sigma, adv, B, num_class, N, C, bestdist, bestscore, o_bestdist, o_bestscore,
                   o_bestattack, pred_out, dist_val_out, iter_counter, kind, kappa, loss, dlogits, scratch,
                   reinterpret_cast<int *>(scratch + B), feat, Wlog, blog, feat_dim, const_cast<float *>(logits)};
  float *part = reg_scratch, *per_cloud = reg_scratch + (size_t)B * RG_NPART, *scal = per_cloud + (size_t)B * 8;
  const RegArgs g{ori, hide_ref, min_sigm, 1.0f / (max_sigm - min_sigm + 1e-7f), part,
                  RegFin{scale_const, B, cd_w, ker_w, hide_w, per_cloud, scal, dist_loss, scaled_loss}};
  iteration_head_reg_k<<<2 * B, 256, 0, (hipStream_t)stream>>>(a, g);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_iteration_head_reg_stack(int G, const float *logits, const int64_t *label, const float *perturb,
                                               const float *sigma, const float *adv, int B, int num_class, int N, int C,
                                               float *bestdist, int64_t *bestscore, float *o_bestdist, int64_t *o_bestscore,
                                               float *o_bestattack, int64_t *pred_out, float *dist_val_out,
                                               int32_t *iter_counter, int kind, float kappa, float *loss, float *dlogits,
                                               float *scratch, const float *ori, const float *hide_ref,
                                               const float *scale_const, float cd_w, float ker_w, float hide_w,
                                               float min_sigm, float max_sigm, float *reg_scratch, float *dist_loss,
                                               float *scaled_loss, const float *feat, const float *Wlog, const float *blog,
                                               int feat_dim, void *stream) {
  if (G <= 0 || G > 65535) return HITADV_E_ARG;
  if (feat != nullptr && (!Wlog || !blog || feat_dim <= 0 || feat_dim > 256 || num_class > 64)) return HITADV_E_ARG;
  if (!logits || !label || !perturb || !sigma || !adv || !bestdist || !bestscore || !o_bestdist || !o_bestscore ||
      !o_bestattack || !pred_out || !dist_val_out || !loss || !dlogits || !scratch || kind < 0 || kind > 2 || B <= 0 ||
      num_class <= 0 || N <= 0 || C <= 0 || !ori || !hide_ref || !scale_const || !reg_scratch || !dist_loss || !scaled_loss)
    return HITADV_E_ARG;
  const HeadArgs a{logits, label, perturb, sigma, adv, B, num_class, N, C, bestdist, bestscore, o_bestdist, o_bestscore,
                   o_bestattack, pred_out, dist_val_out, iter_counter, kind, kappa, loss, dlogits, scratch,
                   reinterpret_cast<int *>(scratch + B), feat, Wlog, blog, feat_dim, const_cast<float *>(logits)};
  float *part = reg_scratch, *per_cloud = reg_scratch + (size_t)B * RG_NPART, *scal = per_cloud + (size_t)B * 8;
  const RegArgs g{ori, hide_ref, min_sigm, 1.0f / (max_sigm - min_sigm + 1e-7f), part,
                  RegFin{scale_const, B, cd_w, ker_w, hide_w, per_cloud, scal, dist_loss, scaled_loss}};
  iteration_head_reg_k<<<dim3(2 * B, G), 256, 0, (hipStream_t)stream>>>(a, g);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_adam_step_partials_reg_stack(int G, float *perturb, float *sigma, const float *partials, int nslab,
                                                   const float *hide_ref, const float *reg_scratch, float cd_w, float ker_w,
                                                   float hide_w, float min_sigm, float max_sigm, float *m_perturb,
                                                   float *v_perturb, float *m_sigma, float *v_sigma, int B, int C,
                                                   float lr_perturb, float lo_perturb, float hi_perturb, float lr_sigma,
                                                   float lo_sigma, float hi_sigma, const int32_t *step, void *stream) {
  if (G <= 0 || G > 65535 || !perturb || !sigma || !partials || !hide_ref || !reg_scratch || !m_perturb || !v_perturb ||
      !m_sigma || !v_sigma || !step || nslab <= 0 || B <= 0 || C <= 0)
    return HITADV_E_ARG;
  const float *per_cloud = reg_scratch + (size_t)B * RG_NPART, *scal = per_cloud + (size_t)B * 8;
  const RegGrad rg{per_cloud, scal, hide_ref, cd_w, ker_w, hide_w, min_sigm, 1.0f / (max_sigm - min_sigm + 1e-7f), B};
  adam_partials_k<<<dim3((B * C + 255) / 256, G), 256, 0, (hipStream_t)stream>>>(AdamArgs{perturb, sigma, partials, nslab, nullptr, nullptr, m_perturb, v_perturb, m_sigma, v_sigma, B, C, lr_perturb, lo_perturb, hi_perturb, lr_sigma, lo_sigma, hi_sigma, step, rg});
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_adam_step_partials(float *perturb, float *sigma, const float *partials, int nslab,
                                         const float *g_perturb2, const float *g_sigma2, float *m_perturb,
                                         float *v_perturb, float *m_sigma, float *v_sigma, int B, int C, float lr_perturb,
                                         float lo_perturb, float hi_perturb, float lr_sigma, float lo_sigma,
                                         float hi_sigma, const int32_t *step, void *stream) {
  if (!perturb || !sigma || !partials || !m_perturb || !v_perturb || !m_sigma || !v_sigma || !step || nslab <= 0 ||
      B <= 0 || C <= 0)
    return HITADV_E_ARG;
  adam_partials_k<<<(B * C + 255) / 256, 256, 0, (hipStream_t)stream>>>(AdamArgs{perturb, sigma, partials, nslab, g_perturb2, g_sigma2, m_perturb, v_perturb, m_sigma, v_sigma, B, C, lr_perturb, lo_perturb, hi_perturb, lr_sigma, lo_sigma, hi_sigma, step, RegGrad{}});
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_adam_step_partials_reg(float *perturb, float *sigma, const float *partials, int nslab,
                                             const float *hide_ref, const float *reg_scratch, float cd_w, float ker_w,
                                             float hide_w, float min_sigm, float max_sigm, float *m_perturb,
                                             float *v_perturb, float *m_sigma, float *v_sigma, int B, int C,
                                             float lr_perturb, float lo_perturb, float hi_perturb, float lr_sigma,
                                             float lo_sigma, float hi_sigma, const int32_t *step, void *stream) {
  if (!perturb || !sigma || !partials || !hide_ref || !reg_scratch || !m_perturb || !v_perturb || !m_sigma || !v_sigma ||
      !step || nslab <= 0 || B <= 0 || C <= 0)
    return HITADV_E_ARG;
  const float *per_cloud = reg_scratch + (size_t)B * RG_NPART, *scal = per_cloud + (size_t)B * 8;
  const RegGrad rg{per_cloud, scal, hide_ref, cd_w, ker_w, hide_w, min_sigm, 1.0f / (max_sigm - min_sigm + 1e-7f), B};
  adam_partials_k<<<(B * C + 255) / 256, 256, 0, (hipStream_t)stream>>>(AdamArgs{perturb, sigma, partials, nslab, nullptr, nullptr, m_perturb, v_perturb, m_sigma, v_sigma, B, C, lr_perturb, lo_perturb, hi_perturb, lr_sigma, lo_sigma, hi_sigma, step, rg});
  HITADV_LAUNCH_CHECK();
  return 0;
}
