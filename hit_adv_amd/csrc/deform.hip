// HiT-ADV kernel-weighted deformation, forward and backward.
//
//   k[n,j]  = exp(-|x_n - c_j| / (2 sigma_j^2))          (un-squared norm, HiT_ADV.py:298-304)
//   adv_n   = x_n + (sum_j k[n,j] p_j) / (sum_j k[n,j])  (== the 192-step loop of :160-175)
//
//   K3f deform_fwd   lanes own points; the cloud's centre table (c, p, -log2e/(2 sigma^2)) sits in
//                    LDS and is read as broadcasts; the 4 waves of a block split the centre range
//                    and merge partial sums through LDS in fixed order.  Transcendental/VALU-bound.
//   K3b deform_bwd   lanes own CENTRES (the reduction axis N then runs inside a lane, no cross-lane
//                    traffic); each block covers a 64-point slab whose per-point terms are
//                    precomputed into LDS; per-slab partials are summed in fixed order by
//                    deform_bwd_reduce -> bitwise reproducible gradients, no atomics.
#include "common.hpp"
#include "hitadv.h"
#include "regulariser_body.hpp"
#include "deform_body.hpp"
#include "adam_body.hpp"

namespace hitadv {

constexpr int DF_CMAX = 1024; // centres staged per LDS pass

__global__ __launch_bounds__(256) void deform_fwd(const float *__restrict__ ori,
                                                  const float *__restrict__ central,
                                                  const float *__restrict__ perturb,
                                                  const float *__restrict__ sigma, int N, int C,
                                                  float *__restrict__ adv, float *__restrict__ inv_den) {
  deform_fwd_body<DF_CMAX>(ori, central, perturb, sigma, N, C, adv, inv_den, blockIdx.y, blockIdx.x, nullptr);
}

// ---------------------------------------------------------------------------------- backward
// With S = sum_j k p_j, Dn = sum_j k:  adv = x + S/Dn
//   dL/dp_j[d]  = sum_n g[n,d]/Dn * k[n,j]
//   dL/dk[n,j]  = sum_d g[n,d]/Dn * p_j[d] - sum_d g[n,d]/Dn * (adv[n,d]-x[n,d])
//   dL/dsig_j   = sum_n dL/dk[n,j] * k[n,j] * r[n,j] / sig_j^3
constexpr int DB_PTS = 64;  // points per slab

struct AdamTail {
  int *tickets;  // [B], zeroed once by the caller; nullptr = no tail
  AdamArgs adam;
};

// threads per block of deform_bwd (one per centre): whole waves, at most four -- C = 192 on 256 threads left a wave idle
static inline int deform_bwd_width(int C) { return C >= 256 ? 256 : 64 * ((C + 63) / 64); }

__global__ __launch_bounds__(256) void deform_bwd(const float *__restrict__ ori,
                                                  const float *__restrict__ central,
                                                  const float *perturb,  // (no __restrict__: the Adam tail writes them)
                                                  const float *sigma,
                                                  const float *__restrict__ adv,
                                                  const float *__restrict__ inv_den,
                                                  const float *__restrict__ g_adv, int N, int C,
                                                  int nslab, float *partials, RegGrad rg, AdamTail tail) {
  __shared__ float4 sxyz[DB_PTS];  // x y z cn
  __shared__ float4 sg[DB_PTS];    // gDx gDy gDz -
  __shared__ int s_last;
  const int b = blockIdx.z, slab = blockIdx.y;
  // stacked groups (csrc/iteration.hip::shift_group): the per-cloud arrays are dense over all groups' clouds, so b indexes
  // them as it is; only the regularisers' per-group scratch (rg.B clouds per group) moves with the group
  if (rg.per_cloud != nullptr && b >= rg.B) {
    const size_t rs = (size_t)(b / rg.B) * ((size_t)rg.B * (RG_NPART + 8) + RG_NSCAL) - (size_t)(b / rg.B) * rg.B * 8;
    rg.per_cloud += rs;  // per_cloud is indexed by the GLOBAL b below: take the group's base back by its clouds' rows
    rg.scal += (size_t)(b / rg.B) * ((size_t)rg.B * (RG_NPART + 8) + RG_NSCAL);
  }
  const int n0 = slab * DB_PTS;
  const int cnt = min(DB_PTS, N - n0);
  // the thread's centre is requested together with the slab's points, not after the barrier behind them (one global round
  // trip instead of two); threads past C read centre C-1 and leave before they would write
  const int j = blockIdx.x * blockDim.x + threadIdx.x;  // blockDim.x = the centres rounded up to whole waves, at most 256
  const int jc = min(j, C - 1);
  const float *cp = central + (size_t)b * 3 * C + jc;
  const float cx = cp[0], cy = cp[C], cz = cp[2 * C];
  const float *pp = perturb + ((size_t)b * C + jc) * 3;
  const float px = pp[0], py = pp[1], pz = pp[2];
  const float s = sigma[(size_t)b * C + jc];
  if (threadIdx.x < DB_PTS) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), g = a;
    if (threadIdx.x < cnt) {
      const int n = n0 + threadIdx.x;
      const float *op = ori + (size_t)b * 3 * N;
      const float *ap = adv + (size_t)b * 3 * N;
      const float *gp = g_adv + (size_t)b * 3 * N;
      const float inv = inv_den[(size_t)b * N + n];
      const float x = op[n], y = op[N + n], z = op[2 * N + n];
      const float ax = ap[n], ay = ap[N + n], az = ap[2 * N + n];
      float g0 = gp[n], g1 = gp[N + n], g2 = gp[2 * N + n];
      if (rg.per_cloud != nullptr) {  // block-uniform: the regularisers' gradient at the deformed cloud, added on the way in
        const float *pc = rg.per_cloud + (size_t)b * 8;
        const int a0 = (int)pc[3], a1 = (int)pc[4], a2 = (int)pc[5];
        g0 = reg_grad_adv(rg, ax, a0 == 0 ? x : (a0 == 1 ? y : z)) + g0;
        g1 = reg_grad_adv(rg, ay, a1 == 0 ? x : (a1 == 1 ? y : z)) + g1;
        g2 = reg_grad_adv(rg, az, a2 == 0 ? x : (a2 == 1 ? y : z)) + g2;
      }
      const float gx = g0 * inv, gy = g1 * inv, gz = g2 * inv;
      const float cn = gx * (ax - x) + gy * (ay - y) + gz * (az - z);
      a = make_float4(x, y, z, cn);
      g = make_float4(gx, gy, gz, 0.f);
    }
    sxyz[threadIdx.x] = a;
    sg[threadIdx.x] = g;
  }
  __syncthreads();
  if (j >= C && tail.tickets == nullptr) return;
  const float a2 = -LOG2E / (2.0f * s * s);
  float apx = 0.f, apy = 0.f, apz = 0.f, asg = 0.f;
#pragma unroll 4
  for (int t = 0; t < (j < C ? cnt : 0); ++t) {
    const float4 pt = sxyz[t];
    const float4 g = sg[t];
    const float r = sqrt_rn_ranged(sqdist3(pt.x, pt.y, pt.z, cx, cy, cz));
    const float k = exp2_flush(r * a2);
    apx = fmaf(g.x, k, apx);
    apy = fmaf(g.y, k, apy);
    apz = fmaf(g.z, k, apz);
    const float dk = fmaf(g.x, px, fmaf(g.y, py, g.z * pz)) - pt.w;
    asg = fmaf(dk * k, r, asg);
  }
  float *out = partials + (((size_t)b * nslab + slab) * 4) * C + j;
  if (tail.tickets == nullptr) {
    out[0] = apx;
    out[C] = apy;
    out[2 * C] = apz;
    out[3 * C] = asg / (s * s * s);
    return;
  }
  // With an Adam tail: the partials are written through, the blocks of the cloud (all slabs, all centre groups) draw a
  // ticket, and the last one to arrive runs the update for the cloud's centres -- every block of the cloud has read
  // perturb / sigma long before (hand-off protocol: common.hpp).  The ticket goes back to zero.
  if (j < C) {
    __hip_atomic_store(out, apx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(out + C, apy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(out + 2 * C, apz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(out + 3 * C, asg / (s * s * s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!handoff_last_arriver(tail.tickets, b, nslab * (int)gridDim.x, &s_last)) return;
  for (int jj = threadIdx.x; jj < C; jj += blockDim.x) adam_partials_body<true>(tail.adam, b, jj);
  if (threadIdx.x == 0) __hip_atomic_store(&tail.tickets[b], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void deform_bwd_reduce(const float *__restrict__ partials, int C,
                                                         int nslab, float *__restrict__ grad_perturb,
                                                         float *__restrict__ grad_sigma) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;  // blockDim.x = the centres rounded up to whole waves, at most 256
  if (j >= C) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int s = 0; s < nslab; ++s) {
    const float *p = partials + (((size_t)b * nslab + s) * 4) * C + j;
    a0 += p[0];
    a1 += p[C];
    a2 += p[2 * C];
    a3 += p[3 * C];
  }
  float *gp = grad_perturb + ((size_t)b * C + j) * 3;
  gp[0] = a0;
  gp[1] = a1;
  gp[2] = a2;
  grad_sigma[(size_t)b * C + j] = a3;
}

}  // namespace hitadv

using namespace hitadv;

extern "C" int hitadv_deform_fwd(const float *ori, const float *central, const float *perturb,
                                 const float *sigma, int B, int N, int C, float *adv, float *inv_den,
                                 void *stream) {
  if (!ori || !central || !perturb || !sigma || !adv || !inv_den || B <= 0 || N <= 0 || C <= 0)
    return HITADV_E_ARG;
  dim3 grid((N + DF_PTS - 1) / DF_PTS, B);
  deform_fwd<<<grid, 256, 0, (hipStream_t)stream>>>(ori, central, perturb, sigma, N, C, adv, inv_den);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t hitadv_deform_bwd_scratch_floats(int B, int N, int C) {
  const int64_t nslab = (N + DB_PTS - 1) / DB_PTS;
  return (int64_t)B * nslab * 4 * C;
}

extern "C" int64_t hitadv_deform_bwd_slabs(int N) { return N > 0 ? (N + DB_PTS - 1) / DB_PTS : 0; }

// The first launch of hitadv_deform_bwd alone: per-slab partials [B,nslab,4,C] for a consumer that sums them itself
// (hitadv_adam_step_partials, in the reduce kernel's order).
extern "C" int hitadv_deform_bwd_partials(const float *ori, const float *central, const float *perturb,
                                          const float *sigma, const float *adv, const float *inv_den,
                                          const float *g_adv, int B, int N, int C, float *partials, void *stream) {
  if (!ori || !central || !perturb || !sigma || !adv || !inv_den || !g_adv || !partials || B <= 0 || N <= 0 || C <= 0)
    return HITADV_E_ARG;
  const int nslab = (N + DB_PTS - 1) / DB_PTS;
  const int bw = deform_bwd_width(C);
  dim3 grid((C + bw - 1) / bw, nslab, B);
  deform_bwd<<<grid, bw, 0, (hipStream_t)stream>>>(ori, central, perturb, sigma, adv, inv_den, g_adv, N, C, nslab, partials, RegGrad{}, AdamTail{});
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_deform_bwd_partials_reg(const float *ori, const float *central, const float *perturb,
                                              const float *sigma, const float *adv, const float *inv_den,
                                              const float *g_victim, const float *reg_scratch, float cd_w, int B, int N,
                                              int C, float *partials, void *stream) {
  if (!ori || !central || !perturb || !sigma || !adv || !inv_den || !g_victim || !reg_scratch || !partials || B <= 0 ||
      N <= 0 || C <= 0)
    return HITADV_E_ARG;
  const float *per_cloud = reg_scratch + (size_t)B * RG_NPART, *scal = per_cloud + (size_t)B * 8;
  const RegGrad rg{per_cloud, scal, nullptr, cd_w, 0.f, 0.f, 0.f, 0.f, B};
  const int nslab = (N + DB_PTS - 1) / DB_PTS;
  const int bw = deform_bwd_width(C);
  dim3 grid((C + bw - 1) / bw, nslab, B);
  deform_bwd<<<grid, bw, 0, (hipStream_t)stream>>>(ori, central, perturb, sigma, adv, inv_den, g_victim, N, C, nslab, partials, rg, AdamTail{});
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_deform_bwd_partials_reg_stack(int G, const float *ori, const float *central, const float *perturb,
                                                    const float *sigma, const float *adv, const float *inv_den,
                                                    const float *g_victim, const float *reg_scratch, float cd_w, int B, int N,
                                                    int C, float *partials, void *stream) {
  if (G <= 0 || !ori || !central || !perturb || !sigma || !adv || !inv_den || !g_victim || !reg_scratch || !partials ||
      B <= 0 || N <= 0 || C <= 0 || (long long)G * B > 65535)
    return HITADV_E_ARG;
  const float *per_cloud = reg_scratch + (size_t)B * RG_NPART, *scal = per_cloud + (size_t)B * 8;
  const RegGrad rg{per_cloud, scal, nullptr, cd_w, 0.f, 0.f, 0.f, 0.f, B};
  const int nslab = (N + DB_PTS - 1) / DB_PTS;
  const int bw = deform_bwd_width(C);
  dim3 grid((C + bw - 1) / bw, nslab, G * B);
  deform_bwd<<<grid, bw, 0, (hipStream_t)stream>>>(ori, central, perturb, sigma, adv, inv_den, g_victim, N, C, nslab, partials, rg, AdamTail{});
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_deform_bwd_adam_reg(const float *ori, const float *central, float *perturb, float *sigma,
                                          const float *adv, const float *inv_den, const float *g_victim,
                                          const float *hide_ref, const float *reg_scratch, float cd_w, float ker_w,
                                          float hide_w, float min_sigm, float max_sigm, float *m_perturb, float *v_perturb,
                                          float *m_sigma, float *v_sigma, int B, int N, int C, float lr_perturb,
                                          float lo_perturb, float hi_perturb, float lr_sigma, float lo_sigma, float hi_sigma,
                                          const int32_t *step, float *partials, int32_t *tickets, void *stream) {
  if (!ori || !central || !perturb || !sigma || !adv || !inv_den || !g_victim || !hide_ref || !reg_scratch || !m_perturb ||
      !v_perturb || !m_sigma || !v_sigma || !step || !partials || !tickets || B <= 0 || N <= 0 || C <= 0)
    return HITADV_E_ARG;
  const float *per_cloud = reg_scratch + (size_t)B * RG_NPART, *scal = per_cloud + (size_t)B * 8;
  const RegGrad rg_adv{per_cloud, scal, nullptr, cd_w, 0.f, 0.f, 0.f, 0.f, B};
  const RegGrad rg{per_cloud, scal, hide_ref, cd_w, ker_w, hide_w, min_sigm, 1.0f / (max_sigm - min_sigm + 1e-7f), B};
  const int nslab = (N + DB_PTS - 1) / DB_PTS;
  const AdamTail tail{tickets, AdamArgs{perturb, sigma, partials, nslab, nullptr, nullptr, m_perturb, v_perturb, m_sigma,
                                        v_sigma, B, C, lr_perturb, lo_perturb, hi_perturb, lr_sigma, lo_sigma, hi_sigma, step,
                                        rg}};
  const int bw = deform_bwd_width(C);
  dim3 grid((C + bw - 1) / bw, nslab, B);
  deform_bwd<<<grid, bw, 0, (hipStream_t)stream>>>(ori, central, perturb, sigma, adv, inv_den, g_victim, N, C, nslab, partials,
                                                    rg_adv, tail);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_deform_bwd(const float *ori, const float *central, const float *perturb,
                                 const float *sigma, const float *adv, const float *inv_den,
                                 const float *g_adv, int B, int N, int C, float *partials,
                                 float *grad_perturb, float *grad_sigma, void *stream) {
  if (!ori || !central || !perturb || !sigma || !adv || !inv_den || !g_adv || !partials || !grad_perturb ||
      !grad_sigma || B <= 0 || N <= 0 || C <= 0)
    return HITADV_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int nslab = (N + DB_PTS - 1) / DB_PTS;
  const int bw = deform_bwd_width(C);
  dim3 grid((C + bw - 1) / bw, nslab, B);
  deform_bwd<<<grid, bw, 0, s>>>(ori, central, perturb, sigma, adv, inv_den, g_adv, N, C, nslab, partials, RegGrad{}, AdamTail{});
  dim3 grid2((C + 255) / 256, B);
  deform_bwd_reduce<<<grid2, 256, 0, s>>>(partials, C, nslab, grad_perturb, grad_sigma);
  HITADV_LAUNCH_CHECK();
  return 0;
}
