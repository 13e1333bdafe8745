// The three distance regularisers of the HiT-ADV loss and their gradients, fused
// (ShapeAttack/HiT_ADV.py:229-245):
//   dist = cd_w * mean_b Q1_b  +  ker_w * (|P|_F + |1 - sigma|_F) / C  +  hide_w * mean_b cos_b
//   Q1_b  = ChamferDist('adv2ori') applied to the [3,N] tensors as the reference does (quirk Q1): the mean
//           over the 3 coordinate rows i of adv of  min_j |adv_row_i - ori_row_j|^2   (HiT_ADV.py:230)
//   cos_b = cosine_similarity(hide_ref_b, (sigma_b - min)/(max - min + 1e-7))          (HiT_ADV.py:341-346)
// and the loss term  mean_b(scale_const_b * dist) = mean(scale_const) * dist          (HiT_ADV.py:243-245).
// Forward = per-cloud partial sums (one block per cloud) + a one-block finalise; backward = one
// elementwise kernel.  Replaces ~45 torch launches per iteration; all sums in fixed order.
#include "common.hpp"
#include "hitadv.h"
#include "regulariser_body.hpp"

namespace hitadv {

__global__ __launch_bounds__(256) void reg_partials(const float *__restrict__ P, const float *__restrict__ sigma,
                                                    const float *__restrict__ adv, const float *__restrict__ ori,
                                                    const float *__restrict__ hide_ref, int N, int C, float min_s,
                                                    float inv_range, float *part, RegFin fin, int fused) {
  reg_partials_body(P, sigma, adv, ori, hide_ref, N, C, min_s, inv_range, part, fin, fused, blockIdx.x);
}

// One block: combine the clouds.  per_cloud[b] = {cos_b, inv(|r||n|), inv(|n|^2), arg0, arg1, arg2}
__global__ __launch_bounds__(256) void reg_finalise(const float *__restrict__ part, const float *__restrict__ scale_const,
                                                    int B, int C, float cd_w, float ker_w, float hide_w,
                                                    float *__restrict__ per_cloud, float *__restrict__ scal,
                                                    float *__restrict__ dist_out, float *__restrict__ scaled_out) {
  __shared__ float sm[4];
  reg_finalise_body(part, scale_const, B, C, cd_w, ker_w, hide_w, per_cloud, scal, dist_out, scaled_out, sm);
}

__global__ __launch_bounds__(256) void reg_backward(const float *__restrict__ P, const float *__restrict__ sigma,
                                                    const float *__restrict__ adv, const float *__restrict__ ori,
                                                    const float *__restrict__ hide_ref,
                                                    const float *__restrict__ per_cloud, const float *__restrict__ scal,
                                                    const float *__restrict__ go, int B, int N, int C, float cd_w,
                                                    float ker_w, float hide_w, float min_s, float inv_range,
                                                    float *__restrict__ gP, float *__restrict__ gS,
                                                    float *__restrict__ gA, const float *__restrict__ addA) {
  const long long nP = (long long)B * C * 3, nS = (long long)B * C, nA = (long long)B * 3 * N;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  const float k = (go ? go[0] : 1.0f) * scal[0];  // upstream * mean(scale_const)
  if (e < nP) {
    gP[e] = ker_w != 0.f ? k * (ker_w / (float)C) * scal[1] * P[e] : 0.f;
  } else if (e < nP + nS) {
    const long long i = e - nP;
    const int b = (int)(i / C);
    const float s = sigma[i];
    float g = 0.f;
    if (ker_w != 0.f) g += (ker_w / (float)C) * scal[2] * (s - 1.0f);
    if (hide_w != 0.f) {
      const float *pc = per_cloud + (size_t)b * 8;
      const float n = (s - min_s) * inv_range;
      g += (hide_w / (float)B) * inv_range * (hide_ref[i] * pc[1] - pc[0] * n * pc[2]);
    }
    gS[i] = k * g;
  } else if (e < nP + nS + nA) {
    const long long i = e - nP - nS;
    float g = 0.f;
    if (cd_w != 0.f) {
      const int b = (int)(i / (3LL * N));
      const int row = (int)((i / N) % 3);
      const int n = (int)(i % N);
      const int arg = (int)per_cloud[(size_t)b * 8 + 3 + row];
      g = k * (cd_w / (3.0f * (float)B)) * 2.0f * (adv[i] - ori[((size_t)b * 3 + arg) * N + n]);
    }
    gA[i] = addA ? g + addA[i] : g;
  }
}

}  // namespace hitadv

using namespace hitadv;

extern "C" int64_t hitadv_regulariser_scratch_floats(int B) { return (int64_t)B * (RG_NPART + 8) + RG_NSCAL; }

extern "C" int hitadv_regulariser_fwd(const float *perturb, const float *sigma, const float *adv, const float *ori,
                                      const float *hide_ref, const float *scale_const, int B, int N, int C,
                                      float cd_w, float ker_w, float hide_w, float min_sigm, float max_sigm,
                                      float *scratch, float *dist_loss, float *scaled_loss, void *stream) {
  if (!perturb || !sigma || !adv || !ori || !hide_ref || !scale_const || !scratch || !dist_loss || !scaled_loss ||
      B <= 0 || N <= 0 || C <= 0)
    return HITADV_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  float *part = scratch, *per_cloud = scratch + (size_t)B * RG_NPART, *scal = per_cloud + (size_t)B * 8;
  const float inv_range = 1.0f / (max_sigm - min_sigm + 1e-7f);
  reg_partials<<<B, 256, 0, s>>>(perturb, sigma, adv, ori, hide_ref, N, C, min_sigm, inv_range, part, RegFin{}, 0);
  reg_finalise<<<1, 256, 0, s>>>(part, scale_const, B, C, cd_w, ker_w, hide_w, per_cloud, scal, dist_loss, scaled_loss);
  HITADV_LAUNCH_CHECK();
  return 0;
}

// The same forward in ONE launch: the last block of the per-cloud pass combines the clouds.  The scratch's last float
// (the ticket) must be ZERO when this is first called on a scratch buffer; every call leaves it at zero.
extern "C" int hitadv_regulariser_fwd_fused(const float *perturb, const float *sigma, const float *adv, const float *ori,
                                            const float *hide_ref, const float *scale_const, int B, int N, int C,
                                            float cd_w, float ker_w, float hide_w, float min_sigm, float max_sigm,
                                            float *scratch, float *dist_loss, float *scaled_loss, void *stream) {
  if (!perturb || !sigma || !adv || !ori || !hide_ref || !scale_const || !scratch || !dist_loss || !scaled_loss ||
      B <= 0 || N <= 0 || C <= 0)
    return HITADV_E_ARG;
  float *part = scratch, *per_cloud = scratch + (size_t)B * RG_NPART, *scal = per_cloud + (size_t)B * 8;
  const float inv_range = 1.0f / (max_sigm - min_sigm + 1e-7f);
  RegFin fin{scale_const, B, cd_w, ker_w, hide_w, per_cloud, scal, dist_loss, scaled_loss};
  reg_partials<<<B, 256, 0, (hipStream_t)stream>>>(perturb, sigma, adv, ori, hide_ref, N, C, min_sigm, inv_range, part, fin, 1);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_regulariser_bwd(const float *perturb, const float *sigma, const float *adv, const float *ori,
                                      const float *hide_ref, const float *scratch, const float *grad_out, int B, int N,
                                      int C, float cd_w, float ker_w, float hide_w, float min_sigm, float max_sigm,
                                      float *grad_perturb, float *grad_sigma, float *grad_adv, void *stream) {
  if (!perturb || !sigma || !adv || !ori || !hide_ref || !scratch || !grad_out || !grad_perturb || !grad_sigma ||
      !grad_adv || B <= 0 || N <= 0 || C <= 0)
    return HITADV_E_ARG;
  const float *per_cloud = scratch + (size_t)B * RG_NPART, *scal = per_cloud + (size_t)B * 8;
  const float inv_range = 1.0f / (max_sigm - min_sigm + 1e-7f);
  const long long total = (long long)B * C * 3 + (long long)B * C + (long long)B * 3 * N;
  reg_backward<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      perturb, sigma, adv, ori, hide_ref, per_cloud, scal, grad_out, B, N, C, cd_w, ker_w, hide_w, min_sigm, inv_range,
      grad_perturb, grad_sigma, grad_adv, nullptr);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_regulariser_bwd_add(const float *perturb, const float *sigma, const float *adv, const float *ori,
                                          const float *hide_ref, const float *scratch, const float *grad_out,
                                          const float *add_adv, int B, int N, int C, float cd_w, float ker_w,
                                          float hide_w, float min_sigm, float max_sigm, float *grad_perturb,
                                          float *grad_sigma, float *grad_adv, void *stream) {
  if (!perturb || !sigma || !adv || !ori || !hide_ref || !scratch || !grad_perturb || !grad_sigma || !grad_adv ||
      B <= 0 || N <= 0 || C <= 0)
    return HITADV_E_ARG;
  const float *per_cloud = scratch + (size_t)B * RG_NPART, *scal = per_cloud + (size_t)B * 8;
  const float inv_range = 1.0f / (max_sigm - min_sigm + 1e-7f);
  const long long total = (long long)B * C * 3 + (long long)B * C + (long long)B * 3 * N;
  reg_backward<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      perturb, sigma, adv, ori, hide_ref, per_cloud, scal, grad_out, B, N, C, cd_w, ker_w, hide_w, min_sigm, inv_range,
      grad_perturb, grad_sigma, grad_adv, add_adv);
  HITADV_LAUNCH_CHECK();
  return 0;
}
