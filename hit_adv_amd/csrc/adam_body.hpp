// Device-side body of the Adam step on (perturb, sigma) with the deformation's gradient still in its per-slab partials
// (csrc/iteration.hip), shared with the deformation's backward kernel, whose last block per cloud can run it as a tail.
#pragma once
#include "common.hpp"
#include "regulariser_body.hpp"

namespace hitadv {

struct AdamArgs {
  float *P, *S;
  const float *partials;  // [B,nslab,4,C]
  int nslab;
  const float *hP, *hS;   // optional second gradient terms (or nullptr)
  float *mP, *vP, *mS, *vS;
  int B, C;
  float lrP, loP, hiP, lrS, loS, hiS;
  const int32_t *step;
  RegGrad rg;             // per_cloud != nullptr: the regularisers' gradients are evaluated here instead (hP = hS = nullptr)
};

// One (cloud b, centre j).  The partials are summed in ascending slab order (= deform_bwd_reduce's order), then the update
// and the projection of adam2_k.  SC1: the partials were written by other blocks of THIS launch -- read them past the L1
// (hand-off protocol of common.hpp).
template <bool SC1>
__device__ __forceinline__ void adam_partials_body(const AdamArgs &a, const int b, const int j) {
  float *P = a.P, *S = a.S, *mP = a.mP, *vP = a.vP, *mS = a.mS, *vS = a.vS;
  const float *hP = a.hP, *hS = a.hS;
  const int C = a.C, nslab = a.nslab;
  const RegGrad &rg = a.rg;
  const int e = b * C + j;
  // the state the update needs is requested first, then the slab partials eight slabs at a time (32 loads in flight,
  // added in ascending slab order): one round trip each instead of one per slab and per state word
  const int t = *a.step;
  float m0[4], v0[4], p0[4], h0[4];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const size_t i = (size_t)e * 3 + c;
    m0[c] = mP[i]; v0[c] = vP[i]; p0[c] = P[i];
    h0[c] = (hP ? hP : P)[i];  // a pointer select: no load under a condition
  }
  m0[3] = mS[e]; v0[3] = vS[e]; p0[3] = S[e];
  h0[3] = (hS ? hS : S)[e];
  const bool reg = rg.per_cloud != nullptr;
  const float href = reg ? rg.hide_ref[e] : 0.f;
  float g[4] = {0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < nslab; s0 += 8) {
    float q[8][4];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float *p = a.partials + (((size_t)b * nslab + min(s0 + s, nslab - 1)) * 4) * C + j;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        q[s][c] = SC1 ? __hip_atomic_load(p + c * C, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p[c * C];
    }
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int c = 0; c < 4; ++c) g[c] += s0 + s < nslab ? q[s][c] : 0.f;
  }
  auto upd = [&](float *p, float *m, float *v, size_t i, int c, float gi, double lr, float lo, float hi) {
    const AdamCoef k = adam_coef(t, lr);  // csrc/arith.hpp: torch.optim.Adam's step, operation for operation
    float mi = m0[c], vi = v0[c];
    float q = adam_update(p0[c], gi, mi, vi, k);
    m[i] = mi;
    v[i] = vi;
    if (lo <= hi) q = q < lo ? lo : (q > hi ? hi : q);
    p[i] = q;
  };
  if (reg) {
#pragma unroll
    for (int c = 0; c < 3; ++c) h0[c] = reg_grad_perturb(rg, C, p0[c]);
    h0[3] = reg_grad_sigma(rg, C, b, p0[3], href);
  }
#pragma unroll
  for (int c = 0; c < 3; ++c)
    upd(P, mP, vP, (size_t)e * 3 + c, c, (hP || reg) ? g[c] + h0[c] : g[c], (double)a.lrP, a.loP, a.hiP);
  upd(S, mS, vS, (size_t)e, 3, (hS || reg) ? g[3] + h0[3] : g[3], (double)a.lrS, a.loS, a.hiS);
}

}  // namespace hitadv
