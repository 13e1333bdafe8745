// The ARITHMETIC the bit-exact kernels are made of -- the four squared-distance forms, PCT's distance and its exact tie threshold, the
// three-piece bf16 split, the two-piece fp16 split -- as plain C++ with no GPU construct in it, so that the SAME SOURCE TEXT compiles for gfx950 (through
// common.hpp) and for the host: tests/test_arith_host.py builds tests/native/arith_host.cpp with g++ -ffp-contract=off and compares every
// function here, bit for bit, with the C oracle (oracle/pointnet2_oracle.c::pair_value, itself pinned against torch in
// tests/test_oracle_gram.py) and with exact float64 / integer arithmetic.  What the GPU adds to that chain is the hardware's IEEE add / mul /
// fma / sqrt, which the -m gpu tests check on the device.  (Round 6; nothing moved here changed a kernel's instructions: tools/isa_diff.py.)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__) || defined(HITADV_EMULATED)  // (HITADV_EMULATED: the CPU wave emulator of tests/native/emu, which supplies the HIP vocabulary itself)
#define HITADV_HD __device__ __forceinline__
#else  // host build (tests only)
#include <cmath>
#include <cstring>
#define HITADV_HD static inline
static inline uint32_t __float_as_uint(float v) { uint32_t u; std::memcpy(&u, &v, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float v; std::memcpy(&v, &u, 4); return v; }
#endif

namespace hitadv {

// a = hi + mid + lo exactly, each the bf16 truncation of what is left (upper 16 bits of an fp32 = a bf16)
HITADV_HD void split3(float a, uint32_t &hi, uint32_t &mid, uint32_t &lo) {
  hi = __float_as_uint(a) & 0xffff0000u;
  const float r1 = a - __uint_as_float(hi);
  mid = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(mid);
  lo = __float_as_uint(r2);  // at most 8 significant bits are left: its lower 16 bits are zero
}

HITADV_HD uint32_t pack_hi(uint32_t even, uint32_t odd) {  // two bf16 (upper halves) -> one dword
  return (even >> 16) | (odd & 0xffff0000u);
}

// Canonical squared distance: ((dx*dx + dy*dy) + dz*dz), one fp32 rounding per operation.
// The translation unit is built with -ffp-contract=off so nothing here fuses into an FMA.
HITADV_HD float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
  float dx = ax - bx, dy = ay - by, dz = az - bz;
  return (dx * dx + dy * dy) + dz * dz;
}

// A squared distance as one of three fp32 expressions (include/hitadv.h, HITADV_FORM_*):
//   0  direct       ((dx*dx + dy*dy) + dz*dz)                                 the project's canonical rule
//   1  Gram         (|q|^2 + |p|^2) - 2 q.p, every dot product an FMA chain   _Distance.batch_pairwise_dist
//                   fma(a2,b2, fma(a1,b1, a0*b0)) -- what the GEMM behind     (util/set_distance.py:15-32)
//                   torch.bmm executes for a K = 3 inner dimension
//   2  Gram (kNN)   (|p|^2 + (-2 q.p)) + |q|^2, q.p an FMA chain, |.|^2 a     KNNDist (util/dist_utils.py:148-150)
//                   plain sum of squares ((a0*a0 + a1*a1) + a2*a2)
//   3  square_distance  ((-2 q.p) + |q|^2) + |p|^2, q.p an FMA chain, |.|^2 a    the victims' square_distance(src = q, dst = p)
//                   plain sum of squares                                      (model/pointnet2_utils.py:19-41,
//                                                                              model/pct_utils.py:40-58)
//   4  PCT get_dists  sqrt(d < 0 ? 1e-7 : d), d = (|q|^2 + |p|^2) - 2 q.p     util/other_utils.py:237-251 as PCT's sampler
//                   with q.p = fma(q1, p1, q0 p0) + q2 p2: the product of a   calls it (:254-272); only pct_dist() below
//                   ONE-row matrix does not take the GEMM kernel's chain
// q = the row / query point, p = the column / reference point.  Forms 1 to 4 reproduce the reference's values bit for
// bit (oracle/pointnet2_oracle.c::pair_value is checked against torch itself, tests/test_oracle_gram.py).
HITADV_HD float dot3_fma(float ax, float ay, float az, float bx, float by, float bz) {
  return fmaf(az, bz, fmaf(ay, by, ax * bx));
}

// form 4 (rq, rp = plain sums of squares); sqrtf is the correctly rounded one (hipcc's default for fp32 sqrt)
HITADV_HD float pct_dist(float qx, float qy, float qz, float rq, float px, float py, float pz, float rp) {
  const float zz = fmaf(qy, py, qx * px) + qz * pz;
  const float d = fmaf(-2.0f, zz, rq + rp);  // 2*zz is exact: (rq + rp) - 2*zz rounded once
  return __builtin_sqrtf(d < 0.f ? 1e-7f : d);
}

template <int FORM>
HITADV_HD float sq_norm(float x, float y, float z) {
  if (FORM == 1) return dot3_fma(x, y, z, x, y, z);
  return (x * x + y * y) + z * z;
}

template <int FORM>
HITADV_HD float pair_dist(float qx, float qy, float qz, float rq, float px, float py, float pz, float rp) {
  if (FORM == 0) return sqdist3(qx, qy, qz, px, py, pz);
  const float zz = dot3_fma(qx, qy, qz, px, py, pz);
  if (FORM == 1) return fmaf(-2.0f, zz, rq + rp);  // 2*zz is exact, so this is (rq + rp) - 2*zz rounded once
  if (FORM == 3) return fmaf(-2.0f, zz, rq) + rp;  // ((-2*zz) + rq) + rp
  return fmaf(-2.0f, zz, rp) + rq;                 // (rp + (-2*zz)) + rq
}

// Order-preserving key for non-negative floats (and +inf): the raw bit pattern.
HITADV_HD uint32_t fbits(float v) { return __float_as_uint(v); }

// the smallest float x with sqrt_rn(x) == s (s > 0 finite, the correctly rounded sqrt of some float)
HITADV_HD float sqrt_preimage_floor(float s) {
  const float sp = __uint_as_float(__float_as_uint(s) - 1u);
  const double mid = ((double)s + (double)sp) * 0.5;
  const double m2 = mid * mid;
  const float t = (float)m2;
  return (double)t < m2 ? __uint_as_float(__float_as_uint(t) + 1u) : t;
}

// torch.optim.Adam's single-tensor step (torch/optim/adam.py::_single_tensor_adam, the CPU path the reference's optimiser takes:
// ShapeAttack/HiT_ADV.py:139-145, CW/*.py), operation for operation in fp32 with the bias corrections in double as Python computes them:
//   exp_avg.lerp_(grad, 1 - beta1)            m + (g - m) * (1 - beta1)
//   exp_avg_sq.mul_(beta2).addcmul_(g, g, value = 1 - beta2)      v * beta2 + ((1 - beta2) * g) * g
//   denom = exp_avg_sq.sqrt() / sqrt(bias_correction2) + eps;  param.addcdiv_(exp_avg, denom, value = -lr / bias_correction1)
// betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad.  tests/test_arith_host.py steps this source beside torch.optim.Adam.
struct AdamCoef {
  float step_size, bc2_sqrt;
};
HITADV_HD AdamCoef adam_coef(int t, double lr) {  // t = the 1-based step number
  const double beta1 = 0.9, beta2 = 0.999;
  const double bc1 = 1.0 - pow(beta1, (double)t);
  const double bc2 = 1.0 - pow(beta2, (double)t);
  AdamCoef k;
  k.step_size = (float)(lr / bc1);
  k.bc2_sqrt = (float)sqrt(bc2);
  return k;
}
// one coordinate: moments updated in place, the new parameter value returned (the caller projects and stores it)
HITADV_HD float adam_update(float p, float gi, float &m, float &v, const AdamCoef &k) {
  const double beta1 = 0.9, beta2 = 0.999, eps = 1e-8;
  const float mi = m + (gi - m) * (float)(1.0 - beta1);
  const float vi = v * (float)beta2 + ((float)(1.0 - beta2) * gi) * gi;
  m = mi;
  v = vi;
  const float denom = __builtin_sqrtf(vi) / k.bc2_sqrt + (float)eps;
  return p - (k.step_size * mi) / denom;
}

#if defined(__HIPCC__) || defined(__HIP__) || defined(__clang__)  // (_Float16: hipcc, and clang++ on the host; g++ 11 has no such type)
// An fp32 value as TWO fp16 pieces, a = hi + 2^-11 lo + r (csrc/victim_bf3.hip's header has the error analysis): hi = fp16(a) to nearest,
// lo = fp16(2^11 (a - hi)).  Written as ONE fused multiply-add on the converted-back hi piece: 2048 v is exact, hi (-2048) + 2048 v =
// 2048 (v - hi) is exact before its single rounding to fp16 -- the same bits as converting hi back, subtracting, scaling and converting.
constexpr float F16X2_PIECE_SCALE = 2048.f;  // 2^11
HITADV_HD void split_pair(float v, _Float16 &hi, _Float16 &lo) {
  hi = (_Float16)v;
  lo = (_Float16)__builtin_fmaf((float)hi, -F16X2_PIECE_SCALE, v * F16X2_PIECE_SCALE);
}
#endif

}  // namespace hitadv
