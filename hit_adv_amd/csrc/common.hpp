// Shared device helpers for libhitadv_hip (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HITADV_WAVE 64

#define HITADV_LAUNCH_CHECK()                       \
  do {                                              \
    hipError_t e__ = hipGetLastError();             \
    if (e__ != hipSuccess) return (int)e__;         \
  } while (0)

namespace hitadv {

// Canonical squared distance: ((dx*dx + dy*dy) + dz*dz), one fp32 rounding per operation.
// The translation unit is built with -ffp-contract=off so nothing here fuses into an FMA.
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
  float dx = ax - bx, dy = ay - by, dz = az - bz;
  return (dx * dx + dy * dy) + dz * dz;
}

// Order-preserving key for non-negative floats (and +inf): the raw bit pattern.
__device__ __forceinline__ uint32_t fbits(float v) { return __float_as_uint(v); }

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  lo = __shfl_xor(lo, mask, HITADV_WAVE);
  hi = __shfl_xor(hi, mask, HITADV_WAVE);
  return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    unsigned long long o = shfl_xor_u64(v, m);
    v = o > v ? o : v;
  }
  return v;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    unsigned long long o = shfl_xor_u64(v, m);
    v = o < v ? o : v;
  }
  return v;
}

// Wave-wide unsigned max in 6 DPP steps (row_shr 1/2/4/8 inside each 16-lane row, then row_bcast:15 and
// row_bcast:31 across rows) instead of 6 ds_bpermute round trips; the result is read from lane 63 and is
// wave-uniform.  Lanes that receive nothing from a shift see the identity 0.
__device__ __forceinline__ uint32_t wave_max_u32_dpp(uint32_t v) {
  uint32_t t;
  t = __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false); v = t > v ? t : v;  // row_shr:1
  t = __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false); v = t > v ? t : v;  // row_shr:2
  t = __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, false); v = t > v ? t : v;  // row_shr:4
  t = __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, false); v = t > v ? t : v;  // row_shr:8
  t = __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false); v = t > v ? t : v;  // row_bcast:15 -> rows 1,3
  t = __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false); v = t > v ? t : v;  // row_bcast:31 -> rows 2,3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Lexicographic max of (hi, lo) pairs across the wave: max hi first, then max lo among its holders.
__device__ __forceinline__ unsigned long long wave_max_u64_dpp(unsigned long long key) {
  const uint32_t hi = (uint32_t)(key >> 32), lo = (uint32_t)key;
  const uint32_t mh = wave_max_u32_dpp(hi);
  const uint32_t ml = wave_max_u32_dpp(hi == mh ? lo : 0u);
  return ((unsigned long long)mh << 32) | ml;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, HITADV_WAVE);
  return v;
}

}  // namespace hitadv
