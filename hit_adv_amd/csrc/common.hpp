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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, HITADV_WAVE);
  return v;
}

}  // namespace hitadv
