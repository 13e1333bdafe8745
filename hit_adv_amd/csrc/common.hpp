// Shared device helpers for libhitadv_hip (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "arith.hpp"  // the distance forms, the three-piece split, PCT's tie threshold: plain C++, also compiled for the HOST by the CPU tests

#define HITADV_WAVE 64

// Marks a place where the lanes of ONE wave hand data to each other through LDS without a block barrier: the region is read by this
// wave only, a wave's LDS operations are performed in program order, and its lanes execute in lockstep -- so nothing is needed on the
// GPU and the macro is empty.  (The CPU wave emulator of tests/native/emu, whose lanes are fibres, defines it as a wave rendezvous.)
#ifndef HITADV_WAVE_LDS_HANDOFF
#define HITADV_WAVE_LDS_HANDOFF() ((void)0)
#endif

#define HITADV_LAUNCH_CHECK()                       \
  do {                                              \
    hipError_t e__ = hipGetLastError();             \
    if (e__ != hipSuccess) return (int)e__;         \
  } while (0)

// ---------------------------------------------------------------------------------------------------------------
// Hand-off of split partials to the last block to arrive (fc_layer_k, linear_max_fwd_k, linear_max_fwd_bf3_k).
//
// gfx950 form (default): every partial is stored write-through (relaxed agent-scope atomic store = `sc1`), every storing
// wave drains its stores (`s_waitcnt vmcnt(0)`), the workgroup meets at a barrier, ONE lane draws the ticket with a
// relaxed agent-scope fetch_add, the other waves learn the outcome behind a second barrier, and the last workgroup reads
// every partial with `sc1` loads (relaxed agent-scope atomic loads, which bypass the per-CU L1).  This is the first row
// of the "Valid forms ... with sc1 loads in place of the acquire" table of MI355X_MICROARCH.md (measured on gfx950 /
// ROCm 7.2, not an architectural guarantee); an agent-scope release here costs a whole-L2 write-back per publishing
// block (34 us instead of 6.8 us for a 256-block layer).  tests/test_gpu_handoff.py stresses it under uneven load on
// three streams.
//
// Portable form (-DHITADV_PORTABLE_HANDOFF): agent-scope RELEASE fence before the ticket, agent-scope ACQUIRE fence in the
// last block before it reads -- what the HIP memory model asks for.  Any target other than gfx950 must use it.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(HITADV_PORTABLE_HANDOFF)
#error "the fence-free split hand-off is validated on gfx950 only: build other targets with -DHITADV_PORTABLE_HANDOFF"
#endif
#ifdef HITADV_PORTABLE_HANDOFF
#define HITADV_HANDOFF_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#define HITADV_HANDOFF_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#else
#define HITADV_HANDOFF_RELEASE() ((void)0)
#define HITADV_HANDOFF_ACQUIRE() ((void)0)
#endif

// ---------------------------------------------------------------------------------------------------------------
// Diagnostic build only (make ablate -> tools/build/libhitadv_hip_ablate.so, loaded through HITADV_LIBRARY): entry points of
// the kernel families named in the environment variable HITADV_ABLATE ("fc", "v1", "v2", "v3", comma separated) return
// without launching, so that a timing run shows what a family costs the LOOP (its share of throughput, which is not its
// share of summed kernel time when several attacks overlap).  Results of such a run are garbage by construction; the
// product library contains none of this.
#ifdef HITADV_ABLATE
#include <cstdlib>
#include <cstring>
static inline bool hitadv_ablated(const char *tag) {
  const char *e = getenv("HITADV_ABLATE");
  if (!e) return false;
  const size_t n = strlen(tag);
  for (const char *p = e; (p = strstr(p, tag)) != nullptr; p += n)
    if ((p == e || p[-1] == ',') && (p[n] == 0 || p[n] == ',')) return true;
  return false;
}
#define HITADV_ABLATE_RETURN(tag)                    \
  do {                                               \
    static const bool off__ = hitadv_ablated(tag);   \
    if (off__) return 0;                             \
  } while (0)
#else
#define HITADV_ABLATE_RETURN(tag) ((void)0)
#endif

namespace hitadv {

// A kernel's dynamic-LDS limit, raised once PER DEVICE (the attribute belongs to the function on one device; a process
// may drive several GPUs), and the error of a refusal handed back to the entry point instead of surfacing at the launch.
struct LdsRaised {
  unsigned long long done = 0;  // bit d: raised on device d
};
static inline hipError_t raise_dynamic_lds(LdsRaised &st, const void *kernel, int bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (st.done & bit) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) st.done |= bit;
  return e;
}
// in a function that returns the entry point's int code; `kernel_ptr` in parentheses when its template arguments hold commas
#define HITADV_RAISE_LDS(kernel_ptr, bytes)                                                                            \
  do {                                                                                                                 \
    static hitadv::LdsRaised st__;                                                                                     \
    const hipError_t e__ = hitadv::raise_dynamic_lds(st__, reinterpret_cast<const void *>(kernel_ptr), (int)(bytes));   \
    if (e__ != hipSuccess) return (int)e__;                                                                            \
  } while (0)

// An index table that leaves a kernel holds VALID rows whatever the input was.  The selection kernels start their lists
// at a sentinel (0x7fffffff / -1) that a comparison with NaN never replaces: a cloud that went NaN (a diverged attack, an
// fp16-range overflow that the caller only notices when it reads its results back) would otherwise hand the sentinel to the
// gather of the backward pass -- a wild read, on a bad day a memory access fault that takes the process down instead of the
// range flag's clean re-run (round 5: PCT x3 weights under CW.attack_concurrently).  The values that belong to such an
// entry stay NaN / inf, which is what the caller can see.
__device__ __forceinline__ int sane_index(int i, int limit, int fallback) {
  return (unsigned)i < (unsigned)limit ? i : min(fallback, limit - 1);
}

// Drain this wave's stores, meet the workgroup, draw the ticket of `slot`; true in every thread of the workgroup whose
// ticket was the last of `total`.  `flag` is a __shared__ int of the caller.
__device__ __forceinline__ bool handoff_last_arriver(int *tickets, int slot, int total, int *flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-through stores are acknowledged ...
  __syncthreads();                                   // ... for every wave of the block, before its ticket is drawn
  if (threadIdx.x == 0) {
    HITADV_HANDOFF_RELEASE();
    *flag = __hip_atomic_fetch_add(&tickets[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1;
  }
  __syncthreads();
  const bool last = *flag != 0;
  if (last) HITADV_HANDOFF_ACQUIRE();
  return last;
}

// ---- fp16 range watch of the two-piece (fp16x2) kernels: the largest magnitude seen, as raw bits with the sign cleared.
// Unsigned order on those bits is the order of the magnitudes, +inf sorts above every finite value and every NaN above
// +inf -- so a NaN operand raises the flag too (fmaxf(big, fabsf(v)) silently drops it).
struct RangeWatch {
  uint32_t m = 0u;
  __device__ __forceinline__ void see(float v) {
    const uint32_t b = __float_as_uint(v) & 0x7fffffffu;
    m = b > m ? b : m;
  }
  // `relu_of_pre` = max(pre, 0) as the kernels write it (which maps a NaN to 0): the NaN is caught on `pre` itself
  __device__ __forceinline__ void see_relu(float pre, float relu_of_pre) {
    const uint32_t b = __float_as_uint(relu_of_pre);  // >= +0: no sign to clear
    m = b > m ? b : m;
    nan_ = fmaf(pre, 0.f, nan_);                       // NaN (or an infinity) in -> NaN
  }
  __device__ __forceinline__ bool beyond_fp16() const { return m >= 0x477fe000u || nan_ != nan_; }  // |v| >= 65504, inf, NaN
  float nan_ = 0.f;
};

// The same watch on the fp16 hi pieces of a two-piece split, two per word: the split (and with it the product) breaks down
// exactly when a hi piece is an infinity or a NaN, i.e. |a| >= 65520 or a not finite.  One AND and one packed 16-bit maximum
// per PAIR of values.
struct PieceWatch {
  uint32_t m = 0u;
  __device__ __forceinline__ void see_f16x2(uint32_t hi_pair) {
    const uint32_t b = hi_pair & 0x7fff7fffu;
    uint32_t r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(b), "v"(m));
    m = r;
  }
  __device__ __forceinline__ bool beyond_fp16() const { return (m & 0xffffu) >= 0x7c00u || (m >> 16) >= 0x7c00u; }
};

// ---- an fp32 value as three bf16 pieces (csrc/victim_bf3.hip, csrc/knn.hip: fp32-accurate products on the bf16 MFMAs)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 as_bf16x8(uint4 u) { return __builtin_bit_cast(bf16x8, u); }

// eight consecutive floats -> their three pieces, eight bf16 (one uint4) each
__device__ __forceinline__ void split3x8(const float4 &lo4, const float4 &hi4, uint4 &p1, uint4 &p2, uint4 &p3) {
  const float a[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
  uint32_t h[8], m[8], l[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) split3(a[i], h[i], m[i], l[i]);
  p1 = make_uint4(pack_hi(h[0], h[1]), pack_hi(h[2], h[3]), pack_hi(h[4], h[5]), pack_hi(h[6], h[7]));
  p2 = make_uint4(pack_hi(m[0], m[1]), pack_hi(m[2], m[3]), pack_hi(m[4], m[5]), pack_hi(m[6], m[7]));
  p3 = make_uint4(pack_hi(l[0], l[1]), pack_hi(l[2], l[3]), pack_hi(l[4], l[5]), pack_hi(l[6], l[7]));
}

// sqrt(d), correctly rounded, for d = 0 or 2^-96 <= d < 2^96 (squared distances of clouds in the unit ball: never subnormal,
// never huge).  hipcc's own lowering of __builtin_sqrtf is this very fix-up of v_sqrt_f32 (the neighbours one ulp down / up,
// two FMAs for the signs of the residuals, two selects) wrapped in a scaling branch for d < 2^-96 and a class test for 0 / inf:
// five more vector instructions per value of the deformation's 38 (round 5: the deformation kernels are bound by their vector
// instructions -- 32.9 M per deform_bwd launch = 100 % of its duration, profiles/r05_loop_traffic.json).  Same bits on the
// stated range; d = 0 gives 0 (the down-neighbour of 0 is a NaN pattern whose comparisons fail).
// OUTSIDE the range (0 < d < 2^-96: two distinct points within 3.5e-15 of each other, i.e. every coordinate of both below ~1e-14)
// the residuals underflow and the result may be one ulp off __builtin_sqrtf, or flushed for a subnormal d: a distance < 2^-48 that
// the deformation only uses as exp2(r * a2) (= 1 to the last bit for sigma >= 3e-4; min_sigm is 0.1) and as the factor of one term
// < 2^-48 |dk| of the sigma gradient -- below the last bit of either sum.  The deformation's parity bar is a tolerance (1e-5
// relative), not bit equality; tests/test_z_r06_edges.py::test_deform_near_duplicate_points_at_the_origin_...
// holds it on such a cloud.  A wave-uniform fall-back would cost a compare, a ballot and a branch per value of a kernel bound by
// its 38 vector instructions per value.
__device__ __forceinline__ float sqrt_rn_ranged(float d) {
  float s = __builtin_amdgcn_sqrtf(d);
  const float dn = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
  const float vp = fmaf(-dn, s, d), vs = fmaf(-up, s, d);
  s = vp <= 0.f ? dn : s;
  s = vs > 0.f ? up : s;
  return s;
}
// 2^x for x <= 0 with results below 2^-126 flushed to zero (v_exp_f32 alone: exp2f() scales such arguments to return
// subnormals, four more instructions).  For sums that are not divided by: a term of 1e-38 is not seen by an fp32 sum.
__device__ __forceinline__ float exp2_flush(float x) { return __builtin_amdgcn_exp2f(x); }

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
