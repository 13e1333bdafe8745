// A wave64 SIMT emulator for the CPU -- TEST INFRASTRUCTURE, never part of the product (tests/test_emulated_kernels.py).
//
// tests/native/emu_build.py takes a kernel file of hit_adv_amd/csrc/ AS IT IS, rewrites the three things plain C++ cannot parse (the
// `kernel<<<grid, block, lds, stream>>>(args)` launch syntax, `extern __shared__` arrays, one LDS atomic written as inline asm) and
// compiles it against this header instead of <hip/hip_runtime.h>: the kernels' bodies, their launchers and their extern "C" entry points
// are the product's text.  A launch runs its blocks one after the other; the threads of a block are fibres (ucontext) on one OS thread,
// scheduled round-robin and switched only at __syncthreads and at the wave-level operations (__shfl*, __ballot, DPP, readlane, ...),
// which are rendezvous points of the 64 lanes of a wave: every live lane deposits its operand, the last to arrive publishes the
// snapshot, every lane reads what the hardware would have given it.  Threads that have returned count as inactive lanes (EXEC = 0).
// What this checks: the kernels' LOGIC -- index arithmetic, tie rules, reductions, barriers placement, LDS protocols -- bit for bit against
// the oracle, on the CPU.  What it cannot check: the hardware (IEEE rounding of the real instructions, memory ordering between CUs,
// timing); the -m gpu tests are for that.  fp32 arithmetic here is the host's IEEE arithmetic with -ffp-contract=off, fmaf = one rounding.
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

// ---------------------------------------------------------------------------------------------------------------- HIP vocabulary
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define HITADV_EMULATED 1
#define HITADV_WAVE_LDS_HANDOFF() ((void)emu::wave_gather(0, nullptr))  // lanes are fibres here: a wave-private LDS hand-off needs a rendezvous

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float a, float b) { return {a, b}; }
static inline float3 make_float3(float a, float b, float c) { return {a, b, c}; }
static inline float4 make_float4(float a, float b, float c, float d) { return {a, b, c, d}; }
static inline int2 make_int2(int a, int b) { return {a, b}; }
static inline int4 make_int4(int a, int b, int c, int d) { return {a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return {a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return {a, b, c, d}; }

typedef void *hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void *, int, int) { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
enum { hipMemcpyDeviceToDevice = 3 };
#define HIP_SYMBOL(x) (&(x))
static inline hipError_t hipMemcpyFromSymbol(void *d, const void *sym, size_t n) { memcpy(d, sym, n); return hipSuccess; }

namespace emu {

struct Thread {
  ucontext_t ctx;
  std::vector<char> stack;
  dim3 tid;
  int lin = 0;
  bool done = false;
};
struct Wave {
  uint64_t slots[64], snap[2][64], snapmask[2] = {0, 0};
  uint64_t arrived = 0, alive = 0;
  unsigned gen = 0;
};
struct Block {
  std::vector<Thread> threads;
  std::vector<Wave> waves;
  int alive = 0, at_barrier = 0;
  unsigned barrier_gen = 0;
  std::function<void()> body;
};

inline Thread *cur = nullptr;
inline Block *blk = nullptr;
inline ucontext_t sched_ctx;
inline dim3 g_blockIdx, g_blockDim, g_gridDim;
inline std::vector<char> dyn_lds_store;
inline char *dyn_lds = nullptr;
inline unsigned long progress = 0;  // bumped whenever a rendezvous completes or a thread ends: the scheduler's deadlock detector

inline void yield() { swapcontext(&cur->ctx, &sched_ctx); }

inline void wave_release(Wave &w) {
  for (int l = 0; l < 64; ++l) w.snap[w.gen & 1][l] = (w.arrived >> l) & 1 ? w.slots[l] : 0;
  w.snapmask[w.gen & 1] = w.arrived;
  w.arrived = 0;
  ++w.gen;
  ++progress;
}
// every live lane of the calling wave deposits v; returns the snapshot (values of all 64 lanes, 0 for inactive ones) and the active mask
inline const uint64_t *wave_gather(uint64_t v, uint64_t *active) {
  Wave &w = blk->waves[cur->lin >> 6];
  const int lane = cur->lin & 63;
  const unsigned g = w.gen;
  w.slots[lane] = v;
  w.arrived |= 1ull << lane;
  if (w.arrived == w.alive) wave_release(w);
  while (w.gen == g) yield();
  if (active) *active = w.snapmask[g & 1];
  return w.snap[g & 1];
}
// the same rendezvous for a payload of up to 32 bytes per lane (the operands of a matrix instruction); returns the 64 lanes' payloads
struct WaveBulk { unsigned char slots[64][32], snap[2][64][32]; };
inline std::vector<WaveBulk> *bulk_store = nullptr;
inline const unsigned char (*wave_gather_bytes(const void *p, size_t n))[32] {
  static std::vector<WaveBulk> store;
  Wave &w = blk->waves[cur->lin >> 6];
  const size_t wi = (size_t)(cur->lin >> 6);
  if (store.size() <= wi) store.resize(wi + 1);
  const int lane = cur->lin & 63;
  memcpy(store[wi].slots[lane], p, n);
  const unsigned g = w.gen;
  w.slots[lane] = 0;
  w.arrived |= 1ull << lane;
  if (w.arrived == w.alive) {
    memcpy(store[wi].snap[g & 1], store[wi].slots, sizeof(store[wi].slots));
    wave_release(w);
  }
  while (w.gen == g) yield();
  return store[wi].snap[g & 1];
}

inline void barrier() {
  const unsigned g = blk->barrier_gen;
  if (++blk->at_barrier == blk->alive) {
    blk->at_barrier = 0;
    ++blk->barrier_gen;
    ++progress;
  }
  while (blk->barrier_gen == g) yield();
}
inline void thread_exit_hook() {  // a returning thread leaves its wave's and its block's rendezvous
  Wave &w = blk->waves[cur->lin >> 6];
  w.alive &= ~(1ull << (cur->lin & 63));
  if (w.arrived && w.arrived == w.alive) wave_release(w);
  --blk->alive;
  if (blk->alive > 0 && blk->at_barrier == blk->alive) {
    blk->at_barrier = 0;
    ++blk->barrier_gen;
  }
  ++progress;
}
inline void trampoline() {
  blk->body();
  cur->done = true;
  thread_exit_hook();
  swapcontext(&cur->ctx, &sched_ctx);
}

inline void launch(dim3 grid, dim3 block, size_t shm, std::function<void()> body) {
  const int nthreads = (int)(block.x * block.y * block.z);
  if (dyn_lds_store.size() < shm + 64) dyn_lds_store.resize(shm + 64);
  dyn_lds = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(dyn_lds_store.data()) + 63) & ~uintptr_t(63));
  g_blockDim = block;
  g_gridDim = grid;
  Block b;
  b.body = body;
  b.threads.resize(nthreads);
  for (auto &t : b.threads) t.stack.resize(256 * 1024);
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_blockIdx = dim3(bx, by, bz);
        blk = &b;
        b.waves.assign((nthreads + 63) / 64, Wave());
        b.alive = nthreads;
        b.at_barrier = 0;
        for (int i = 0; i < nthreads; ++i) {
          Thread &t = b.threads[i];
          t.lin = i;
          t.done = false;
          t.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
          b.waves[i >> 6].alive |= 1ull << (i & 63);
          getcontext(&t.ctx);
          t.ctx.uc_stack.ss_sp = t.stack.data();
          t.ctx.uc_stack.ss_size = t.stack.size();
          t.ctx.uc_link = &sched_ctx;
          makecontext(&t.ctx, (void (*)())trampoline, 0);
        }
        // the order in which the scheduler visits the fibres.  The hardware promises NO order between waves, so results must not depend
        // on it: HITADV_EMU_SCHEDULE=desc runs the waves of a block last-to-first, =random:<seed> in a shuffled order that changes every
        // round (lanes inside a wave stay ascending: a wave is in lockstep on the GPU).  A kernel that passes ascending and fails otherwise
        // has a missing barrier between waves -- a race the GPU's timing may be hiding (tests/test_emulated_kernels.py runs all three).
        static const char *sched_env = getenv("HITADV_EMU_SCHEDULE");
        const int nw = (nthreads + 63) / 64;
        std::vector<int> worder(nw);
        for (int w = 0; w < nw; ++w) worder[w] = sched_env && sched_env[0] == 'd' ? nw - 1 - w : w;
        unsigned rng = sched_env && sched_env[0] == 'r' ? (unsigned)atoi(sched_env + (strchr(sched_env, ':') ? strchr(sched_env, ':') - sched_env + 1 : 0)) * 2654435761u + 12345u + bx * 97u + by * 31u : 0u;
        int left = nthreads;
        while (left > 0) {
          const unsigned long before = progress;
          int ran = 0;
          if (rng) {
            for (int w = nw - 1; w > 0; --w) {
              rng = rng * 1664525u + 1013904223u;
              std::swap(worder[w], worder[(rng >> 8) % (unsigned)(w + 1)]);
            }
          }
          for (int oi = 0; oi < nthreads; ++oi) {
            const int i = worder[oi >> 6] * 64 + (oi & 63);
            if (i >= nthreads) continue;
            Thread &t = b.threads[i];
            if (t.done) continue;
            cur = &t;
            swapcontext(&sched_ctx, &t.ctx);
            ++ran;
            if (t.done) --left;
          }
          if (left > 0 && progress == before) {
            fprintf(stderr, "simt_emu: deadlock in block (%u,%u,%u): %d threads wait at a rendezvous that cannot complete (divergent collective?)\n", bx, by, bz, left);
            abort();
          }
        }
      }
  blk = nullptr;
  cur = nullptr;
}

}  // namespace emu

#define threadIdx (emu::cur->tid)
#define blockIdx (emu::g_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)
#define warpSize 64

static inline void __syncthreads() { emu::barrier(); }

// ---------------------------------------------------------------------------------------------------------------- bit helpers
static inline uint32_t __float_as_uint(float v) { uint32_t u; memcpy(&u, &v, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float v; memcpy(&v, &u, 4); return v; }
static inline int __float_as_int(float v) { int u; memcpy(&u, &v, 4); return u; }
static inline float __int_as_float(int u) { float v; memcpy(&v, &u, 4); return v; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline uint32_t __brev(uint32_t v) {
  v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
  v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
  v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
  return __builtin_bswap32(v);
}
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
#define __expf(x) expf(x)  // (glibc declares these names itself; the index kernels under emulation do not use them)
#define __logf(x) logf(x)
#define __sinf(x) sinf(x)
#define __cosf(x) cosf(x)
template <class T> static inline T min(T a, T b) { return b < a ? b : a; }
template <class T> static inline T max(T a, T b) { return a < b ? b : a; }
static inline long long min(long long a, int b) { return a < b ? a : b; }
static inline long long min(int a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, int b) { return a < (unsigned)b ? a : (unsigned)b; }
static inline size_t min(size_t a, int b) { return a < (size_t)b ? a : (size_t)b; }

// ---------------------------------------------------------------------------------------------------------------- atomics (one OS thread)
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) ((void)(*(p) = (v)))
template <class T, class U> static inline T emu_fetch_add(T *p, U v) { T o = *p; *p = (T)(o + v); return o; }
#define __hip_atomic_fetch_add(p, v, order, scope) emu_fetch_add((p), (v))
template <class T, class U> static inline T atomicAdd(T *p, U v) { T o = *p; *p = (T)(o + v); return o; }
template <class T, class U> static inline T atomicOr(T *p, U v) { T o = *p; *p = (T)(o | v); return o; }
template <class T, class U> static inline T atomicMax(T *p, U v) { T o = *p; *p = o < (T)v ? (T)v : o; return o; }
template <class T, class U> static inline T atomicMin(T *p, U v) { T o = *p; *p = (T)v < o ? (T)v : o; return o; }

// ---------------------------------------------------------------------------------------------------------------- wave operations
template <class T> static inline uint64_t emu_bits(T v) { uint64_t u = 0; static_assert(sizeof(T) <= 8, ""); memcpy(&u, &v, sizeof(T)); return u; }
template <class T> static inline T emu_unbits(uint64_t u) { T v; memcpy(&v, &u, sizeof(T)); return v; }
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
  (void)width;
  const uint64_t *s = emu::wave_gather(emu_bits(v), nullptr);
  return emu_unbits<T>(s[(emu::cur->lin & 63) ^ mask]);
}
template <class T> static inline T __shfl(T v, int src, int width = 64) {
  (void)width;
  const uint64_t *s = emu::wave_gather(emu_bits(v), nullptr);
  return emu_unbits<T>(s[src & 63]);
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
  (void)width;
  const uint64_t *s = emu::wave_gather(emu_bits(v), nullptr);
  const int l = (emu::cur->lin & 63) + (int)d;
  return l < 64 ? emu_unbits<T>(s[l]) : v;
}
static inline unsigned long long __ballot(int pred) {
  uint64_t active;
  const uint64_t *s = emu::wave_gather(pred ? 1 : 0, &active);
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l) m |= (uint64_t)(s[l] & 1) << l;
  return m & active;
}
static inline unsigned long long __builtin_amdgcn_ballot_w64(bool pred) { return __ballot(pred); }
static inline int __builtin_amdgcn_readfirstlane(int v) {
  uint64_t active;
  const uint64_t *s = emu::wave_gather((uint32_t)v, &active);
  return (int)(uint32_t)s[__builtin_ctzll(active)];
}
static inline int __builtin_amdgcn_readlane(int v, int lane) {
  const uint64_t *s = emu::wave_gather((uint32_t)v, nullptr);
  return (int)(uint32_t)s[lane & 63];
}
// v_mov_b32_dpp: the controls the kernels use -- row_shr:n (0x111..0x11f), row_shl:n (0x101..0x10f), row_bcast:15 (0x142), row_bcast:31
// (0x143), quad_perm (0x00..0xff).  A lane whose row / bank is masked off, or whose source lane does not exist, keeps `old` (bound_ctrl:
// 0 instead, for a source outside the row).
static inline unsigned __builtin_amdgcn_update_dpp(unsigned old, unsigned src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  uint64_t active;
  const uint64_t *s = emu::wave_gather(src, &active);
  const int lane = emu::cur->lin & 63, row = lane >> 4, in_row = lane & 15;
  if (!((row_mask >> row) & 1) || !((bank_mask >> (in_row >> 2)) & 1)) return old;
  int from = -1;
  bool outside = false;
  if (ctrl >= 0x111 && ctrl <= 0x11f) {
    const int n = ctrl - 0x110;
    if (in_row - n >= 0) from = lane - n; else outside = true;
  } else if (ctrl >= 0x101 && ctrl <= 0x10f) {
    const int n = ctrl - 0x100;
    if (in_row + n < 16) from = lane + n; else outside = true;
  } else if (ctrl == 0x142) {
    if (row >= 1) from = 16 * (row - 1) + 15; else outside = true;
  } else if (ctrl == 0x143) {
    if (row >= 2) from = 31; else outside = true;
  } else if (ctrl >= 0 && ctrl <= 0xff) {
    from = (lane & ~3) + ((ctrl >> (2 * (lane & 3))) & 3);
  } else {
    fprintf(stderr, "simt_emu: DPP control 0x%x is not emulated\n", ctrl);
    abort();
  }
  if (outside) return bound_ctrl ? 0u : old;
  if (!((active >> from) & 1)) return bound_ctrl ? 0u : old;
  return (unsigned)s[from];
}
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) {  // the median of three (no NaN handling beyond the comparisons')
  return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c));
}
static inline float __builtin_amdgcn_sqrtf(float x) { return sqrtf(x); }
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline void __builtin_amdgcn_s_waitcnt(int) {}
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_s_barrier() { emu::barrier(); }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
// v_mfma_f32_16x16x32_{f16,bf16}: D[16 x 16] = A[16 x 32] B[32 x 16] + C.  Lane l holds A[l % 16][8 (l / 16) .. + 7], B[8 (l / 16) .. + 7][l % 16]
// and C / D[4 (l / 16) + i][l % 16], i = 0 .. 3 (the layout csrc/victim_bf3.hip's comments state).  Every product of two 16-bit values is
// exact; the model here: the instruction's products and C are summed exactly (in double: 32 products of 22-bit significands) and rounded
// ONCE to fp32 -- what a matrix unit with a wide internal accumulator does, and what keeps a K = 1024 chain as accurate as the hardware's
// (a first model that rounded after every product lost to a blocked CPU GEMM by 30 %).  The hardware's own internal order is unknown, so
// results are fp32-accurate, not the hardware's bits.  Equality tests BETWEEN kernels that issue the same instructions are unaffected.
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
static inline float emu_half_to_float(uint16_t h) { _Float16 v; memcpy(&v, &h, 2); return (float)v; }
static inline float emu_bf16_to_float(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
template <bool BF16, class V8>
static inline emu_f32x4 emu_mfma_16x16x32(V8 a, V8 b, emu_f32x4 c) {
  unsigned char pay[32];
  memcpy(pay, &a, 16);
  memcpy(pay + 16, &b, 16);
  const unsigned char (*s)[32] = emu::wave_gather_bytes(pay, 32);
  const int lane = emu::cur->lin & 63, n = lane & 15, m0 = 4 * (lane >> 4);
  emu_f32x4 d = c;
  for (int i = 0; i < 4; ++i) {
    double acc = (double)c[i];  // the instruction's 32 exact products and C meet in one sum, rounded ONCE to fp32 (see the comment above)
    for (int k = 0; k < 32; ++k) {
      uint16_t av, bv;
      memcpy(&av, s[(m0 + i) + 16 * (k >> 3)] + 2 * (k & 7), 2);
      memcpy(&bv, s[n + 16 * (k >> 3)] + 16 + 2 * (k & 7), 2);
      const float x = BF16 ? emu_bf16_to_float(av) : emu_half_to_float(av), y = BF16 ? emu_bf16_to_float(bv) : emu_half_to_float(bv);
      acc += (double)x * (double)y;
    }
    d[i] = (float)acc;
  }
  return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) emu_mfma_16x16x32<false>((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) emu_mfma_16x16x32<true>((a), (b), (c))
// v_mfma_f32_32x32x16_{f16,bf16}: D[32 x 32] = A[32 x 16] B[16 x 32] + C.  Lane l = (r = l % 32, h = l / 32) holds A[r][8 h .. 8 h + 7],
// B[8 h .. 8 h + 7][r] and the sixteen C / D values of column r in rows (e & 3) + 8 (e >> 2) + 4 h, e = 0 .. 15 (csrc/pointnet.hip::acc_row).
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
template <bool BF16, class V8>
static inline emu_f32x16 emu_mfma_32x32x16(V8 a, V8 b, emu_f32x16 c) {
  unsigned char pay[32];
  memcpy(pay, &a, 16);
  memcpy(pay + 16, &b, 16);
  const unsigned char (*s)[32] = emu::wave_gather_bytes(pay, 32);
  const int lane = emu::cur->lin & 63, r = lane & 31, h = lane >> 5;
  emu_f32x16 d = c;
  for (int e = 0; e < 16; ++e) {
    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
    double acc = (double)c[e];
    for (int k = 0; k < 16; ++k) {
      uint16_t av, bv;
      memcpy(&av, s[row + 32 * (k >> 3)] + 2 * (k & 7), 2);
      memcpy(&bv, s[r + 32 * (k >> 3)] + 16 + 2 * (k & 7), 2);
      const float x = BF16 ? emu_bf16_to_float(av) : emu_half_to_float(av), y = BF16 ? emu_bf16_to_float(bv) : emu_half_to_float(bv);
      acc += (double)x * (double)y;
    }
    d[e] = (float)acc;
  }
  return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emu_mfma_32x32x16<false>((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu_mfma_32x32x16<true>((a), (b), (c))
// v_mfma_f32_32x32x2_f32: A[32 x 2], B[2 x 32]; lane (r, h) holds A[r][h], B[h][r]; the same C / D layout.  An fmaf chain in ascending k
// (MI355X_MICROARCH.md: the f32 matrix instruction is "exact f32 (= fmaf chain, bitwise)").
static inline emu_f32x16 emu_mfma_32x32x2_f32(float a, float b, emu_f32x16 c) {
  float pay[2] = {a, b};
  const unsigned char (*s)[32] = emu::wave_gather_bytes(pay, 8);
  const int lane = emu::cur->lin & 63, r = lane & 31, h = lane >> 5;
  emu_f32x16 d = c;
  for (int e = 0; e < 16; ++e) {
    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
    float acc = c[e];
    for (int k = 0; k < 2; ++k) {
      float x, y;
      memcpy(&x, s[row + 32 * k], 4);
      memcpy(&y, s[r + 32 * k] + 4, 4);
      acc = fmaf(x, y, acc);
    }
    d[e] = acc;
  }
  return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu_mfma_32x32x2_f32((a), (b), (c))
// the fp16x2 split's three instructions (csrc/pointnet.hip::split8v and its copies), which emu_build.py rewrites from asm to these calls
static inline uint32_t emu_cvt_pk_f16_f32(float a, float b) {
  _Float16 x = (_Float16)a, y = (_Float16)b;
  uint16_t ux, uy;
  memcpy(&ux, &x, 2);
  memcpy(&uy, &y, 2);
  return (uint32_t)ux | ((uint32_t)uy << 16);
}
static inline uint16_t emu_fma_mix_f16(uint16_t h, float s1, float s2) {  // fp16(fma((float)h, s1, s2)): exact before its single rounding here
  _Float16 r = (_Float16)fmaf(emu_half_to_float(h), s1, s2);
  uint16_t u;
  memcpy(&u, &r, 2);
  return u;
}
// the other matrix instructions are NOT emulated: kernels that use them compile and abort if run
template <class A, class B, class C> static inline C emu_no_mfma(A, B, C c, int, int, int) { fprintf(stderr, "simt_emu: MFMA kernels are not emulated\n"); abort(); return c; }
#define __builtin_amdgcn_mfma_f32_16x16x4f32 emu_no_mfma
#define __builtin_amdgcn_perm(a, b, sel) emu_perm((a), (b), (sel))
static inline uint32_t emu_perm(uint32_t a, uint32_t b, uint32_t sel) {  // v_perm_b32: bytes 0-3 of b, 4-7 of a
  const uint64_t both = ((uint64_t)a << 32) | b;
  uint32_t r = 0;
  for (int i = 0; i < 4; ++i) {
    const unsigned s = (sel >> (8 * i)) & 0xff;
    const unsigned byte = s < 8 ? (unsigned)((both >> (8 * s)) & 0xff) : (s == 0x0c ? 0u : 0xffu);
    r |= byte << (8 * i);
  }
  return r;
}
