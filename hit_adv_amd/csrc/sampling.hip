// Farthest point sampling (two semantics) and gather_points (+ deterministic grad).
//
//   K5  fps<PT, EXT>   one workgroup per cloud, points and running distances in VGPRs
//                      (PT points per lane), cloud mirrored in LDS as float4 for the winner's
//                      coordinates.  Per step: PT distance updates, a 64-bit (distance-bits,
//                      tie-key) max via two DPP reductions, one barrier, 4-slot merge.  Latency-bound
//                      serial chain of m steps.
//        MODE 0 : ShapeAttack/HiT_ADV.py:489-510 = model/pointnet2_utils.py:63-84 semantics (given start, direct-form
//                 squared distances, running distance 1e10, lowest index on ties)
//        MODE 1 : sampling_gpu.cu:69-173 semantics (start 0, |p|^2 <= 1e-3 skipped, the
//                 thread-slot tie order of the reference's shared-memory tree)
//        MODE 2 : PCT's sampler, util/other_utils.py:254-272: given start, distances by get_dists (:237-251: sqrt of
//                 the clamped Gram form in torch's own fp32 arithmetic, common.hpp::pct_dist), running distance 1e5
#include <stdlib.h>

#include "common.hpp"
#include "hitadv.h"

namespace hitadv {

__device__ __forceinline__ uint32_t bitrev_n(uint32_t v, int bits) { return __brev(v) >> (32 - bits); }

template <int PT, int MODE, typename IdxT>
__global__ __launch_bounds__(256) void fps(const float *__restrict__ xyz, const int64_t *__restrict__ start,
                                           int N, int m, int ref_bs, int ref_bits, int use_lds,
                                           IdxT *__restrict__ idx) {
  extern __shared__ float4 spts[];  // N entries when use_lds
  __shared__ unsigned long long slot[2][4];
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  xyz += (size_t)b * N * 3;
  idx += (size_t)b * m;
  constexpr bool EXT = MODE == 1, PCT = MODE == 2;
  float px[PT], py[PT], pz[PT], run[PT], rp[PT];
  uint32_t tb[PT];
#pragma unroll
  for (int u = 0; u < PT; ++u) {
    const int k = threadIdx.x + 256 * u;
    const bool in = k < N;
    const int kk = in ? k : 0;
    px[u] = xyz[kk * 3];
    py[u] = xyz[kk * 3 + 1];
    pz[u] = xyz[kk * 3 + 2];
    run[u] = PCT ? 1e5f : 1e10f;
    rp[u] = (px[u] * px[u] + py[u] * py[u]) + pz[u] * pz[u];
    bool ok = in;
    uint32_t tie = (uint32_t)k;
    if (EXT) {
      const float mag = (px[u] * px[u] + py[u] * py[u]) + pz[u] * pz[u];
      ok = ok && !((double)mag <= 1e-3);
      const uint32_t s = (uint32_t)k & (uint32_t)(ref_bs - 1);
      tie = (ref_bits ? (bitrev_n(s, ref_bits) << 16) : 0u) | ((uint32_t)k >> ref_bits);
    }
    tb[u] = ok ? 0xFFFFFFFFu - tie : 0u;  // 0 marks "never a candidate"
    if (use_lds && in) spts[k] = make_float4(px[u], py[u], pz[u], rp[u]);
  }
  int far = EXT ? 0 : (int)start[b];
  __syncthreads();
  const int steps = EXT ? m - 1 : m;
  if (EXT && threadIdx.x == 0) idx[0] = 0;
  for (int j = 0; j < steps; ++j) {
    if (!EXT && threadIdx.x == 0) idx[j] = (IdxT)far;
    float cx, cy, cz, rc;
    if (use_lds) {
      const float4 c = spts[far];
      cx = c.x; cy = c.y; cz = c.z; rc = c.w;
    } else {
      cx = xyz[far * 3]; cy = xyz[far * 3 + 1]; cz = xyz[far * 3 + 2];
      rc = (cx * cx + cy * cy) + cz * cz;
    }
    unsigned long long best = 0ull;
#pragma unroll
    for (int u = 0; u < PT; ++u) {
      const float d = PCT ? pct_dist(cx, cy, cz, rc, px[u], py[u], pz[u], rp[u]) : sqdist3(px[u], py[u], pz[u], cx, cy, cz);
      if (EXT) {
        if (tb[u] != 0u) run[u] = d < run[u] ? d : run[u];
      } else {
        run[u] = d < run[u] ? d : run[u];
      }
      const unsigned long long key = tb[u] ? (((unsigned long long)fbits(run[u]) << 32) | tb[u]) : 0ull;
      best = key > best ? key : best;
    }
    best = wave_max_u64_dpp(best);
    if (lane == 0) slot[j & 1][wave] = best;
    __syncthreads();
    unsigned long long w = slot[j & 1][0];
#pragma unroll
    for (int t = 1; t < 4; ++t) {
      const unsigned long long o = slot[j & 1][t];
      w = o > w ? o : w;
    }
    if (w == 0ull) {
      far = 0;  // no candidate anywhere: the reference's (best=-1, besti=0) fall-through
    } else {
      const uint32_t tie = 0xFFFFFFFFu - (uint32_t)(w & 0xffffffffu);
      if (EXT) {
        const uint32_t s = ref_bits ? bitrev_n(tie >> 16, ref_bits) : 0u;
        far = (int)(((tie & 0xffffu) << ref_bits) | s);
      } else {
        far = (int)tie;
      }
    }
    if (EXT && threadIdx.x == 0) idx[j + 1] = (IdxT)far;
  }
}

// Round 5: fps_lean, the step of MODE 0 / MODE 2 (256 < N <= 4080).  A step is a serial chain on ONE CU; with one wave per SIMD
// every instruction of any kind costs the wave ~4.6 cycles and every LDS round trip ~90, so the step is priced in instructions
// and trips (docs/kernels/round5.md §8; tools/tune/fps_step_probe.hip stamps the parts).  Against fps<> above:
//   * (NOT two points per instruction: with v_pk_add_f32 / v_pk_mul_f32 on the coordinates -- bit-exact in every test, -20 % of
//     a step's instructions -- a lane now and then missed ONE update of its running distance, only while another stream's kernels
//     shared the GPU; both samples caught were in lanes 48-63, right behind a v_mov_b32 that fed the packed instruction's
//     broadcast operand.  tools/fps_check_modes.py, docs/kernels/round5.md section 8.  Plain instructions since.)
//   * the running distance alone is reduced, as its bit pattern: a max inside the lane, 6 DPP maxima across the wave -- the
//     64-bit (distance, tie) key per point and its two 32-bit reductions are gone.  The winner's index is then found from the
//     wave's maximum M: ballot(run[u] >= M) for u = PT-1 .. 0 leaves the lowest u that holds it and the lowest lane of that u,
//     i.e. the lowest point index (k = thread + threads * u), which is what the key's tie word encoded.
//   * PCT's distance is sqrt(clamped d): sqrt is monotone and correctly rounded, so min(run, sqrt(d)) = sqrt(min(run^2, d))
//     exactly (1e5 = sqrt(1e10) exactly): the running value is kept SQUARED and one sqrt per wave and step replaces one per
//     point.  Points tie with the maximum when their sqrt rounds to the same float s, i.e. when run^2 >= t, t = the smallest
//     float whose sqrt rounds to s = the smallest float >= ((s + pred(s)) / 2)^2, evaluated exactly in fp64 (a 25-bit number
//     squared has 50 bits; it is never itself a float, so no rounding tie exists).
//   * the waves' (bits of the distance, ~index) keys meet in ONE LDS word by ds_max_u64: after the barrier the winner is one
//     8-byte read, not NW slots and NW - 1 64-bit compare-selects (three words in rotation, so that clearing needs no barrier).
//   * 8 waves per cloud for N > 512 (MODE 0; two per SIMD interleave, each carries half of the points).
// Tried and dropped: the candidates' coordinates out of the holder lane's registers (v_readlane under a scalar branch tree on
// u) posted beside the key, to save the LDS read of the winner's coordinates -- the branch tree and the five reads after the
// barrier cost more than the trip (N = 1024: 0.56 us per step against 0.36).
// (sqrt_preimage_floor, the exact tie threshold of PCT's sampler: csrc/arith.hpp)

// NW = waves per cloud.  A wave alone on its SIMD issues one instruction (of any kind) every 4-5 cycles at best and waits out
// every dependency itself; two waves per SIMD (NW = 8) interleave, and each carries half of the points.
// (The instrumented copy of this kernel -- part-removal probes, cycle stamps, the packed-f32 distance code that failed beside other
// streams and the in-kernel self-checks of docs/kernels/round5.md section 8 -- lives in tools/tune/fps_lean_diag.hpp, outside the product.)
template <int PT, bool PCT, int NW, typename IdxT>
__global__ __launch_bounds__(64 * NW) void fps_lean(const float *__restrict__ xyz, const int64_t *__restrict__ start, int N, int m,
                                                    IdxT *__restrict__ idx) {
  constexpr int TH = 64 * NW;
  extern __shared__ float4 spts[];  // the cloud: (x, y, z, |p|^2)
  __shared__ unsigned long long s_key[3];  // step j's winner: the waves' keys meet in word j % 3 by ds_max_u64 (no merge to compute)
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  xyz += (size_t)b * N * 3;
  idx += (size_t)b * m;
  float px[PT], py[PT], pz[PT], rp[PT];
  // running distances as BIT PATTERNS: they are >= +0, where unsigned order is float order, so v_min_u32 / v_max_u32 / the
  // unsigned DPP max do the float work without the canonicalising v_max_f32 x, x that IEEE mode puts in front of every float
  // min / max; a NaN distance (any sign) is a large unsigned number and never replaces a running value, like `d < run`.
  // Points past N hold 0 = the distance +0: they tie with an exhausted cloud's points and lose to them on the index.
  uint32_t run[PT];
#pragma unroll
  for (int u = 0; u < PT; ++u) {
    const int k = threadIdx.x + TH * u;
    const bool in = k < N;
    const int kk = in ? k : 0;
    const float x = xyz[kk * 3], y = xyz[kk * 3 + 1], z = xyz[kk * 3 + 2];
    const float r = (x * x + y * y) + z * z;
    px[u] = x, py[u] = y, pz[u] = z, rp[u] = r;
    run[u] = in ? fbits(1e10f) : 0u;  // PCT: (1e5)^2
    if (in) spts[k] = make_float4(x, y, z, r);
  }
  int far = (int)start[b];
  if (threadIdx.x < 3) s_key[threadIdx.x] = 0ull;
  typedef __attribute__((address_space(3))) unsigned long long lds_u64;
  const uint32_t key_at = (uint32_t)(uintptr_t)(lds_u64 *)&s_key[0];
  int j3 = 0;  // j % 3
  __syncthreads();
  for (int j = 0; j < m; ++j) {
    if (wave == 0) idx[j] = (IdxT)far;  // a scalar branch; the wave's lanes store one value to one address
    const float4 c = spts[far];
    uint32_t lb = 0u;
#pragma unroll
    for (int u = 0; u < PT; ++u) {
      uint32_t d;
      if (PCT) {  // common.hpp::pct_dist before its sqrt
        const float zz = fmaf(c.y, py[u], c.x * px[u]) + c.z * pz[u];
        const float dd = fmaf(-2.0f, zz, c.w + rp[u]);
        d = dd < 0.f ? fbits(1e-7f) : fbits(dd);
      } else {
        d = fbits(sqdist3(px[u], py[u], pz[u], c.x, c.y, c.z));
      }
      run[u] = d < run[u] ? d : run[u];
      lb = run[u] > lb ? run[u] : lb;
    }
    const uint32_t M = wave_max_u32_dpp(lb);  // wave-uniform
    uint32_t value = M, floor = M;
    if (PCT) {
      const float sq = __builtin_sqrtf(__uint_as_float(M));
      value = fbits(sq);
      floor = sq > 0.f ? fbits(sqrt_preimage_floor(sq)) : 0u;
    }
    int U = 0;
    unsigned long long holders = 0ull;
#pragma unroll
    for (int u = PT - 1; u >= 0; --u) {
      const unsigned long long h = __builtin_amdgcn_ballot_w64(run[u] >= floor);
      if (h) holders = h, U = u;
    }
    const uint32_t k = (uint32_t)(TH * U + 64 * wave + (int)__builtin_ctzll(holders));  // u = 0 of every wave is inside the cloud
    const unsigned long long key = ((unsigned long long)value << 32) | (0xFFFFFFFFu - k);
    // (the wait is part of the asm: the compiler does not know this is an LDS operation and puts no s_waitcnt between it and the
    // barrier -- the winner read after the barrier then depends on the order the LDS happens to serve the waves in, which a
    // co-resident kernel's LDS traffic changed: tests/test_gpu_attack.py::test_cw_attacks_in_flight_at_once_...)
    if (lane == 0) {
      unsigned long long before;  // the RETURNING form: its data coming back is proof that the LDS has performed the operation
      asm volatile("ds_max_rtn_u64 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(before) : "v"(key_at + 8u * (uint32_t)j3), "v"(key) : "memory");
      (void)before;
    }
    __syncthreads();
    const unsigned long long w = s_key[j3];
    j3 = j3 == 2 ? 0 : j3 + 1;
    // word (j + 2) % 3 was last read before this barrier and is next written after the next one: clear it in between
    if (wave == 0) s_key[j3 == 2 ? 0 : j3 + 1] = 0ull;
    far = (int)(0xFFFFFFFFu - (uint32_t)(w & 0xffffffffu));
  }
}

__global__ __launch_bounds__(256) void gather_points_k(int c, int n, int npoints,
                                                       const float *__restrict__ points,
                                                       const int32_t *__restrict__ idx,
                                                       float *__restrict__ out, long long total) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int j = (int)(e % npoints);
  const long long bc = e / npoints;
  const int b = (int)(bc / c);
  out[e] = points[bc * n + idx[(size_t)b * npoints + j]];
}

// grad_points[b,:,i] = sum_{j: idx[b,j]==i} grad_out[b,:,j]; lane (b,i) is the only writer of column i.
__global__ __launch_bounds__(256) void gather_points_grad_k(int c, int n, int npoints,
                                                            const float *__restrict__ grad_out,
                                                            const int32_t *__restrict__ idx,
                                                            float *__restrict__ grad_points) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float *gp = grad_points + (size_t)b * c * n + i;
  for (int l = 0; l < c; ++l) gp[(size_t)l * n] = 0.f;
  const int32_t *ip = idx + (size_t)b * npoints;
  const float *go = grad_out + (size_t)b * c * npoints;
  for (int j = 0; j < npoints; ++j)
    if (ip[j] == i)
      for (int l = 0; l < c; ++l) gp[(size_t)l * n] += go[(size_t)l * npoints + j];
}

// 0 (default since round 6) = the 64-bit-key kernel fps<> for every size; 1 (HITADV_FPS_FORM=1, or hitadv_debug_fps_form: A/B, tests) =
// fps_lean for MODE 0 / 2 with 256 < N <= 4080.  Why the faster kernel is not the default: its first build (distances on packed f32
// instructions) lost single running-distance updates while another stream's kernels shared the GPU, the cause was never isolated
// (docs/kernels/round5.md section 8, docs/kernels/round6.md section 1: the probes that would isolate it have not had a GPU to run
// on), and the plain-instruction build that shows 0 wrong tables in 31,872 differs from it only in instruction selection.  fps<> is the
// kernel the driver's own GPU runs have verified (rounds 1-4); 4.5 % of cfg4 is the price.
static int g_fps_form = [] { const char *e = getenv("HITADV_FPS_FORM"); return e && e[0] == '1' ? 1 : 0; }();

template <int MODE, typename IdxT>
static int launch_fps(const float *xyz, const int64_t *start, int B, int N, int m, IdxT *idx, hipStream_t s) {
  int ref_bs = 1, ref_bits = 0;
  if (MODE == 1) {  // include/cuda_utils.h:15-18 opt_n_threads: clamp(2^floor(log2 n), 1, 512)
    while (ref_bs * 2 <= N && ref_bs < 512) {
      ref_bs *= 2;
      ++ref_bits;
    }
  }
  const int use_lds = N <= 4080;  // the cloud as float4 + the slots inside the 64 KB a launch gets without opting in
  const size_t shm = use_lds ? (size_t)N * sizeof(float4) : 0;
  if (MODE != 1 && g_fps_form != 0 && use_lds && N > 256) {  // fps_lean: the cloud in LDS, 4 or 8 waves per cloud
#define HITADV_FPS_LEAN(PT, NW)                                                                                   \
  if (N <= 64 * NW * PT) {                                                                                        \
    fps_lean<PT, MODE == 2, NW, IdxT><<<B, 64 * NW, shm, s>>>(xyz, start, N, m, idx);                              \
    return 0;                                                                                                     \
  }
    // waves per cloud (tools/tune/fps_step_probe.hip, us per step at 4 / 8 / 16 waves, before the posting lane's wait): N = 2048
    // 0.381 / 0.349 / 0.379, N = 1024 0.301 / 0.293; PCT's distance at N = 1024 0.379 / 0.419 (its sqrt and threshold are per wave)
    static const int eight_from = [] { const char *e = getenv("HITADV_FPS_EIGHT_FROM"); return e ? atoi(e) : 512; }();  // tuning (plain instructions: N = 1024 0.333 on 4 waves, 0.314 on 8)
    if (MODE != 2 && N > eight_from) {
      HITADV_FPS_LEAN(2, 8)
      HITADV_FPS_LEAN(4, 8)
      HITADV_FPS_LEAN(8, 8)
    } else {
      HITADV_FPS_LEAN(2, 4)
      HITADV_FPS_LEAN(4, 4)
      HITADV_FPS_LEAN(8, 4)
      HITADV_FPS_LEAN(16, 4)
    }
#undef HITADV_FPS_LEAN
  }
#define HITADV_FPS_CASE(PT)                                                                       \
  if (N <= 256 * PT) {                                                                            \
    fps<PT, MODE, IdxT><<<B, 256, shm, s>>>(xyz, start, N, m, ref_bs, ref_bits, use_lds, idx);     \
    return 0;                                                                                     \
  }
  HITADV_FPS_CASE(1)
  HITADV_FPS_CASE(2)
  HITADV_FPS_CASE(4)
  HITADV_FPS_CASE(8)
  HITADV_FPS_CASE(16)
  HITADV_FPS_CASE(32)
  HITADV_FPS_CASE(64)
#undef HITADV_FPS_CASE
  return HITADV_E_ARG;  // N > 16384 not supported
}

}  // namespace hitadv

using namespace hitadv;

extern "C" int hitadv_debug_fps_form(int form) {
  const int old = g_fps_form;
  if (form == 0 || form == 1) g_fps_form = form;
  return old;
}

extern "C" int hitadv_fps_from_start(const float *xyz, const int64_t *start, int B, int N, int m,
                                     int64_t *idx, void *stream) {
  if (!xyz || !start || !idx || B <= 0 || N <= 0 || m <= 0) return HITADV_E_ARG;
  int rc = launch_fps<0, int64_t>(xyz, start, B, N, m, idx, (hipStream_t)stream);
  if (rc) return rc;
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_fps_pct(const float *xyz, const int64_t *start, int B, int N, int m, int64_t *idx, void *stream) {
  if (!xyz || !start || !idx || B <= 0 || N <= 0 || m <= 0) return HITADV_E_ARG;
  int rc = launch_fps<2, int64_t>(xyz, start, B, N, m, idx, (hipStream_t)stream);
  if (rc) return rc;
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp,
                                              int32_t *idxs, void *stream) {
  (void)temp;  // running distances live in registers; the scratch tensor of the reference is unused
  if (!dataset || !idxs || b <= 0 || n <= 0) return HITADV_E_ARG;
  if (m <= 0) return 0;  // sampling_gpu.cu:73
  int rc = launch_fps<1, int32_t>(dataset, nullptr, b, n, m, idxs, (hipStream_t)stream);
  if (rc) return rc;
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_gather_points(int b, int c, int n, int npoints, const float *points, const int32_t *idx,
                                    float *out, void *stream) {
  if (!points || !idx || !out || b <= 0 || c <= 0 || n <= 0 || npoints <= 0) return HITADV_E_ARG;
  const long long total = (long long)b * c * npoints;
  gather_points_k<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(c, n, npoints, points, idx,
                                                                                    out, total);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                                         const int32_t *idx, float *grad_points, void *stream) {
  if (!grad_out || !idx || !grad_points || b <= 0 || c <= 0 || n <= 0 || npoints <= 0) return HITADV_E_ARG;
  dim3 grid((n + 255) / 256, b);
  gather_points_grad_k<<<grid, 256, 0, (hipStream_t)stream>>>(c, n, npoints, grad_out, idx, grad_points);
  HITADV_LAUNCH_CHECK();
  return 0;
}
