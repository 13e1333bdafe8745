// The INSTRUMENTED copy of sampling.hip::fps_lean (round 5's hunt, docs/kernels/round5.md section 8) -- not part of the product library.
// Included behind hit_adv_amd/csrc/sampling.hip by tools/tune/fps_step_probe.hip (what each part of a step costs) and by
// tools/tune/fps_diag_lib.hip (the small library tools/fps_check.py loads to run the failing packed-f32 build beside the shipped
// kernels: tools/fps_packed_repro.sh).  Everything but the instrumentation is the shipped kernel's text; with PROBE = 0 and
// HITADV_FPS_DIAG = 0 it compiles to the shipped kernel's instructions.
#pragma once
namespace hitadv {
#ifndef HITADV_FPS_DIAG
#define HITADV_FPS_DIAG 0
#endif
// tuning builds (-DHITADV_FPS_DIAG=n; docs/kernels/round5.md section 8): 2 = MODE 0's distances on packed f32 instructions, the build
// that tools/fps_check_modes.py shows failing; 4 = at the end, the coordinates in registers against memory and
// the LDS copy against the registers; 5 = every wave's step counter behind every barrier; 9 = a log of every wave's key and centre
__device__ unsigned int g_fps_dbg[8];  // [0] / [1] mismatches, [2] checked

// PROBE (tools/tune/fps_step_probe.hip only; results are garbage): 1 no read of the winner's coordinates, 2 no exchange between
// the waves, 3 no search for the holder, 4 no reduction across the lanes, 5 cycle stamps -- what each part of the step costs.
// NW = waves per cloud.  A wave alone on its SIMD issues one instruction (of any kind) every 4-5 cycles at best and waits out
// every dependency itself; two waves per SIMD (NW = 8) interleave, and each carries half of the points.
template <int PT, bool PCT, int NW, typename IdxT, int PROBE = 0>
__global__ __launch_bounds__(64 * NW) void fps_lean_diag(const float *__restrict__ xyz, const int64_t *__restrict__ start, int N, int m,
                                                    IdxT *__restrict__ idx, unsigned long long *dbg_log = nullptr) {
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
#ifdef HITADV_FPS_COHERENT_LOADS  // diagnostic: agent-scope loads (past this CU's L1)
    const float x = __hip_atomic_load(&xyz[kk * 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float y = __hip_atomic_load(&xyz[kk * 3 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float z = __hip_atomic_load(&xyz[kk * 3 + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    const float x = xyz[kk * 3], y = xyz[kk * 3 + 1], z = xyz[kk * 3 + 2];
#endif
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
  unsigned long long acc_t[5] = {0, 0, 0, 0, 0}, t_prev = 0;  // PROBE 5: shader cycles per part of the step, summed over the steps
  auto stamp = [&](int i) {
    if (PROBE == 5) {
      const unsigned long long t = __builtin_readcyclecounter();
      acc_t[i] += t - t_prev;
      t_prev = t;
    }
  };
  if (PROBE == 5) t_prev = __builtin_readcyclecounter();
  for (int j = 0; j < m; ++j) {
    if (wave == 0) idx[j] = (IdxT)far;  // a scalar branch; the wave's lanes store one value to one address
    float4 c;
    if (PROBE == 1)
      c = make_float4(far * 1e-4f, far * 2e-4f, far * 3e-4f, far * 1e-5f);
    else {
      c = spts[far];
#if HITADV_FPS_DIAG == 9  // what lane 63 of every wave used as this step's winner and centre
      if (dbg_log != nullptr && lane == 63) {
        unsigned long long *l2 = dbg_log + (size_t)gridDim.x * m * NW;
        l2[((size_t)blockIdx.x * m + j) * NW + wave] = ((unsigned long long)__float_as_uint(c.x) << 32) | (unsigned int)(far & 0xffffff);
      }
#endif
    }
    if (PROBE == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    stamp(0);
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
#if HITADV_FPS_DIAG == 2  // the build that failed beside other streams' kernels: MODE 0's distances two points per packed instruction
      if (!PCT && (u & 1) == 0 && u + 1 < PT) {
        typedef float f2v __attribute__((ext_vector_type(2)));
        const f2v X = {px[u], px[u + 1]}, Y = {py[u], py[u + 1]}, Z = {pz[u], pz[u + 1]};
        const f2v cx2 = {c.x, c.x}, cy2 = {c.y, c.y}, cz2 = {c.z, c.z};
        const f2v dx = X - cx2, dy = Y - cy2, dz = Z - cz2;
        const f2v dd = (dx * dx + dy * dy) + dz * dz;
        d = fbits(dd[0]);
        const uint32_t d1 = fbits(dd[1]);
        run[u + 1] = d1 < run[u + 1] ? d1 : run[u + 1];
        lb = run[u + 1] > lb ? run[u + 1] : lb;
      } else if (!PCT && (u & 1) == 1) {
        continue;  // done with its even neighbour
      }
#endif
      run[u] = d < run[u] ? d : run[u];
      lb = run[u] > lb ? run[u] : lb;
    }
    if (PROBE == 5) asm volatile("" : "+v"(lb));
    stamp(1);
    const uint32_t M = PROBE == 4 ? (uint32_t)__builtin_amdgcn_readlane((int)lb, 63) : wave_max_u32_dpp(lb);  // wave-uniform
    uint32_t value = M, floor = M;
    if (PCT) {
      const float sq = __builtin_sqrtf(__uint_as_float(M));
      value = fbits(sq);
      floor = sq > 0.f ? fbits(sqrt_preimage_floor(sq)) : 0u;
    }
    int U = 0;
    unsigned long long holders = 0ull;
#pragma unroll
    for (int u = PT - 1; u >= (PROBE == 3 ? PT - 1 : 0); --u) {
      const unsigned long long h = __builtin_amdgcn_ballot_w64(run[u] >= floor);
      if (h) holders = h, U = u;
    }
    const uint32_t k = (uint32_t)(TH * U + 64 * wave + (int)__builtin_ctzll(holders));  // u = 0 of every wave is inside the cloud
    const unsigned long long key = ((unsigned long long)value << 32) | (0xFFFFFFFFu - k);
    stamp(2);
    if (PROBE == 2) {
      far = (int)k % N;
      continue;
    }
    // (the wait is part of the asm: the compiler does not know this is an LDS operation and puts no s_waitcnt between it and the
    // barrier -- the winner read after the barrier then depends on the order the LDS happens to serve the waves in, which a
    // co-resident kernel's LDS traffic changed: tests/test_gpu_attack.py::test_cw_attacks_in_flight_at_once_...)
#if HITADV_FPS_DIAG == 9  // every wave's key of every step: log[block][step][wave]
    if (dbg_log != nullptr && lane == 0) dbg_log[((size_t)blockIdx.x * m + j) * NW + wave] = key;
#endif
    unsigned long long w;
    {
      if (lane == 0) {
        unsigned long long before;  // the RETURNING form: its data coming back is proof that the LDS has performed the operation
        asm volatile("ds_max_rtn_u64 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(before) : "v"(key_at + 8u * (uint32_t)j3), "v"(key) : "memory");
        (void)before;
      }
#if HITADV_FPS_DIAG == 5  // every wave posts the step it is in before the barrier; behind it, all of them must have
      __shared__ int s_step[NW];
      if (lane == 0) s_step[wave] = j + 1;
#endif
      __syncthreads();
#if HITADV_FPS_DIAG == 5
      if (lane < NW && s_step[lane] != j + 1) atomicAdd(&g_fps_dbg[0], 1u);
      if (lane == 0) atomicAdd(&g_fps_dbg[2], 1u);
#endif
      stamp(3);
      w = s_key[j3];
      j3 = j3 == 2 ? 0 : j3 + 1;
      // word (j + 2) % 3 was last read before this barrier and is next written after the next one: clear it in between
      if (wave == 0) s_key[j3 == 2 ? 0 : j3 + 1] = 0ull;
    }
    far = (int)(0xFFFFFFFFu - (uint32_t)(w & 0xffffffffu));
    if (PROBE == 5) asm volatile("" : "+v"(far));
    stamp(4);
  }
  if (PROBE == 5 && threadIdx.x == 0 && m >= 5)
    for (int i = 0; i < 5; ++i) idx[i] = (IdxT)acc_t[i];
#if HITADV_FPS_DIAG == 4
  __syncthreads();
#pragma unroll
  for (int u = 0; u < PT; ++u) {
    const int k = threadIdx.x + TH * u;
    if (k < N) {
      const float gx = __hip_atomic_load(&xyz[k * 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const float gy = __hip_atomic_load(&xyz[k * 3 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const float gz = __hip_atomic_load(&xyz[k * 3 + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const float rx = px[u], ry = py[u], rz = pz[u];
      const float4 l = spts[k];
      if (!(gx == rx && gy == ry && gz == rz)) atomicAdd(&g_fps_dbg[0], 1u);
      if (!(l.x == rx && l.y == ry && l.z == rz)) atomicAdd(&g_fps_dbg[1], 1u);
      atomicAdd(&g_fps_dbg[2], 1u);
    }
  }
#endif
}
}  // namespace hitadv
