// V1F (round 5): the victim's 128 -> 1024 shared layer + max / arg-max over the points (model/feature_models.py:113-147,
// 165-177) with ONE fp16 product per value instead of three -- and the same fp32-accurate result.
//
// The fp16x2 form (csrc/victim_bf3.hip) evaluates y[n,c] = sum_k a[n,k] w[c,k] for all 1024 x 1024 (point, channel) pairs of a
// cloud to fp32 accuracy (three exact fp16 products per pair and k) and then keeps ONE value per channel.  Its matrix pipe is
// busy 64-75 % of the kernel's duration (profiles/r04_mfma_pmc.txt, r05_loop_traffic.json): the kernel is bound by the number
// of MFMAs.  But the maximum only needs the exact value of the few points that can BE the maximum:
//
//   seed     theta[b,c] = the exact value (three products) at the point that won channel c in the PREVIOUS iteration of the
//            attack (any point gives a valid lower bound of the maximum; last iteration's winner gives a tight one).   vf_seed_k
//   stream   yhat[n,c] = sum_k hi(a) hi(w): the first of the three products alone, 32 instead of 96 MFMAs per tile and wave.
//            |y - yhat| <= eps |a_n| |w_c| (below), so every point with yhat >= theta - eps |w_c| max_tile|a_n| is a CANDIDATE
//            and no other point can reach theta, let alone the maximum.  Candidates are rare (1.3-2.4 per cloud and channel on
//            the engine's own activations, tools/v1_filter_probe.py): their (point, channel) go to a list.                vf_stream_k
//   refine   the exact value of every candidate, the maximum and its FIRST point per channel, bias, ReLU.                 vf_refine_k
//
// The result does not depend on the seed (a poor seed only lengthens the lists), nor on the order of the lists (maximum, ties
// to the lower point).  The exact values are the three-product sums of the fp16x2 form evaluated on the VALU (v_fma_mix, 16
// lanes per candidate, fixed tree): fp32-accurate like the MFMA form's, not the same last bit.
// Error bound.  a = ah + al / 2048 + ra with |a - ah| <= u |a|, u = 2^-11 (fp16, round to nearest; |a - ah| <= 2^-25 where ah
// is subnormal), the same for w: |a w - ah wh| <= (2 u + u^2) |a| |w|; the products are exact in fp32 and the two accumulations
// (MFMA, VALU chain) each stay within 128 x 2^-24 sum |a||w| of their exact sums; sum_k |a_k| |w_k| <= |a_n| |w_c| (Cauchy-Schwarz).
// eps = 1.05e-3 > 2^-10 (1 + 2^-11) + 3 x 128 x 2^-24 with 4 % to spare; |a_n| is floored at 1e-3 (covers the subnormal case:
// 128 x 2^-25 |w|_inf < 1.05e-3 x 1e-3 x |w_c|), and taken from the hi pieces with a 0.2 % allowance.
// Lists that do not fit (HITADV_V1F_CAP = 2048 entries per cloud and 32 channels: pathological input, e.g. a cloud of identical
// points) raise the caller's range flag.
//
// STATUS (round 5, docs/kernels/round5.md): correct (tests/test_gpu_kernels.py::test_filtered_linear_max_*: maxima to fp32
// roundoff, a maximiser as arg-max, independent of the seeds, overfull lists reported) and NOT faster yet, so the engine does not
// use it: 257 us per 256 clouds on 128 workgroups against the unfiltered kernel's 257 (seed 22 + stream 185 + refine 50).  What
// the stream taught: on this chip a tile's MFMA time (2 waves x 32 x 16 cycles per SIMD), its LDS fragment reads (8 waves x 16 KB
// at 128 B / clock per CU) and its vector instructions (4 cycles each) ADD UP here (3040 cycles per tile = 1024 + 1024 + ~1000)
// where the unfiltered kernel overlaps them (4256 measured against 3072 + 2048 + 1184): with a third of the matrix work the
// per-tile latencies (bound -> accumulator start values, accumulators -> sign bits -> lists) are no longer hidden by two waves
// per SIMD.  Next steps if taken up again: 64 channels per wave (half the fragment reads per MFMA), the tile norms from the
// producer (rowmlp_stream_k) instead of the stash, candidates grouped by channel in the refine pass (half its traffic).
#include <stdlib.h>

#include "../common.hpp"
#include "hitadv.h"
#include "hitadv_experimental.h"

namespace hitadv {

typedef float f32x4f __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8f __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2f __attribute__((ext_vector_type(2)));
constexpr int VF_TM = 64;                 // points per tile
constexpr int VF_CIN = 128;
constexpr int VF_NSL = VF_CIN / 32;       // 32-deep MFMA slices
constexpr int VF_RS = 2 * VF_CIN + 32;    // bytes per LDS row of the hi image (conflict-free 16x16x32 A-fragment reads)
constexpr int VF_PIECE = VF_TM * VF_RS;
constexpr int VF_CAP = HITADV_V1F_CAP;    // candidate entries per (cloud, 32 channels)
constexpr float VF_EPS = 1.05e-3f;
constexpr float VF_AMIN = 1.0e-3f;

// The exact value of (row, channel c): 16 lanes, lane q takes k = 8 q .. 8 q + 7; every lane of the group returns the sum.
//   row: the activation's packed words (fp16 hi | fp16 lo << 16, lo scaled by 2^11), W2: hitadv_split_weights_f16x2's image
__device__ __forceinline__ float vf_exact(const uint32_t *__restrict__ row, const uint16_t *__restrict__ W2, int Cout, int c, int q) {
  const uint4 x0 = *reinterpret_cast<const uint4 *>(row + 8 * q), x1 = *reinterpret_cast<const uint4 *>(row + 8 * q + 4);
  const size_t wo = ((size_t)(c >> 4) * VF_NSL + (q >> 2)) * 64 + (q & 3) * 16 + (c & 15);
  const uint4 *Wq = reinterpret_cast<const uint4 *>(W2);
  const uint4 wh = Wq[wo], wl = Wq[(size_t)(Cout >> 4) * VF_NSL * 64 + wo];
  const uint32_t xw[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
  const uint32_t hw[4] = {wh.x, wh.y, wh.z, wh.w}, lw[4] = {wl.x, wl.y, wl.z, wl.w};
  float ph = 0.f, pl = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f16x2f a = __builtin_bit_cast(f16x2f, xw[i]);  // [0] = hi piece, [1] = lo piece
    const f16x2f bh2 = __builtin_bit_cast(f16x2f, hw[i >> 1]), bl2 = __builtin_bit_cast(f16x2f, lw[i >> 1]);
    const float ah = (float)a[0], al = (float)a[1], bh = (float)bh2[i & 1], bl = (float)bl2[i & 1];
    ph = fmaf(ah, bh, ph);
    pl = fmaf(al, bh, pl);
    pl = fmaf(ah, bl, pl);
  }
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) {  // a fixed butterfly over the group's 16 lanes
    ph += __shfl_xor(ph, m, HITADV_WAVE);
    pl += __shfl_xor(pl, m, HITADV_WAVE);
  }
  return fmaf(pl, 1.f / 2048.f, ph);
}

// theta[b, c] = the exact value at last iteration's winner (any valid point; out-of-range entries are read as point 0)
__global__ __launch_bounds__(256) void vf_seed_k(const uint32_t *__restrict__ Xp, const uint16_t *__restrict__ W2, int N, int Cout,
                                                 const int64_t *__restrict__ seed, float *__restrict__ theta) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rg = blockIdx.x * 4 + wave;  // 32 channels
  if (32 * rg >= Cout) return;
  const int q = lane & 15, grp = lane >> 4;
#pragma unroll
  for (int e0 = 0; e0 < 32; e0 += 4) {  // (fully unrolled: the eight rounds' loads are all in flight together)
    const int c = 32 * rg + e0 + grp;
    const int n = sane_index((int)seed[(size_t)b * Cout + c], N, 0);
    const float v = vf_exact(Xp + ((size_t)b * N + n) * VF_CIN, W2, Cout, c, q);
    if (q == 0) theta[(size_t)b * Cout + c] = v;
  }
}

// The stream: the clouds of a workgroup as one sequence of 64-point tiles (linear_max_fwd_bf3_k's FLAT form: N a multiple of
// 128, no split), hi pieces only.  block = 8 waves, 256 channels (32 per wave); a lane's accumulators: acc[rt][ct][i] = point
// 16 rt + 4 (lane / 16) + i of the tile, channel 16 ct + lane % 16.
// ABL != 0: tuning builds only (hitadv_debug_vf_ablate: the kernel with one cost removed; results are garbage)
template <int ABL>
__global__ __launch_bounds__(512) void vf_stream_k(const uint32_t *__restrict__ Xp, const uint16_t *__restrict__ W2,
                                                   const float *__restrict__ wnorm, int B_, int N, int Cout, int ncg, int cpb,
                                                   const float *__restrict__ theta, uint32_t *__restrict__ cand,
                                                   int32_t *__restrict__ ccount, int *overflow) {
  constexpr int G8 = VF_CIN / 8;             // 8-value groups per row
  constexpr int ST = VF_TM * G8 / 512;       // groups staged per thread per tile (2)
  extern __shared__ __attribute__((aligned(16))) char sVF[];  // 2 x hi image, then the waves' candidate lists
  uint32_t *const sEnt = reinterpret_cast<uint32_t *>(sVF + 2 * VF_PIECE);
  __shared__ __attribute__((aligned(16))) float sNorm[2][8];              // per tile buffer and wave: max over the wave's rows of sum_k ah^2
  int cg, b;
  {  // XCD-aware block order (linear_max_fwd_bf3_k): the column-group blocks that stream the same tiles share an XCD
    const int id = blockIdx.x, nrg = (B_ + cpb - 1) / cpb;
    if ((nrg & 7) == 0) {
      const int xcd = id & 7, slot = id >> 3;
      cg = slot % ncg;
      b = (slot / ncg) * 8 + xcd;
    } else {
      cg = id % ncg;
      b = id / ncg;
    }
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l16 = lane & 15, g4 = lane >> 4;
  const int col0 = cg * 256 + wave * 32;
  const int b0 = b * cpb, nb = min(cpb, B_ - b0);
  const int tpc = N / VF_TM;                 // tiles per cloud (even)
  const int ntiles = nb * tpc;
  const uint32_t *const X = Xp + (size_t)b0 * N * VF_CIN;

  uint4 w[2][VF_NSL];  // hi pieces of the wave's 32 channels: column tile ct, slice j
  {
    const uint4 *wp = reinterpret_cast<const uint4 *>(W2) + (size_t)(col0 / 16) * VF_NSL * 64 + lane;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int j = 0; j < VF_NSL; ++j) w[ct][j] = wp[(size_t)(ct * VF_NSL + j) * 64];
  }
  float ew[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) ew[ct] = VF_EPS * wnorm[col0 + 16 * ct + l16];

  uint32_t soff[ST];
#pragma unroll
  for (int u = 0; u < ST; ++u) {
    const int e = threadIdx.x + 512 * u;
    soff[u] = (uint32_t)((e / G8) * VF_CIN + 8 * (e % G8)) * 4u;
  }
  uint4 stA[ST][2], stB[ST][2];
  auto fetch = [&](uint4 (&st)[ST][2], int tile) {
    const char *tb = reinterpret_cast<const char *>(X) + (size_t)tile * VF_TM * VF_CIN * 4;
#pragma unroll
    for (int u = 0; u < ST; ++u) {
      st[u][0] = *reinterpret_cast<const uint4 *>(tb + soff[u]);
      st[u][1] = *reinterpret_cast<const uint4 *>(tb + soff[u] + 16);
    }
  };
  auto stash = [&](const uint4 (&st)[ST][2], int tile) {
    const int buf = tile & 1;
    float sq[ST];
#pragma unroll
    for (int u = 0; u < ST; ++u) {
      const int e = threadIdx.x + 512 * u;
      uint4 hi;
      hi.x = __builtin_amdgcn_perm(st[u][0].y, st[u][0].x, 0x05040100u);
      hi.y = __builtin_amdgcn_perm(st[u][0].w, st[u][0].z, 0x05040100u);
      hi.z = __builtin_amdgcn_perm(st[u][1].y, st[u][1].x, 0x05040100u);
      hi.w = __builtin_amdgcn_perm(st[u][1].w, st[u][1].z, 0x05040100u);
      *reinterpret_cast<uint4 *>(sVF + (size_t)buf * VF_PIECE + (e / G8) * VF_RS + 16 * (e % G8)) = hi;
      // |ah|^2 of the row: this thread's eight values ...
      float q = 0.f;
      q = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2f, hi.x), __builtin_bit_cast(f16x2f, hi.x), q, false);
      q = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2f, hi.y), __builtin_bit_cast(f16x2f, hi.y), q, false);
      q = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2f, hi.z), __builtin_bit_cast(f16x2f, hi.z), q, false);
      q = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2f, hi.w), __builtin_bit_cast(f16x2f, hi.w), q, false);
      sq[u] = q;
    }
    if (ABL != 3) {
      // ... then the row's 16 lanes (row_shr sums end in the row's last lane), the larger of the thread's two rows, the wave's four
      // row groups (row_bcast), and ONE plain store per wave: no atomics (the compiler turns a same-address LDS atomic of a
      // divergent value into a loop over the active lanes)
#pragma unroll
      for (int u = 0; u < ST; ++u) {
        sq[u] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq[u]), 0x111, 0xf, 0xf, true));  // row_shr:1
        sq[u] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq[u]), 0x112, 0xf, 0xf, true));  // row_shr:2
        sq[u] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq[u]), 0x114, 0xf, 0xf, true));  // row_shr:4
        sq[u] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq[u]), 0x118, 0xf, 0xf, true));  // row_shr:8
      }
      float m = ST > 1 ? __builtin_fmaxf(sq[0], sq[ST - 1]) : sq[0];  // valid in lanes 15, 31, 47, 63 (sums of squares: >= 0)
      m = l16 == 15 ? m : 0.f;
      const float a = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x142, 0xa, 0xf, false));  // row_bcast:15
      m = __builtin_fmaxf(m, a);
      const float c = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x143, 0xc, 0xf, false));  // row_bcast:31
      m = __builtin_fmaxf(m, c);
      if (lane == 63) sNorm[buf][wave] = m;
    }
  };

  int cnt = 0;                 // wave-uniform: entries in this wave's list for the current cloud
  float th[2];                 // theta of the current cloud for the lane's two channels
  uint32_t *const myEnt = sEnt + wave * VF_CAP;
  auto load_theta = [&](int cloud) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) th[ct] = theta[(size_t)(b0 + cloud) * Cout + col0 + 16 * ct + l16];
  };
  auto compute = [&](int tile, int tile_in_cloud) {
    const char *base = sVF + (size_t)(tile & 1) * VF_PIECE + l16 * VF_RS + 16 * g4;
    // the tile's bound first: eps |w_c| max_rows |a_n| (hi pieces, 0.2 % allowance; floor: see the header).  The accumulators START
    // at -t (t = the lane's current lower bound of the channel's maximum minus the bound): after the products their SIGN BIT says
    // whether a value can still be the maximum -- one instruction per value (below) instead of a compare and a select
    const float4 na = *reinterpret_cast<const float4 *>(&sNorm[tile & 1][0]), nb = *reinterpret_cast<const float4 *>(&sNorm[tile & 1][4]);
    const float a2 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(na.x, na.y), __builtin_fmaxf(na.z, na.w)),
                                     __builtin_fmaxf(__builtin_fmaxf(nb.x, nb.y), __builtin_fmaxf(nb.z, nb.w)));
    const float amax = fmaxf(__builtin_sqrtf(a2) * 1.002f, VF_AMIN);
    const float d0 = ew[0] * amax, d1 = ew[1] * amax;
    const float t0 = th[0] - d0, t1 = th[1] - d1;
    f32x4f acc[4][2];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      acc[rt][0] = f32x4f{-t0, -t0, -t0, -t0};
      acc[rt][1] = f32x4f{-t1, -t1, -t1, -t1};
    }
    uint4 fa[2][4];  // [buffer][row tile]: the next slice's A fragments are read while this slice's MFMAs run
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) fa[0][rt] = *reinterpret_cast<const uint4 *>(base + rt * 16 * VF_RS);
#pragma unroll
    for (int j = 0; j < VF_NSL; ++j) {
      if (j + 1 < VF_NSL) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) fa[(j + 1) & 1][rt] = *reinterpret_cast<const uint4 *>(base + rt * 16 * VF_RS + 64 * (j + 1));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
          if (ABL != 2) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8f, fa[j & 1][rt]), __builtin_bit_cast(f16x8f, w[ct][j]),
                                                               acc[rt][ct], 0, 0, 0);
    }
    // acc = yhat - t (the start value adds one rounding of size 2^-24 |t| to the sum: inside eps's 4 % reserve).  Every value of the
    // tile is itself a lower bound of the maximum once its error is taken off: the lane's bound rises with the stream (a poor seed --
    // the first iteration of an attack -- costs some tens of candidates per channel, not hundreds).
    uint32_t miss[2] = {0u, 0u};  // bit 15 - j: value j = 4 rt + i is BELOW the bound (v_alignbit: miss = 2 miss + sign(acc))
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      float m = __builtin_fmaxf(__builtin_fmaxf(acc[0][ct][0], acc[0][ct][1]), acc[0][ct][2]);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int i = (rt == 0 ? 3 : 0); i < 4; i += 2)
          m = i + 1 < 4 ? __builtin_fmaxf(__builtin_fmaxf(m, acc[rt][ct][i]), acc[rt][ct][i + 1]) : __builtin_fmaxf(m, acc[rt][ct][i]);
      // max yhat = m + t; the new bound: max(th, max yhat - d)  (rounded down by the 0.2 % / 4 % reserves)
      float nt = __builtin_fmaxf(th[ct], (m + (ct ? t1 : t0)) - (ct ? d1 : d0));
      // the channel's four lanes (other rows of the tile) share it -- where it moves most (a cloud's first tiles), then now and then:
      // two dependent LDS round trips at the end of a tile are not free with two waves per SIMD
      if (ABL != 5 && (tile_in_cloud < 2 || (tile_in_cloud & 3) == 3)) {  // wave-uniform
        nt = __builtin_fmaxf(nt, __shfl_xor(nt, 16, HITADV_WAVE));
        nt = __builtin_fmaxf(nt, __shfl_xor(nt, 32, HITADV_WAVE));
      }
      th[ct] = nt;
      if (ABL != 4) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
          for (int i = 0; i < 4; ++i) miss[ct] = __builtin_amdgcn_alignbit(miss[ct], __float_as_uint(acc[rt][ct][i]), 31);
      }
    }
    uint32_t bits[2] = {~miss[0] & 0xffffu, ~miss[1] & 0xffffu};
    if (ABL == 4) bits[0] = bits[1] = 0u;
    uint32_t both = bits[0] | (bits[1] << 16);
    if (ABL == 1) both = 0u;
    if (__builtin_amdgcn_ballot_w64(both != 0u) != 0ull) {  // wave-uniform; the lanes with hits take them out one by one
      const uint32_t pbase = (uint32_t)(tile_in_cloud * VF_TM + 4 * g4) << 8;
      unsigned long long act;
      while ((act = __builtin_amdgcn_ballot_w64(both != 0u)) != 0ull) {
        const bool on = both != 0u;
        const int k = on ? 31 - __builtin_clz(both) : 0;   // highest set bit first: ct = k / 16, value j = 15 - k % 16
        both &= ~(on ? (1u << k) : 0u);
        const int j = 15 - (k & 15);
        const int pos = cnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
        if (on && pos < VF_CAP) myEnt[pos] = pbase + ((uint32_t)(16 * (j >> 2) + (j & 3)) << 8) + (uint32_t)(16 * (k >> 4) + l16);
        cnt += __builtin_popcountll(act);
      }
    }
  };
  auto flush = [&](int cloud) {  // the cloud's list leaves; an overfull one raises the flag (the caller re-runs without the filter)
    const size_t region = (size_t)(b0 + cloud) * (Cout / 32) + (col0 >> 5);
    const int n = min(cnt, VF_CAP);
    for (int i = lane; i < n; i += 64) cand[region * VF_CAP + i] = myEnt[i];
    if (lane == 0) {
      ccount[region] = n;
      if (cnt > VF_CAP) *overflow = 1;
    }
    cnt = 0;
  };

  fetch(stA, 0);
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): W and tile 0 are complete before the loop (see victim_bf3.hip)
  __syncthreads();                     // sNorm is zero
  stash(stA, 0);
  fetch(stA, min(1, ntiles - 1));
  load_theta(0);
  __syncthreads();
  const bool late = wave >= 4;
  auto step = [&](int tile, int tic, uint4 (&have)[ST][2], uint4 (&next)[ST][2]) {
    fetch(next, min(tile + 2, ntiles - 1));  // past the end: the last tile again, into the buffer nobody reads any more
    if (late) stash(have, tile + 1);
    compute(tile, tic);
    if (!late) stash(have, tile + 1);
    __syncthreads();
  };
  for (int cloud = 0, tile = 0; cloud < nb; ++cloud) {
    for (int tic = 0; tic < tpc; tic += 2, tile += 2) {
      step(tile, tic, stA, stB);
      step(tile + 1, tic + 1, stB, stA);
    }
    flush(cloud);
    if (cloud + 1 < nb) load_theta(cloud + 1);
  }
}

// The candidates' exact values, the maximum and its first point per channel; bias, ReLU; the winners are the next call's seeds.
__global__ __launch_bounds__(256) void vf_refine_k(const uint32_t *__restrict__ Xp, const uint16_t *__restrict__ W2,
                                                   const float *__restrict__ bias, int N, int Cout, int relu,
                                                   const uint32_t *__restrict__ cand, const int32_t *__restrict__ ccount,
                                                   float *__restrict__ out, int64_t *__restrict__ idx, int64_t *__restrict__ seed) {
  constexpr int CH = 256;  // entries per pass
  __shared__ float sVal[4][CH];
  __shared__ uint32_t sE[4][CH];
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rg = blockIdx.x * 4 + wave;
  if (32 * rg >= Cout) return;
  const size_t region = (size_t)b * (Cout / 32) + rg;
  const int cnt = min(ccount[region], VF_CAP);
  const int q = lane & 15, grp = lane >> 4;
  // lane c of the wave: the best of its channel's entries (ties: the lower point); lanes 32-63 take the odd entries
  const int ch = lane & 31, half = lane >> 5;
  float best = -__builtin_inff();
  int bp = 0x7fffffff;
  for (int c0 = 0; c0 < cnt; c0 += CH) {
    const int m = min(CH, cnt - c0);
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < m; i += 64) sE[wave][i] = cand[region * VF_CAP + c0 + i];
    __builtin_amdgcn_wave_barrier();
#pragma unroll 4
    for (int e0 = 0; e0 < m; e0 += 4) {
      const int e = min(e0 + grp, m - 1);
      const uint32_t ent = sE[wave][e];
      const int n = sane_index((int)(ent >> 8), N, 0), c = 32 * rg + (int)(ent & 31u);
      const float v = vf_exact(Xp + ((size_t)b * N + n) * VF_CIN, W2, Cout, c, q);
      if (q == 0 && e0 + grp < m) sVal[wave][e0 + grp] = v;
    }
    __builtin_amdgcn_wave_barrier();
    for (int e = half; e < m; e += 2) {
      const uint32_t ent = sE[wave][e];
      const float v = sVal[wave][e];
      const int p = (int)(ent >> 8);
      const bool mine = (int)(ent & 31u) == ch;
      const bool take = mine && (v > best || (v == best && p < bp));
      best = take ? v : best;
      bp = take ? p : bp;
    }
  }
  {
    const float ov = __shfl_xor(best, 32, HITADV_WAVE);
    const int op = __shfl_xor(bp, 32, HITADV_WAVE);
    if (ov > best || (ov == best && op < bp)) { best = ov; bp = op; }
  }
  if (half == 0) {
    const int c = 32 * rg + ch;
    const int point = sane_index(bp, N, 0);  // (no candidate at all: a NaN cloud; the range flag is the caller's business)
    float v = best + (bias ? bias[c] : 0.f);
    v = relu ? (v > 0.f ? v : 0.f) : v;
    out[(size_t)b * Cout + c] = v;
    idx[(size_t)b * Cout + c] = point;
    if (seed != nullptr) seed[(size_t)b * Cout + c] = point;
  }
}

// (the grid rule of csrc/victim_bf3.hip::bf3_split, FLAT case: clouds per workgroup for about `blocks` workgroups)
static int vf_clouds_per_block(int B, int Cout, int blocks) {
  const int colgroups = Cout / 256, cus = blocks > 0 ? blocks : 256;
  return cus < B * colgroups ? (B * colgroups + cus - 1) / cus : 1;
}

}  // namespace hitadv

using namespace hitadv;

// tuning only: 1 = no candidate bookkeeping, 2 = no MFMAs, 3 = no row-norm atomics, 4 = no compares (results are garbage)
static int g_vf_ablate = 0;
extern "C" int hitadv_debug_vf_ablate(int what) {
  const int before = g_vf_ablate;
  if (what >= 0 && what <= 5) g_vf_ablate = what;
  return before;
}

extern "C" int hitadv_linear_max_filter_supported(int B, int N, int Cin, int Cout, int blocks) {
  if (B <= 0 || N <= 0 || Cin != VF_CIN || Cout <= 0 || (Cout & 255) || N % (2 * VF_TM) || N > (1 << 20) ||
      (blocks != 0 && (blocks < 8 || blocks > 256)))
    return 0;
  // the filter streams whole clouds through a workgroup: only where the unfiltered form would not split a cloud either
  const int cus = blocks > 0 ? blocks : 256;
  return B * (Cout / 256) >= cus ? 1 : 0;
}

extern "C" int64_t hitadv_linear_max_filter_scratch_words(int B, int Cout) {
  // theta [B, Cout] f32 | ccount [B, Cout / 32] i32 | cand [B, Cout / 32, CAP] u32
  return B > 0 && Cout > 0 ? (int64_t)B * Cout + (int64_t)B * (Cout / 32) * (1 + VF_CAP) : 0;
}

extern "C" int hitadv_linear_max_fwd_f16x2_filtered(const uint32_t *Xp, const uint16_t *W2, const float *wnorm, const float *bias,
                                                    int B, int N, int Cin, int Cout, int relu, int blocks, int64_t *seed,
                                                    uint32_t *scratch, float *out, int64_t *idx, int32_t *range_flag, void *stream) {
  if (!Xp || !W2 || !wnorm || !seed || !scratch || !out || !idx || !range_flag ||
      !hitadv_linear_max_filter_supported(B, N, Cin, Cout, blocks) || ((uintptr_t)Xp & 15) || ((uintptr_t)W2 & 15))
    return HITADV_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  float *theta = reinterpret_cast<float *>(scratch);
  int32_t *ccount = reinterpret_cast<int32_t *>(scratch + (size_t)B * Cout);
  uint32_t *cand = scratch + (size_t)B * Cout + (size_t)B * (Cout / 32);
  const dim3 small((Cout / 32 + 3) / 4, B);
  vf_seed_k<<<small, 256, 0, s>>>(Xp, W2, N, Cout, seed, theta);
  const int ncg = Cout / 256, cpb = vf_clouds_per_block(B, Cout, blocks);
  const int shm = 2 * VF_PIECE + 8 * VF_CAP * 4;
  const dim3 grid(ncg * ((B + cpb - 1) / cpb));
#define HITADV_VF_LAUNCH(ABL_)                                                                                           \
  do {                                                                                                                   \
    HITADV_RAISE_LDS((&vf_stream_k<ABL_>), shm);                                                                         \
    vf_stream_k<ABL_><<<grid, 512, shm, s>>>(Xp, W2, wnorm, B, N, Cout, ncg, cpb, theta, cand, ccount, range_flag);       \
  } while (0)
  switch (g_vf_ablate) {
    case 1: HITADV_VF_LAUNCH(1); break;
    case 2: HITADV_VF_LAUNCH(2); break;
    case 3: HITADV_VF_LAUNCH(3); break;
    case 4: HITADV_VF_LAUNCH(4); break;
    case 5: HITADV_VF_LAUNCH(5); break;
    default: HITADV_VF_LAUNCH(0);
  }
#undef HITADV_VF_LAUNCH
  vf_refine_k<<<small, 256, 0, s>>>(Xp, W2, bias, N, Cout, relu, cand, ccount, out, idx, seed);
  HITADV_LAUNCH_CHECK();
  return 0;
}
