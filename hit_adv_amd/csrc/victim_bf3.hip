// V1 on the bf16 matrix cores at fp32 accuracy: y = x[B*N,CIN] @ Wt[CIN,Cout] -> max / arg-max over the N points of each
// cloud (PointNet's 128 -> 1024 shared layer + global max-pool, model/feature_models.py:113-147,165-177), with every
// fp32 operand carried as THREE bf16 pieces.
//
// Why: gfx950 has no reduced-precision f32 MFMA; v_mfma_f32_32x32x2_f32 runs at the f32 vector rate (64 flop / cycle /
// SIMD, csrc/victim.hip reaches 82 % of it).  The bf16 MFMAs run 16x faster.  An fp32 value splits EXACTLY into
// three bf16 numbers, a = a1 + a2 + a3 (8 + 8 + 8 significant bits, each piece the truncation of what the previous ones
// left), so   a.b = a1.b1 + (a1.b2 + a2.b1) + (a1.b3 + a3.b1 + a2.b2) + [terms below 2^-24 |a.b|],
// six bf16 MFMAs (every bf16 x bf16 product is exact in fp32, the accumulator is fp32) instead of eight f32 MFMAs per 16
// values of k: 2.67x less matrix time at the accuracy of an fp32 GEMM (measured against float64: the dropped terms are
// 4e-8 of the output scale, the fp32 accumulation itself 4e-7 -- tests/test_gpu_attack.py::test_pointnet_engine_*).
// This is NOT bf16 precision: nothing is rounded to 8 bits.
//
//   block   = 8 waves (two per SIMD); one cloud, one split of its points, 256 output channels (32 per wave)
//   W       : split once per attack (weights are constants) into three bf16 images in fragment order; the wave's 32
//             columns x CIN rows x 3 pieces live in VGPRs for the whole kernel (96 registers at CIN = 128)
//   x       : 64-point tiles, global fp32 -> registers -> split (4 VALU per value) -> three bf16 LDS images (row stride
//             2 CIN + 32 bytes: conflict-free ds_read_b128 for the fragment pattern below), double buffered, one barrier
//             per tile
//   compute : v_mfma_f32_16x16x32_bf16 (the chip holds a higher clock under it than under the 32x32x16 form, and four
//             accumulators rotate): per tile and wave 4 row tiles x 2 column tiles x (CIN/32 slices) x 6 MFMAs; the next
//             unit's A fragments are read while the current unit's 24 MFMAs run
//   epilogue: a lane scans its two channels' 16 points per tile into a running (max, first arg-max): the tile maximum by
//             v_max3, then the first value equal to it; four lane groups and the splits merge at the end (the last block
//             to arrive, as in the f32 kernel).
//   FLAT    : (template parameter, chosen by the launcher when there is no split and N is a multiple of 128) the clouds of a
//             workgroup as one stream of tiles, no ragged-tile or merge code in the instantiation.
// Where the time goes.  Round 2 (tools/tune/v1bf3: variants of this file with one cost removed, in-kernel clock stamped): the
// MFMAs alone (no split, no scan, A fragments read once) take 36 us of the 40 of the bf16x3 form.  Round 4: on this part the
// VECTOR instructions are not hidden behind the matrix pipe, also across the two waves of a SIMD, and the compiler had doubled
// them (a bounds test if-converted into two selects per value; compare -> select hazards); 281 -> 137 per tile and wave took
// the fp16x2 form from 330 to 256 us per 256 clouds on 128 workgroups (docs/kernels/round4.md section 8).
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"
#include "hitadv.h"

namespace hitadv {

typedef float f32x4b __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f16x8 as_f16x8(uint4 u) { return __builtin_bit_cast(f16x8, u); }
constexpr int B3_TM = 64;

// MODE 1 -- the same kernel on the fp16 matrix cores with TWO pieces per operand (csrc header of hitadv_linear_max_fwd_f16x2):
//   a = a1 + 2^-11 a2 + ra,  a1 = fp16(a) (round to nearest: |a - a1| <= 2^-12 |a|, and a - a1 is exact in fp32),
//   a2 = fp16(2^11 (a - a1)) (the residual scaled back into fp16's normal range), |ra| <= 2^-24 |a| for most a
//   (each rounding is half an ulp of an 11-bit significand: 2^-12 of its operand in the mean, 2^-11 at a binade's lower edge, so the
//   bound proper is 2^-22 |a| for 2^-14 <= |a| < 65520 -- below fp16's normal range the pieces' own subnormal spacing takes over: |ra| <= 2^-36
//   absolute --; root mean square 2^-24.4, 99.9th percentile 2^-22.5: tests/test_arith_host.py, on the host build of csrc/arith.hpp::split_pair),
// so  a.b = a1.b1 + 2^-11 (a1.b2 + a2.b1) + [a2.b2 2^-22 + ra.b + a.rb: each <= 2^-24 |a.b|, fp32's own unit roundoff]:
// THREE fp16 MFMAs (every fp16 x fp16 product is exact in fp32) into two fp32 accumulator sets (one per power of two),
// joined once per tile -- half the matrix time of the six-product bf16 form.  fp16 tops out at 65504: an operand beyond
// that raises the caller's range flag (the engine then refuses the result; `matrix_mode = 'bf16x3'` has fp32's range).
constexpr float F16X2_SCALE = 2048.f;  // 2^11

// Two compares into SGPR pairs / two selects on SGPR pairs, as written (see the scan of linear_max_fwd_bf3_k).
__device__ __forceinline__ void v1_cmp2(unsigned long long &ma, unsigned long long &mb, float a, float ta, float b, float tb) {
  asm volatile("v_cmp_eq_f32_e64 %0, %2, %3\n\tv_cmp_eq_f32_e64 %1, %4, %5" : "=s"(ma), "=s"(mb) : "v"(a), "v"(ta), "v"(b), "v"(tb));
}
template <int Q>
__device__ __forceinline__ void v1_sel2(int &ca, int &cb, unsigned long long ma, unsigned long long mb) {
  asm volatile("v_cndmask_b32_e64 %0, %0, %4, %2\n\tv_cndmask_b32_e64 %1, %1, %4, %3" : "+v"(ca), "+v"(cb) : "s"(ma), "s"(mb), "n"(Q));
}

// FLAT (chosen by the launcher when S == 1 and N is a multiple of 128, i.e. every tile is whole and a cloud is an even number
// of them): the clouds of a workgroup are ONE stream of tiles -- they are contiguous in X --, the running maximum is finished
// and started again every N / 64 tiles, and the tile pipeline is neither drained nor refilled at a cloud boundary (3.3 us each
// at N = 1024, tools/v1_bubble_probe.py).  The instantiation holds no ragged-tile or split-merge code at all.
// DEFER (round 6, NOT YET RUN ON A GPU; off unless HITADV_V1_DEFER=1 -- see the launcher): the search for the tile's first arg-max (30
// compares + 32 selects per tile and wave, the largest single item of the tile's ~148 vector instructions) leaves the tile.  A lane keeps,
// per channel, the 16 joined values of the LAST TILE THAT IMPROVED its running maximum (16 selects under the improvement mask) and that
// tile's number; the search runs once per cloud, in finish(), over the kept values.  The same (value, point): the kept tile is the one
// whose maximum the running maximum equals, and "first of its 16 values equal to it" is what the per-tile search computed.
template <int CIN, int MODE, bool FLAT, bool DEFER = false>
__global__ __launch_bounds__(512) void linear_max_fwd_bf3_k(const float *__restrict__ X, const uint16_t *__restrict__ W3,
                                                            int B_, int N, int Cout, int rows_per_split, int S, int ncg, int cpb,
                                                            float *pval, int32_t *pidx, const float *__restrict__ bias,
                                                            int relu, float *__restrict__ out, int64_t *__restrict__ idx,
                                                            int *tickets, int *range_flag) {
  constexpr int NP = MODE >= 1 ? 2 : 3;  // pieces per operand (MODE 2: fp16x2 whose input arrives as packed pieces)
  constexpr int NSL = CIN / 32;          // 32-deep MFMA slices
  constexpr int RS = 2 * CIN + 32;       // bytes per LDS row of one piece: rows 2 x 16 bytes apart mod 256 make the
                                         // 16x16x32 A-fragment reads (lane -> row lane % 16, chunk lane / 16) conflict-free
  constexpr int PIECE = B3_TM * RS;      // bytes per piece image
  constexpr int G8 = CIN / 8;            // groups of 8 consecutive k per row
  constexpr int ST = B3_TM * G8 / 512;   // 8-value groups staged per thread per tile
  extern __shared__ __attribute__((aligned(16))) char sB3[];  // 2 buffers x NP pieces x PIECE
  int cg, s, b;
  {  // XCD-aware block order (see linear_max_fwd_k): the column-group blocks that stream the same x tiles share an XCD
    const int NCG = ncg, id = blockIdx.x, nrg = S * ((B_ + cpb - 1) / cpb);  // cpb > 1 (clouds per block) only with S == 1
    if ((nrg & 7) == 0) {
      const int xcd = id & 7, slot = id >> 3;
      cg = slot % NCG;
      const int rg = (slot / NCG) * 8 + xcd;
      s = rg % S;
      b = rg / S;
    } else {
      cg = id % NCG;
      s = (id / NCG) % S;
      b = id / (NCG * S);
    }
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col0 = cg * 256 + wave * 32;
  const bool active = col0 < Cout;  // wave-uniform
  const float *const X0 = X;
  const int b0 = b * cpb;  // this block's first cloud: it takes cpb of them one after the other with the same W in registers
  const int nb = min(cpb, B_ - b0);
  const int n0 = FLAT ? 0 : s * rows_per_split;  // rows of a pass, relative to its X
  const int n1 = FLAT ? nb * N : min(N, n0 + rows_per_split);
  const int ntiles = (n1 - n0 + B3_TM - 1) / B3_TM;
  int tbase = 0;  // FLAT: the stream's tile number of the current cloud's first tile

  // B operand of slice j (32 values of k), column tile ct (16 columns), piece p: the lane's column 16 ct + lane % 16,
  // k = 32 j + 8 (lane / 16) .. + 7.  W3 is stored in fragment order [piece][16-column block][slice][lane] x 16 bytes, so
  // every load instruction of a wave reads 1 KB contiguous.
  uint4 w[NP][2][NSL];
  {
    const uint4 *wp = reinterpret_cast<const uint4 *>(W3) + (size_t)((active ? col0 : 0) / 16) * NSL * 64 + lane;
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int j = 0; j < NSL; ++j) w[p][ct][j] = wp[((size_t)p * (Cout / 16) * NSL + ct * NSL + j) * 64];
  }

  // x tiles travel global -> registers -> (split) -> LDS.  Two register sets: the loads of tile t+2 are issued at the top
  // of tile t and written to LDS one tile later, so no wave ever waits on HBM.  The two waves of a SIMD are staggered
  // (waves 4-7 convert and store tile t+1 BEFORE their MFMAs of tile t, waves 0-3 after): while one of them is in its
  // vector phase (split + scan) the other owns the matrix pipe (MI355X_MICROARCH.md, "Two waves per SIMD", item 9).
  float4 stA[ST][2], stB[ST][2];
  // fetch() only issues the loads (rows past the split read a valid address); the first instruction that touches the
  // loaded registers is in stash(), a whole tile later -- a select here would make the wave wait for HBM among its MFMAs.
  uint32_t soff[ST];
#pragma unroll
  for (int u = 0; u < ST; ++u) {
    const int e = threadIdx.x + 512 * u;
    soff[u] = (uint32_t)((e / G8) * CIN + 8 * (e % G8)) * 4u;
  }
  auto fetch = [&](float4 (&st)[ST][2], int tile, auto full_c) {
    if constexpr (decltype(full_c)::value) {
      const char *tb = reinterpret_cast<const char *>(X) + (size_t)(n0 + tile * B3_TM) * CIN * 4;
#pragma unroll
      for (int u = 0; u < ST; ++u) {
        st[u][0] = *reinterpret_cast<const float4 *>(tb + soff[u]);
        st[u][1] = *reinterpret_cast<const float4 *>(tb + soff[u] + 16);
      }
    } else {
#pragma unroll
      for (int u = 0; u < ST; ++u) {
        const int e = threadIdx.x + 512 * u;
        const int n = n0 + tile * B3_TM + e / G8;
        const float *sp = n < n1 ? X + (size_t)n * CIN + 8 * (e % G8) : X;
        st[u][0] = *reinterpret_cast<const float4 *>(sp);
        st[u][1] = *reinterpret_cast<const float4 *>(sp + 4);
      }
    }
  };
  auto stash = [&](const float4 (&st)[ST][2], int tile, auto full_c) {
    const int buf = tile & 1;
#pragma unroll
    for (int u = 0; u < ST; ++u) {
      const int e = threadIdx.x + 512 * u;
      float a[8] = {st[u][0].x, st[u][0].y, st[u][0].z, st[u][0].w, st[u][1].x, st[u][1].y, st[u][1].z, st[u][1].w};
      if constexpr (!decltype(full_c)::value) {
        const bool in = n0 + tile * B3_TM + e / G8 < n1;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = in ? a[i] : 0.f;
      }
      char *dst = sB3 + (size_t)buf * NP * PIECE + (e / G8) * RS + 16 * (e % G8);
      if constexpr (MODE == 2) {
        // X holds one word per value, (fp16 hi | fp16 lo << 16), written by the layer in front (csrc/pointnet.hip, pack_o2): two
        // byte permutes per pair of values put the pieces into the two images; rows past the split are zero words
        const uint32_t wd[8] = {__float_as_uint(a[0]), __float_as_uint(a[1]), __float_as_uint(a[2]), __float_as_uint(a[3]),
                                __float_as_uint(a[4]), __float_as_uint(a[5]), __float_as_uint(a[6]), __float_as_uint(a[7])};
        uint4 hi, lo;
        hi.x = __builtin_amdgcn_perm(wd[1], wd[0], 0x05040100u);
        lo.x = __builtin_amdgcn_perm(wd[1], wd[0], 0x07060302u);
        hi.y = __builtin_amdgcn_perm(wd[3], wd[2], 0x05040100u);
        lo.y = __builtin_amdgcn_perm(wd[3], wd[2], 0x07060302u);
        hi.z = __builtin_amdgcn_perm(wd[5], wd[4], 0x05040100u);
        lo.z = __builtin_amdgcn_perm(wd[5], wd[4], 0x07060302u);
        hi.w = __builtin_amdgcn_perm(wd[7], wd[6], 0x05040100u);
        lo.w = __builtin_amdgcn_perm(wd[7], wd[6], 0x07060302u);
        *reinterpret_cast<uint4 *>(dst) = hi;
        *reinterpret_cast<uint4 *>(dst + PIECE) = lo;
      } else if constexpr (MODE == 1) {
        f16x8 h1, h2;
        RangeWatch big;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          h1[i] = (_Float16)a[i];  // round to nearest even
          h2[i] = (_Float16)((a[i] - (float)h1[i]) * F16X2_SCALE);
          big.see(a[i]);
        }
        if (big.beyond_fp16() && range_flag != nullptr) *range_flag = 1;  // beyond fp16 (or NaN): the caller refuses the result
        *reinterpret_cast<uint4 *>(dst) = __builtin_bit_cast(uint4, h1);
        *reinterpret_cast<uint4 *>(dst + PIECE) = __builtin_bit_cast(uint4, h2);
      } else {
        uint32_t p1[8], p2[8], p3[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) split3(a[i], p1[i], p2[i], p3[i]);
        *reinterpret_cast<uint4 *>(dst) = make_uint4(pack_hi(p1[0], p1[1]), pack_hi(p1[2], p1[3]), pack_hi(p1[4], p1[5]),
                                                     pack_hi(p1[6], p1[7]));
        *reinterpret_cast<uint4 *>(dst + PIECE) = make_uint4(pack_hi(p2[0], p2[1]), pack_hi(p2[2], p2[3]),
                                                             pack_hi(p2[4], p2[5]), pack_hi(p2[6], p2[7]));
        *reinterpret_cast<uint4 *>(dst + (NP - 1) * PIECE) = make_uint4(pack_hi(p3[0], p3[1]), pack_hi(p3[2], p3[3]),
                                                                        pack_hi(p3[4], p3[5]), pack_hi(p3[6], p3[7]));
      }
    }
  };

  // v_mfma_f32_16x16x32_bf16: the same flops per cycle as the 32x32x16 form on paper, but the chip holds a higher clock
  // under it (MI355X_MICROARCH.md, DVFS item 7: ~1.13x the rate on random data) -- measured here: tools/tune/v1bf3.
  // Accumulators: acc[rt][ct], row tile rt (16 points) x column tile ct (16 channels); element i of a lane = point
  // 16 rt + 4 (lane / 16) + i, channel 16 ct + lane % 16.  A lane therefore scans TWO channels.
  float bv[2] = {-__builtin_inff(), -__builtin_inff()};
  int bi[2] = {-1, -1};
  float sv[2][16];          // DEFER: the joined values of the last tile that improved bv (per channel)
  int bt[2] = {-1, -1};     // DEFER: that tile's number (FLAT: within its cloud); -1 = none yet
  if constexpr (DEFER) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 16; ++q) sv[ct][q] = 0.f;
  }
  const int l16 = lane & 15, g4 = lane >> 4;
  const bool late = wave >= 4;  // wave-uniform
  // one tile: 8 units of (slice j, row-tile pair rp): 6 A fragments (2 row tiles x 3 pieces) feed 24 MFMAs; the next
  // unit's fragments are read from LDS while this unit's MFMAs run.  Then the scan.
  constexpr int NL = MODE >= 1 ? 4 : 1;  // fp16x2: accl collects the 2^-11 terms
  auto products = [&](int tile, f32x4b (&acc)[4][2], f32x4b (&accl)[NL][2]) {
    const char *base = sB3 + (size_t)(tile & 1) * NP * PIECE + l16 * RS + 16 * g4;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        acc[rt][ct] = f32x4b{0.f, 0.f, 0.f, 0.f};
        if constexpr (MODE >= 1) accl[rt][ct] = f32x4b{0.f, 0.f, 0.f, 0.f};
      }
    uint4 fa[2][2 * NP];  // [buffer][NP * (row tile within the pair) + piece]
    auto frag = [&](int u, int q) {  // unit u = 2 j + rp
      return *reinterpret_cast<const uint4 *>(base + (q % NP) * PIECE + (2 * (u & 1) + q / NP) * 16 * RS + 64 * (u >> 1));
    };
#pragma unroll
    for (int q = 0; q < 2 * NP; ++q) fa[0][q] = frag(0, q);
#pragma unroll
    for (int u = 0; u < 2 * NSL; ++u) {
      if (u + 1 < 2 * NSL) {
#pragma unroll
        for (int q = 0; q < 2 * NP; ++q) fa[(u + 1) & 1][q] = frag(u + 1, q);
      }
      __builtin_amdgcn_sched_barrier(0);  // the reads stay above this unit's MFMAs
      const int j = u >> 1, rp = u & 1;
      if constexpr (MODE >= 1) {
        f16x8 a[2][2], b[2][2];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            a[x][p] = as_f16x8(fa[u & 1][2 * x + p]);
            b[x][p] = as_f16x8(w[p][x][j]);
          }
        // the eight accumulators of the unit take turns, so consecutive MFMAs are independent
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
              if (t == 0) accl[2 * rp + x][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[x][1], b[ct][0], accl[2 * rp + x][ct], 0, 0, 0);
              if (t == 1) accl[2 * rp + x][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[x][0], b[ct][1], accl[2 * rp + x][ct], 0, 0, 0);
              if (t == 2) acc[2 * rp + x][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[x][0], b[ct][0], acc[2 * rp + x][ct], 0, 0, 0);
            }
      } else {
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            a[x][p] = as_bf16x8(fa[u & 1][NP * x + p]);
            b[x][p] = as_bf16x8(w[p][x][j]);
          }
        // smallest terms first; the four accumulators of the unit take turns, so consecutive MFMAs are independent
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
              acc[2 * rp + x][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[x][PA[t]], b[ct][PB[t]], acc[2 * rp + x][ct], 0, 0, 0);
      }
    }
  };
  auto scan = [&](int tile, const f32x4b (&acc)[4][2], const f32x4b (&accl)[NL][2], auto full_c) {
    constexpr bool ragged = !decltype(full_c)::value;
    const int row0 = n0 + tile * B3_TM + 4 * g4;
    // the tile's maximum first (v_max3: half an instruction per value, no dependent compare / select chain), then the FIRST of
    // the 16 values equal to it -- the same (value, point) as a running strict ">" scan keeps.  The two channels' searches are
    // written interleaved and pinned in front of their use: left to itself the compiler moves each into a branch of its own
    // (taken whenever any lane improves, i.e. always) where every compare -> select pair waits out its hazard alone.
    float v[2][16], tv[2];
    int tc[2] = {15, 15};
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float y = acc[rt][ct][i];
          if constexpr (MODE >= 1) y = fmaf(accl[rt][ct][i], 1.f / F16X2_SCALE, y);  // the two powers of two meet
          if constexpr (ragged) y = row0 + 16 * rt + i < n1 ? y : -__builtin_inff();  // zero-filled rows stay out
          v[ct][4 * rt + i] = y;
        }
      tv[ct] = -__builtin_inff();
#pragma unroll
      for (int q = 0; q < 16; q += 2) tv[ct] = __builtin_fmaxf(__builtin_fmaxf(tv[ct], v[ct][q]), v[ct][q + 1]);
    }
    if constexpr (DEFER) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const bool g = tv[ct] > bv[ct];  // earlier tiles hold earlier points: they keep ties
        bv[ct] = g ? tv[ct] : bv[ct];
        bt[ct] = g ? (FLAT ? tile - tbase : tile) : bt[ct];
#pragma unroll
        for (int q = 0; q < 16; ++q) sv[ct][q] = g ? v[ct][q] : sv[ct][q];
      }
      return;
    }
    // tc = v[q] == tv ? q : tc for q = 14 .. 0, written out: the compiler sends every compare through VCC and pays the
    // compare -> select hazard (2 wait states) thirty times per tile; here the compares run two steps ahead of the selects in
    // four SGPR pairs, so each result is five instructions old when it is read.
    {
      unsigned long long m0, m1, m2, m3;
      v1_cmp2(m0, m1, v[0][14], tv[0], v[1][14], tv[1]);
      v1_cmp2(m2, m3, v[0][13], tv[0], v[1][13], tv[1]);
#define HITADV_STEP(q, a, b)         \
  v1_sel2<q>(tc[0], tc[1], a, b);    \
  v1_cmp2(a, b, v[0][q - 2], tv[0], v[1][q - 2], tv[1])
      HITADV_STEP(14, m0, m1);
      HITADV_STEP(13, m2, m3);
      HITADV_STEP(12, m0, m1);
      HITADV_STEP(11, m2, m3);
      HITADV_STEP(10, m0, m1);
      HITADV_STEP(9, m2, m3);
      HITADV_STEP(8, m0, m1);
      HITADV_STEP(7, m2, m3);
      HITADV_STEP(6, m0, m1);
      HITADV_STEP(5, m2, m3);
      HITADV_STEP(4, m0, m1);
      HITADV_STEP(3, m2, m3);
      HITADV_STEP(2, m0, m1);
#undef HITADV_STEP
      v1_sel2<1>(tc[0], tc[1], m2, m3);
      v1_sel2<0>(tc[0], tc[1], m0, m1);
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const bool g = tv[ct] > bv[ct];  // earlier tiles hold earlier points: they keep ties
      bv[ct] = g ? tv[ct] : bv[ct];
      bi[ct] = g ? (FLAT ? tile - tbase : tile) * 16 + tc[ct] : bi[ct];
    }
  };
  using Full = std::true_type;
  using Ragged = std::false_type;
  const int nfull = (n1 - n0) / B3_TM;
  auto compute = [&](int tile, auto full_c) {
    f32x4b acc[4][2], accl[NL][2];
    products(tile, acc, accl);
    scan(tile, acc, accl, full_c);
  };
  auto steady = [&](int tile, float4 (&have)[ST][2], float4 (&next)[ST][2]) {
    // FLAT runs every tile through here: past the end of the stream the last tile is fetched (and written to the LDS buffer
    // nobody reads any more) again
    fetch(next, FLAT ? min(tile + 2, ntiles - 1) : tile + 2, Full{});
    if (late) stash(have, tile + 1, Full{});
    if (active) compute(tile, Full{});
    if (!late) stash(have, tile + 1, Full{});
    __syncthreads();
  };
  auto step = [&](int tile, float4 (&have)[ST][2], float4 (&next)[ST][2]) {
    if (tile + 2 < nfull) return steady(tile, have, next);
    const bool more = tile + 1 < ntiles;
    if (tile + 2 < ntiles) fetch(next, tile + 2, Ragged{});
    if (more && late) stash(have, tile + 1, Ragged{});
    if (active) compute(tile, Ragged{});  // (the bounds tests also pass every row of a whole tile)
    if (more && !late) stash(have, tile + 1, Ragged{});
    __syncthreads();
  };
  // a cloud's (split's) last tile has been scanned: code -> point index (nothing won -- all rows -inf --: the pass's first
  // point); the four 16-lane groups of the wave hold the same channels, other points: two exchanges, the lower point keeps
  // a tie; then the result (S == 1) or this split's partial leaves, and the running maximum starts again
  auto finish = [&]() {
    if constexpr (DEFER) {  // the search the tiles skipped: the first of the kept tile's 16 values equal to the running maximum
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        int tc = 15;
#pragma unroll
        for (int q = 14; q >= 0; --q) tc = sv[ct][q] == bv[ct] ? q : tc;
        bi[ct] = bt[ct] < 0 ? -1 : bt[ct] * 16 + tc;
        bt[ct] = -1;
      }
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      bi[ct] = bi[ct] < 0 ? n0 : n0 + (bi[ct] >> 4) * B3_TM + 16 * ((bi[ct] >> 2) & 3) + 4 * g4 + (bi[ct] & 3);
#pragma unroll
      for (int m = 16; m <= 32; m <<= 1) {
        const float ov = __shfl_xor(bv[ct], m, HITADV_WAVE);
        const int oi = __shfl_xor(bi[ct], m, HITADV_WAVE);
        if (ov > bv[ct] || (ov == bv[ct] && oi < bi[ct])) { bv[ct] = ov; bi[ct] = oi; }
      }
    }
    if (active && g4 == 0) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int c = col0 + 16 * ct + l16;
        if (FLAT || S == 1) {  // nothing to merge: finish here
          float v = bv[ct] + (bias ? bias[c] : 0.f);  // rounding is monotonic: max_n(y_n + b) == max_n(y_n) + b
          out[(size_t)b * Cout + c] = relu ? (v > 0.f ? v : 0.f) : v;  // max and ReLU commute
          idx[(size_t)b * Cout + c] = max(bi[ct], 0);  // (-1 only if every value was NaN: the range flag is up, the table stays valid)
        } else {
          const size_t o = ((size_t)b * S + s) * Cout + c;
          __hip_atomic_store(&pval[o], bv[ct], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&pidx[o], bi[ct], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    bv[0] = bv[1] = -__builtin_inff();
    bi[0] = bi[1] = -1;
  };
  for (int bb = 0; bb < (FLAT ? 1 : nb); ++bb) {
    b = b0 + bb;  // the pass's first cloud
    X = X0 + (size_t)b * N * CIN;
    if (bb > 0) __syncthreads();  // the previous cloud's last tile is no longer being read
    fetch(stA, 0, Ragged{});
    // every load issued so far (W, tile 0) completes HERE, explicitly: the first use of W is inside the loop, and a load that
    // may still be pending at the loop entry makes the compiler's wait-count pass guard every in-loop use of W with a
    // vmcnt wait that, on the iterations that did issue new loads, waits for THOSE (measured: tools/tune/v1bf3)
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    stash(stA, 0, Ragged{});
    if (ntiles > 1) fetch(stA, 1, Ragged{});
    __syncthreads();
    if constexpr (FLAT) {
      const int tpc = N / B3_TM;  // even
      for (tbase = 0; tbase < ntiles; tbase += tpc, ++b) {
        for (int t = tbase; t < tbase + tpc; t += 2) {
          steady(t, stA, stB);
          steady(t + 1, stB, stA);
        }
        finish();
      }
    } else {
      for (int tile = 0; tile < ntiles; tile += 2) {
        step(tile, stA, stB);
        if (tile + 1 < ntiles) step(tile + 1, stB, stA);
      }
      finish();
    }
  }  // passes of this block
  if (FLAT || S == 1) return;
  b = b0;  // (S > 1: one cloud per block)
  // The S splits of a (cloud, column group) meet here: the last block to draw the group's ticket merges the partials in
  // split order (= ascending points, so ties keep the first point).  Hand-off protocol of fc_layer_k (csrc/pointnet.hip).
  __shared__ int s_last;
  if (!handoff_last_arriver(tickets, b * ncg + cg, S, &s_last)) return;
  const int c = cg * 256 + threadIdx.x;
  if (threadIdx.x < 256 && c < Cout) {
    float best = 0.f;
    int bidx = 0;
    for (int q0 = 0; q0 < S; q0 += 4) {  // four splits' partials per round trip, compared in split order
      float v[4];
      int vi[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t o = ((size_t)b * S + min(q0 + u, S - 1)) * Cout + c;
        v[u] = __hip_atomic_load(&pval[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        vi[u] = __hip_atomic_load(&pidx[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (q0 + u < S && (q0 + u == 0 || v[u] > best)) { best = v[u]; bidx = vi[u]; }
    }
    best += bias ? bias[c] : 0.f;
    out[(size_t)b * Cout + c] = relu ? (best > 0.f ? best : 0.f) : best;
    idx[(size_t)b * Cout + c] = max(bidx, 0);
  }
  if (threadIdx.x == 0) __hip_atomic_store(&tickets[b * ncg + cg], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// W[Cout,Cin] fp32 (row-major, one row per output channel) -> its pieces in FRAGMENT ORDER:
// W3[piece][c / 16][k / 32][(k % 32) / 8][c % 16][k % 8]  (16 bytes per (piece, 16-column block, slice, lane)).
// MODE 0: three bf16 pieces (exact); MODE 1: two fp16 pieces, the second scaled by 2^11 (see the kernel's header).
template <int MODE>
__global__ __launch_bounds__(256) void split_weights_k(const float *__restrict__ W, uint16_t *__restrict__ W3, int Cout,
                                                       int Cin, int *range_flag) {
  const long long total = (long long)Cout * Cin;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e / Cin), k = (int)(e % Cin);
  const long long o = ((((long long)(c / 16) * (Cin / 32) + k / 32) * 4 + (k % 32) / 8) * 16 + c % 16) * 8 + k % 8;
  if (MODE == 1) {
    const float a = W[e];
    const _Float16 h1 = (_Float16)a;
    const _Float16 h2 = (_Float16)((a - (float)h1) * F16X2_SCALE);
    if (!(fabsf(a) < 65504.f) && range_flag != nullptr) *range_flag = 1;
    W3[o] = __builtin_bit_cast(uint16_t, h1);
    W3[total + o] = __builtin_bit_cast(uint16_t, h2);
    return;
  }
  uint32_t a, bb, cc;
  split3(W[e], a, bb, cc);
  W3[o] = (uint16_t)(a >> 16);
  W3[total + o] = (uint16_t)(bb >> 16);
  W3[2 * total + o] = (uint16_t)(cc >> 16);
}

// Point splits per cloud so that the grid has about `cus` blocks (as linear_max_split, csrc/victim.hip: one block per CU;
// every block first loads its 24 KB / wave of W).  `blocks` (a per-call argument; 0 = HITADV_V1_CUS from the environment,
// read once, or 256) lowers the target: with several attacks in flight a V1 grid that leaves part of the chip to the other
// streams' latency-bound kernels is the better trade (tools/README.md); results do not depend on it (the split merge is
// in point order).  The scratch query and the launch derive S / rows / cpb from the SAME function of (B, N, Cout, blocks).
static int bf3_cus(int blocks) {
  if (blocks > 0) return blocks;
  static int cus = [] {
    const char *e = getenv("HITADV_V1_CUS");
    const int v = e ? atoi(e) : 256;
    return v >= 8 && v <= 256 ? v : 256;
  }();
  return cus;
}

static void bf3_split(int B, int N, int Cout, int blocks, int *S, int *rows, int *cpb) {
  const int colgroups = (Cout + 255) / 256;
  const int cus = bf3_cus(blocks);
  // fewer workgroups than (clouds x column groups): a block takes several clouds in turn (W stays in its registers)
  *cpb = cus < B * colgroups ? (B * colgroups + cus - 1) / cus : 1;
  int want = (cus + B * colgroups - 1) / (B * colgroups);
  const int maxs = (N + B3_TM - 1) / B3_TM;
  want = want < 1 ? 1 : (want > maxs ? maxs : want);
  int per = (N + want - 1) / want;
  per = (per + B3_TM - 1) / B3_TM * B3_TM;
  *rows = per;
  *S = (N + per - 1) / per;
}

}  // namespace hitadv

using namespace hitadv;

extern "C" int hitadv_split_weights_bf16x3(const float *W, int Cout, int Cin, uint16_t *W3, void *stream) {
  if (!W || !W3 || Cout <= 0 || Cin <= 0 || (Cout & 15) || (Cin & 31)) return HITADV_E_ARG;
  const long long total = (long long)Cout * Cin;
  split_weights_k<0><<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(W, W3, Cout, Cin, nullptr);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_split_weights_f16x2(const float *W, int Cout, int Cin, uint16_t *W2, int32_t *range_flag, void *stream) {
  if (!W || !W2 || Cout <= 0 || Cin <= 0 || (Cout & 15) || (Cin & 31)) return HITADV_E_ARG;
  const long long total = (long long)Cout * Cin;
  split_weights_k<1><<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(W, W2, Cout, Cin, range_flag);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t hitadv_linear_max_fwd_bf16x3_scratch(int B, int N, int Cout, int blocks) {
  if (B <= 0 || N <= 0 || Cout <= 0 || blocks < 0 || (blocks != 0 && (blocks < 8 || blocks > 256))) return 0;
  int S, rows, cpb;
  bf3_split(B, N, Cout, blocks, &S, &rows, &cpb);
  return (int64_t)B * S * Cout;
}

// 0 (default) = the per-tile arg-max search; 1 (HITADV_V1_DEFER=1, or hitadv_debug_v1_defer: A/B, tests) = the search deferred to the end
// of a cloud in the flat fp16x2 kernel on packed input (template parameter DEFER).  Same results, bit for bit, by construction -- and never
// run on a GPU: written in round 6, which had none (docs/kernels/round6.md section 4).
static int g_v1_defer = [] { const char *e = getenv("HITADV_V1_DEFER"); return e && e[0] == '1' ? 1 : 0; }();
extern "C" int hitadv_debug_v1_defer(int on) {
  const int old = g_v1_defer;
  if (on == 0 || on == 1) g_v1_defer = on;
  return old;
}

template <int MODE>
static int launch_linear_max_pieces(const float *X, const uint16_t *W3, const float *bias, int B, int N, int Cin, int Cout,
                                    int relu, int blocks, float *part_val, int32_t *part_idx, float *out, int64_t *idx,
                                    int32_t *tickets, int32_t *range_flag, void *stream) {
#ifdef HITADV_ABLATE
  {  // "v1" ablated: the arg-max table still has to hold valid point indices for the backward kernels that gather through it
    static const bool off__ = hitadv_ablated("v1");
    if (off__) {
      (void)hipMemsetAsync(idx, 0, sizeof(int64_t) * (size_t)B * Cout, (hipStream_t)stream);
      (void)hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * Cout, (hipStream_t)stream);
      return 0;
    }
  }
#endif
  if (!X || !W3 || !part_val || !part_idx || !out || !idx || !tickets || B <= 0 || N <= 0 || Cout <= 0 || (Cout & 63) ||
      (Cin != 64 && Cin != 128) || ((uintptr_t)X & 15) || ((uintptr_t)W3 & 15) || (blocks != 0 && (blocks < 8 || blocks > 256)))
    return HITADV_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  int S, rows, cpb;
  bf3_split(B, N, Cout, blocks, &S, &rows, &cpb);
  const int ncg = (Cout + 255) / 256;
  dim3 grid((unsigned)(ncg * S * ((B + cpb - 1) / cpb)));
  constexpr int NP = MODE >= 1 ? 2 : 3;
  const size_t shm = (size_t)2 * NP * B3_TM * (2 * Cin + 32);
  const bool flat = S == 1 && N % (2 * B3_TM) == 0;  // whole tiles, an even number per cloud: one stream of tiles per workgroup
#define HITADV_V1_LAUNCH(CIN_, FLAT_)                                                                                          \
  do {                                                                                                                         \
    HITADV_RAISE_LDS((&linear_max_fwd_bf3_k<CIN_, MODE, FLAT_>), 2 * NP * B3_TM * (2 * CIN_ + 32));                            \
    linear_max_fwd_bf3_k<CIN_, MODE, FLAT_><<<grid, 512, shm, s>>>(X, W3, B, N, Cout, rows, S, ncg, cpb, part_val, part_idx,   \
                                                                   bias, relu, out, idx, tickets, range_flag);                 \
  } while (0)
  if (Cin == 128) {
    bool deferred = false;
    if constexpr (MODE == 2) {  // (the stacked loop's kernel only -- the other modes are not instantiated; off by default: never run on a GPU yet)
      if (flat && g_v1_defer) {
        HITADV_RAISE_LDS((&linear_max_fwd_bf3_k<128, 2, true, true>), 2 * NP * B3_TM * (2 * 128 + 32));
        linear_max_fwd_bf3_k<128, 2, true, true><<<grid, 512, shm, s>>>(X, W3, B, N, Cout, rows, S, ncg, cpb, part_val, part_idx, bias,
                                                                        relu, out, idx, tickets, range_flag);
        deferred = true;
      }
    }
    if (deferred) {
    } else if (flat) HITADV_V1_LAUNCH(128, true);
    else HITADV_V1_LAUNCH(128, false);
  } else {
    if (flat) HITADV_V1_LAUNCH(64, true);
    else HITADV_V1_LAUNCH(64, false);
  }
#undef HITADV_V1_LAUNCH
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_linear_max_fwd_bf16x3(const float *X, const uint16_t *W3, const float *bias, int B, int N, int Cin,
                                            int Cout, int relu, int blocks, float *part_val, int32_t *part_idx, float *out,
                                            int64_t *idx, int32_t *tickets, void *stream) {
  return launch_linear_max_pieces<0>(X, W3, bias, B, N, Cin, Cout, relu, blocks, part_val, part_idx, out, idx, tickets, nullptr,
                                     stream);
}

extern "C" int hitadv_linear_max_fwd_f16x2_packed(const uint32_t *Xp, const uint16_t *W2, const float *bias, int B, int N, int Cin,
                                                  int Cout, int relu, int blocks, float *part_val, int32_t *part_idx, float *out,
                                                  int64_t *idx, int32_t *tickets, void *stream) {
  return launch_linear_max_pieces<2>(reinterpret_cast<const float *>(Xp), W2, bias, B, N, Cin, Cout, relu, blocks, part_val,
                                     part_idx, out, idx, tickets, nullptr, stream);
}

extern "C" int hitadv_linear_max_fwd_f16x2(const float *X, const uint16_t *W2, const float *bias, int B, int N, int Cin,
                                           int Cout, int relu, int blocks, float *part_val, int32_t *part_idx, float *out,
                                           int64_t *idx, int32_t *tickets, int32_t *range_flag, void *stream) {
  return launch_linear_max_pieces<1>(X, W2, bias, B, N, Cin, Cout, relu, blocks, part_val, part_idx, out, idx, tickets, range_flag,
                                     stream);
}
