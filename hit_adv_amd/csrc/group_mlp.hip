// Last shared layer of a sample-and-group block fused with the max over the neighbours (PointNet++ set abstraction,
// model/pointnet2_utils.py:197-201: conv -> bn -> relu -> torch.max(new_points, 2); PCT's Local_op, model/pct_cls.py:14-24),
// and its backward, for MANY small groups: G = B*npoint groups of NS = 32 or 64 rows each.
//
//   group_linear_max_fwd_k   out[g,c] = relu(max_j (x[g,j,:] . W[c,:]) + bias[c]),  arg[g,c] = the lowest such j
//                            V1's scheme (csrc/victim_bf3.hip, MODE 1) on a flat stream of 64-row tiles: x split into two
//                            fp16 pieces on the way into LDS, W's pieces in registers, three exact fp16 MFMAs per useful
//                            product into two fp32 accumulator sets, tiles double buffered.  A tile holds 64 / NS whole groups,
//                            so the max / arg-max scan needs no state across tiles: the [G*NS, Cout] activation (537 MB at
//                            cfg4's first level), its ReLU pass and its max pass never exist.
//   group_linear_max_bwd_k   dX[g*NS + j, :] = sum_{c : arg[g,c] == j, out[g,c] > 0} dOut[g,c] W[c,:]
//                            the max routes a channel's gradient to ONE row of its group: per tile the block scatters the
//                            (at most Cout per group) gradient values, split into two fp16 pieces, into an otherwise zero
//                            [64, Cout] A operand in LDS and multiplies it by W on the fp16 matrix cores -- the dense
//                            [G*NS, Cout] gradient (zero fill, scatter, ReLU mask) and the GEMM over it never exist.
// Error model: that of V1's fp16x2 form (two pieces per operand, products exact, terms below 2^-24 dropped).
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"
#include "hitadv.h"

namespace hitadv {

typedef float f32x4g __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8g __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f16x8g as_f16x8g(uint4 u) { return __builtin_bit_cast(f16x8g, u); }
constexpr int GM_TM = 64;             // rows per tile
constexpr float GM_SCALE = 2048.f;    // 2^11: the second piece is the residual scaled back into fp16's normal range

// ------------------------------------------------------------------------------------------------ forward
// block = 8 waves; wave w owns columns [16 NCT w, 16 NCT (w + 1)): Cout = 128 NCT.
template <int CIN, int NS, int NCT>
__global__ __launch_bounds__(512) void group_linear_max_fwd_k(const float *__restrict__ X, const uint16_t *__restrict__ W2,
                                                              const float *__restrict__ bias, long long rows, int tiles_per_block,
                                                              float *__restrict__ out, int32_t *__restrict__ arg, int *range_flag) {
  constexpr int COUT = 128 * NCT;
  constexpr int NSL = CIN / 32;
  constexpr int RS = 2 * CIN + 32;       // bytes per LDS row of one piece (conflict-free 16x16x32 A-fragment reads)
  constexpr int PIECE = GM_TM * RS;
  constexpr int G8 = CIN / 8;
  constexpr int ST = GM_TM * G8 / 512;   // 8-value groups staged per thread per tile
  extern __shared__ __attribute__((aligned(16))) char sG[];  // 2 buffers x 2 pieces x PIECE
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l16 = lane & 15, g4 = lane >> 4;
  const long long ntiles_all = (rows + GM_TM - 1) / GM_TM;
  const long long t0 = (long long)blockIdx.x * tiles_per_block;
  const int ntiles = (int)max(0ll, min((long long)tiles_per_block, ntiles_all - t0));
  if (ntiles <= 0) return;
  const long long n0 = t0 * GM_TM, n1 = min(rows, n0 + (long long)ntiles * GM_TM);
  const int col0 = wave * 16 * NCT;

  // W2 is in V1's fragment order [piece][c / 16][k / 32][lane] x 16 bytes
  uint4 w[2][NCT][NSL];
  {
    const uint4 *wp = reinterpret_cast<const uint4 *>(W2) + (size_t)(col0 / 16) * NSL * 64 + lane;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int j = 0; j < NSL; ++j) w[p][ct][j] = wp[((size_t)p * (COUT / 16) * NSL + ct * NSL + j) * 64];
  }
  float bv[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) bv[ct] = bias != nullptr ? bias[col0 + 16 * ct + l16] : 0.f;

  // Whole tiles and the (possibly ragged) last one are separate instantiations of fetch / stash: a run-time bounds test becomes a
  // select per VALUE, and on this part vector instructions are not hidden behind the matrix pipe (docs/kernels/round4.md section 8;
  // csrc/victim_bf3.hip).  The split: hi pieces two per v_cvt_pk_f16_f32, each lo piece one v_fma_mixlo/hi on the packed hi piece
  // (hi (-2048) + 2048 v is exact before its single rounding: the same bits as convert back / subtract / scale / convert); the
  // range watch looks at the packed hi pieces (an infinity or a NaN there is exactly when the split breaks down).
  using Full = std::true_type;
  using Ragged = std::false_type;
  const int nfull = (int)((n1 - n0) / GM_TM);  // tiles wholly inside the matrix
  uint32_t soff[ST];
#pragma unroll
  for (int u = 0; u < ST; ++u) {
    const int e = threadIdx.x + 512 * u;
    soff[u] = (uint32_t)((e / G8) * CIN + 8 * (e % G8)) * 4u;
  }
  float4 stA[ST][2], stB[ST][2];
  auto fetch = [&](float4 (&st)[ST][2], int tile, auto full_c) {
    if constexpr (decltype(full_c)::value) {
      const char *tb = reinterpret_cast<const char *>(X) + (size_t)(n0 + (long long)tile * GM_TM) * CIN * 4;
#pragma unroll
      for (int u = 0; u < ST; ++u) {
        st[u][0] = *reinterpret_cast<const float4 *>(tb + soff[u]);
        st[u][1] = *reinterpret_cast<const float4 *>(tb + soff[u] + 16);
      }
    } else {
#pragma unroll
      for (int u = 0; u < ST; ++u) {
        const int e = threadIdx.x + 512 * u;
        const long long n = n0 + (long long)tile * GM_TM + e / G8;
        const float *sp = n < n1 ? X + (size_t)n * CIN + 8 * (e % G8) : X;
        st[u][0] = *reinterpret_cast<const float4 *>(sp);
        st[u][1] = *reinterpret_cast<const float4 *>(sp + 4);
      }
    }
  };
  PieceWatch big;
  auto stash = [&](const float4 (&st)[ST][2], int tile, auto full_c) {
    const int buf = tile & 1;
    const float nsc = -GM_SCALE;
#pragma unroll
    for (int u = 0; u < ST; ++u) {
      const int e = threadIdx.x + 512 * u;
      float a[8] = {st[u][0].x, st[u][0].y, st[u][0].z, st[u][0].w, st[u][1].x, st[u][1].y, st[u][1].z, st[u][1].w};
      if constexpr (!decltype(full_c)::value) {
        const bool in = n0 + (long long)tile * GM_TM + e / G8 < n1;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = in ? a[i] : 0.f;
      }
      uint32_t H[4], L[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float s0 = a[2 * p] * GM_SCALE, s1 = a[2 * p + 1] * GM_SCALE;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(H[p]) : "v"(a[2 * p]), "v"(a[2 * p + 1]));
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(L[p]) : "v"(H[p]), "s"(nsc), "v"(s0));
        asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(L[p]) : "v"(H[p]), "s"(nsc), "v"(s1));
        big.see_f16x2(H[p]);
      }
      char *dst = sG + (size_t)buf * 2 * PIECE + (e / G8) * RS + 16 * (e % G8);
      *reinterpret_cast<uint4 *>(dst) = make_uint4(H[0], H[1], H[2], H[3]);
      *reinterpret_cast<uint4 *>(dst + PIECE) = make_uint4(L[0], L[1], L[2], L[3]);
    }
  };
  const bool late = wave >= 4;
  auto compute = [&](int tile) {
    const char *base = sG + (size_t)(tile & 1) * 2 * PIECE + l16 * RS + 16 * g4;
    f32x4g acc[4][NCT], accl[4][NCT];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        acc[rt][ct] = f32x4g{0.f, 0.f, 0.f, 0.f};
        accl[rt][ct] = f32x4g{0.f, 0.f, 0.f, 0.f};
      }
    uint4 fa[2][4];  // [buffer][2 * (row tile within the pair) + piece]
    auto frag = [&](int u, int q) {
      return *reinterpret_cast<const uint4 *>(base + (q % 2) * PIECE + (2 * (u & 1) + q / 2) * 16 * RS + 64 * (u >> 1));
    };
#pragma unroll
    for (int q = 0; q < 4; ++q) fa[0][q] = frag(0, q);
#pragma unroll
    for (int u = 0; u < 2 * NSL; ++u) {
      if (u + 1 < 2 * NSL) {
#pragma unroll
        for (int q = 0; q < 4; ++q) fa[(u + 1) & 1][q] = frag(u + 1, q);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int j = u >> 1, rp = u & 1;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) {
            const f16x8g ahi = as_f16x8g(fa[u & 1][2 * x]), alo = as_f16x8g(fa[u & 1][2 * x + 1]);
            const f16x8g bhi = as_f16x8g(w[0][ct][j]), blo = as_f16x8g(w[1][ct][j]);
            if (t == 0) accl[2 * rp + x][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bhi, accl[2 * rp + x][ct], 0, 0, 0);
            if (t == 1) accl[2 * rp + x][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, blo, accl[2 * rp + x][ct], 0, 0, 0);
            if (t == 2) acc[2 * rp + x][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bhi, acc[2 * rp + x][ct], 0, 0, 0);
          }
    }
    // per segment of NS rows: (max, lowest arg-max row); element i of acc[rt][ct] = row 16 rt + 4 g4 + i, column 16 ct + l16
    constexpr int RT_PER_SEG = NS / 16, NSEG = GM_TM / NS;
    const long long row_tile = n0 + (long long)tile * GM_TM;
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
      const long long grp = (row_tile + sg * NS) / NS;
      const bool live = row_tile + sg * NS < n1;  // wave-uniform (rows is a multiple of NS)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        float tv = -__builtin_inff();
        int tr = 0;
#pragma unroll
        for (int q = 0; q < RT_PER_SEG; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int rt = sg * RT_PER_SEG + q;
            const float v = fmaf(accl[rt][ct][i], 1.f / GM_SCALE, acc[rt][ct][i]);
            const bool gt = v > tv;  // ascending rows within a lane: strict > keeps the first
            tv = gt ? v : tv;
            tr = gt ? 16 * q + i : tr;
          }
        tr += 4 * g4;
#pragma unroll
        for (int m = 16; m <= 32; m <<= 1) {
          const float ov = __shfl_xor(tv, m, HITADV_WAVE);
          const int orow = __shfl_xor(tr, m, HITADV_WAVE);
          if (ov > tv || (ov == tv && orow < tr)) { tv = ov; tr = orow; }
        }
        if (live && g4 == 0) {
          const int c = col0 + 16 * ct + l16;
          const float v = tv + bv[ct];  // rounding is monotonic: max_j(y_j + b) == max_j(y_j) + b
          out[(size_t)grp * COUT + c] = v > 0.f ? v : 0.f;  // max and ReLU commute
          arg[(size_t)grp * COUT + c] = tr;
        }
      }
    }
  };
  auto step = [&](int tile, float4 (&have)[ST][2], float4 (&next)[ST][2]) {
    if (tile + 2 < nfull) {  // tiles t, t+1, t+2 exist and are whole
      fetch(next, tile + 2, Full{});
      if (late) stash(have, tile + 1, Full{});
      compute(tile);
      if (!late) stash(have, tile + 1, Full{});
      __syncthreads();
      return;
    }
    const bool more = tile + 1 < ntiles;
    if (tile + 2 < ntiles) fetch(next, tile + 2, Ragged{});
    if (more && late) stash(have, tile + 1, Ragged{});
    compute(tile);
    if (more && !late) stash(have, tile + 1, Ragged{});
    __syncthreads();
  };
  fetch(stA, 0, Ragged{});
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): W and tile 0 are complete before the loop (see victim_bf3.hip)
  stash(stA, 0, Ragged{});
  if (ntiles > 1) fetch(stA, 1, Ragged{});
  __syncthreads();
  for (int tile = 0; tile < ntiles; tile += 2) {
    step(tile, stA, stB);
    if (tile + 1 < ntiles) step(tile + 1, stB, stA);
  }
  if (big.beyond_fp16() && range_flag != nullptr) *range_flag = 1;  // a hi piece was an infinity or a NaN: the caller refuses the result
}

// ------------------------------------------------------------------------------------------------ backward
// block = 4 waves = 256 threads; thread c (and c + 256 ...) owns channel c of the tile's groups; wave w owns the output
// columns [16 NCB w, 16 NCB (w + 1)): CIN = 64 NCB.  Wb2 = the pieces of Wt[CIN, COUT] in V1's fragment order with the roles
// of the two dimensions swapped (contraction over the channels): hitadv_split_weights_f16x2(Wt, CIN, COUT, ...).
// MASKED: dX is also gated by (xmask > 0), xmask [G*NS, CIN] = the layer's INPUT when that is itself a ReLU output (the shared
// layer in front): the ReLU backward pass of that layer -- a read of dX, a read of the activation and a write as large as
// both -- happens on the way out of this kernel instead.
template <int CIN, int NS, int COUT, bool MASKED>
__global__ __launch_bounds__(256) void group_linear_max_bwd_k(const float *__restrict__ dOut, const float *__restrict__ outv,
                                                              const int32_t *__restrict__ arg, const uint16_t *__restrict__ Wb2,
                                                              long long groups, int tiles_per_block, float *__restrict__ dX,
                                                              int *range_flag, const float *__restrict__ xmask) {
  constexpr int NCB = CIN / 64;           // 16-column tiles per wave
  constexpr int KSL = COUT / 32;          // 32-deep slices of the contraction over the channels
  constexpr int RS = 2 * COUT + 32;       // bytes per LDS row of one piece of the A operand
  constexpr int PIECE = GM_TM * RS;
  constexpr int NSEG = GM_TM / NS;
  constexpr int CPT = COUT / 256 > 0 ? COUT / 256 : 1;  // channels per thread
  constexpr int LDO = CIN + 4;            // floats per row of the output staging tile
  extern __shared__ __attribute__((aligned(16))) char sB[];  // 2 pieces x PIECE, then the [64, LDO] output tile
  float *sOut = reinterpret_cast<float *>(sB + 2 * PIECE);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l16 = lane & 15, g4 = lane >> 4;
  const long long gpt = NSEG;  // groups per tile
  const long long ntiles_all = (groups + gpt - 1) / gpt;
  const long long t0 = (long long)blockIdx.x * tiles_per_block;
  const int ntiles = (int)max(0ll, min((long long)tiles_per_block, ntiles_all - t0));
  if (ntiles <= 0) return;
  uint4 w[2][NCB][KSL];
  {
    const int col0 = wave * 16 * NCB;
    const uint4 *wp = reinterpret_cast<const uint4 *>(Wb2) + (size_t)(col0 / 16) * KSL * 64 + lane;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int ct = 0; ct < NCB; ++ct)
#pragma unroll
        for (int j = 0; j < KSL; ++j) w[p][ct][j] = wp[((size_t)p * (CIN / 16) * KSL + ct * KSL + j) * 64];
  }
  // the A operand starts as zeros and is returned to zeros after every tile (each thread clears the cells it set)
  for (int e = threadIdx.x; e < 2 * PIECE / 16; e += 256) reinterpret_cast<uint4 *>(sB)[e] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  constexpr int NM = GM_TM * CIN / 4 / 256;  // float4 cells of the output tile per thread
  for (int tile = 0; tile < ntiles; ++tile) {
    const long long grp0 = (t0 + tile) * gpt;
    float4 mk[MASKED ? NM : 1];
    if constexpr (MASKED) {  // requested first: the loads land while the tile's products run
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        const int e = threadIdx.x + 256 * i, r = e / (CIN / 4), c4 = e % (CIN / 4);
        const long long row = min(grp0 * NS + r, groups * NS - 1);
        mk[i] = *reinterpret_cast<const float4 *>(xmask + (size_t)row * CIN + 4 * c4);
      }
    }
    int cell[NSEG][CPT];
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg)
#pragma unroll
      for (int q = 0; q < CPT; ++q) {
        const int c = threadIdx.x + 256 * q;
        const long long grp = grp0 + sg;
        const bool in = c < COUT && grp < groups;
        const size_t o = (size_t)min(grp, groups - 1) * COUT + min(c, COUT - 1);
        const float g = dOut[o], ov = outv[o];
        const int j = arg[o];
        const bool on = in && ov > 0.f && g != 0.f;
        cell[sg][q] = on ? (sg * NS + j) * RS + 2 * c : -1;
        if (on) {
          const _Float16 h1 = (_Float16)g;
          const _Float16 h2 = (_Float16)((g - (float)h1) * GM_SCALE);
          if (!(fabsf(g) < 65504.f) && range_flag != nullptr) *range_flag = 1;
          *reinterpret_cast<_Float16 *>(sB + cell[sg][q]) = h1;
          *reinterpret_cast<_Float16 *>(sB + PIECE + cell[sg][q]) = h2;
        }
      }
    __syncthreads();
    f32x4g acc[4][NCB], accl[4][NCB];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int ct = 0; ct < NCB; ++ct) {
        acc[rt][ct] = f32x4g{0.f, 0.f, 0.f, 0.f};
        accl[rt][ct] = f32x4g{0.f, 0.f, 0.f, 0.f};
      }
    const char *base = sB + l16 * RS + 16 * g4;
#pragma unroll
    for (int j = 0; j < KSL; ++j)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const f16x8g ahi = as_f16x8g(*reinterpret_cast<const uint4 *>(base + rt * 16 * RS + 64 * j));
        const f16x8g alo = as_f16x8g(*reinterpret_cast<const uint4 *>(base + PIECE + rt * 16 * RS + 64 * j));
#pragma unroll
        for (int ct = 0; ct < NCB; ++ct) {
          const f16x8g bhi = as_f16x8g(w[0][ct][j]), blo = as_f16x8g(w[1][ct][j]);
          accl[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bhi, accl[rt][ct], 0, 0, 0);
          accl[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, blo, accl[rt][ct], 0, 0, 0);
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bhi, acc[rt][ct], 0, 0, 0);
        }
      }
    // accumulators -> the staging tile (row = 16 rt + 4 g4 + i, column = 16 NCB wave + 16 ct + l16)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int ct = 0; ct < NCB; ++ct)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          sOut[(16 * rt + 4 * g4 + i) * LDO + wave * 16 * NCB + 16 * ct + l16] =
              fmaf(accl[rt][ct][i], 1.f / GM_SCALE, acc[rt][ct][i]);
    __syncthreads();
    // every thread clears the cells it set; coalesced float4 rows of dX
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg)
#pragma unroll
      for (int q = 0; q < CPT; ++q)
        if (cell[sg][q] >= 0) {
          *reinterpret_cast<_Float16 *>(sB + cell[sg][q]) = (_Float16)0.f;
          *reinterpret_cast<_Float16 *>(sB + PIECE + cell[sg][q]) = (_Float16)0.f;
        }
    const long long row0 = grp0 * NS, rows_all = groups * NS;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      const int e = threadIdx.x + 256 * i, r = e / (CIN / 4), c4 = e % (CIN / 4);
      float4 v = *reinterpret_cast<const float4 *>(sOut + r * LDO + 4 * c4);
      if constexpr (MASKED) {
        v.x = mk[i].x > 0.f ? v.x : 0.f;
        v.y = mk[i].y > 0.f ? v.y : 0.f;
        v.z = mk[i].z > 0.f ? v.z : 0.f;
        v.w = mk[i].w > 0.f ? v.w : 0.f;
      }
      if (row0 + r < rows_all) *reinterpret_cast<float4 *>(dX + (size_t)(row0 + r) * CIN + 4 * c4) = v;
    }
    __syncthreads();
  }
}

template <int CIN, int NS, int NCT>
static int launch_fwd(const float *X, const uint16_t *W2, const float *bias, long long G, float *out, int32_t *arg,
                      int32_t *range_flag, hipStream_t s) {
  const long long rows = G * NS, ntiles = (rows + GM_TM - 1) / GM_TM;
  // about two blocks per CU, at least eight tiles per block (every block first loads its slice of W)
  long long blocks = min(ntiles, 512ll);
  int tpb = (int)((ntiles + blocks - 1) / blocks);
  if (tpb < 8) tpb = (int)min(8ll, ntiles);
  blocks = (ntiles + tpb - 1) / tpb;
  const size_t shm = (size_t)2 * 2 * GM_TM * (2 * CIN + 32);
  HITADV_RAISE_LDS((&group_linear_max_fwd_k<CIN, NS, NCT>), (int)(2 * 2 * GM_TM * (2 * CIN + 32)));
  group_linear_max_fwd_k<CIN, NS, NCT><<<(unsigned)blocks, 512, shm, s>>>(X, W2, bias, rows, tpb, out, arg, range_flag);
  return 0;
}

template <int CIN, int NS, int COUT, bool MASKED>
static int launch_bwd(const float *dOut, const float *outv, const int32_t *arg, const uint16_t *Wb2, long long G, float *dX,
                      int32_t *range_flag, const float *xmask, hipStream_t s) {
  const long long ntiles = (G * NS + GM_TM - 1) / GM_TM;
  long long blocks = min(ntiles, 1024ll);
  int tpb = (int)((ntiles + blocks - 1) / blocks);
  if (tpb < 4) tpb = (int)min(4ll, ntiles);
  blocks = (ntiles + tpb - 1) / tpb;
  const size_t shm = (size_t)2 * GM_TM * (2 * COUT + 32) + (size_t)GM_TM * (CIN + 4) * sizeof(float);
  HITADV_RAISE_LDS((&group_linear_max_bwd_k<CIN, NS, COUT, MASKED>), (int)(2 * GM_TM * (2 * COUT + 32) + GM_TM * (CIN + 4) * sizeof(float)));
  group_linear_max_bwd_k<CIN, NS, COUT, MASKED><<<(unsigned)blocks, 256, shm, s>>>(dOut, outv, arg, Wb2, G, tpb, dX, range_flag, xmask);
  return 0;
}

}  // namespace hitadv

using namespace hitadv;

extern "C" int hitadv_group_linear_max_supported(int Cin, int Cout, int ns) {
  return (ns == 32 || ns == 64) && ((Cin == 64 && Cout == 128) || (Cin == 128 && Cout == 128) || (Cin == 128 && Cout == 256));
}

extern "C" int hitadv_group_linear_max_fwd(const float *X, const uint16_t *W2, const float *bias, int64_t G, int ns, int Cin,
                                           int Cout, float *out, int32_t *arg, int32_t *range_flag, void *stream) {
  if (!X || !W2 || !out || !arg || G <= 0 || !hitadv_group_linear_max_supported(Cin, Cout, ns) || ((uintptr_t)X & 15) ||
      ((uintptr_t)W2 & 15))
    return HITADV_E_ARG;
  hipStream_t s = (hipStream_t)stream;
#define HITADV_GLM_FWD(CI, CO)                                                                                          \
  if (Cin == CI && Cout == CO) {                                                                                        \
    if (ns == 32) launch_fwd<CI, 32, CO / 128>(X, W2, bias, G, out, arg, range_flag, s);                                \
    else launch_fwd<CI, 64, CO / 128>(X, W2, bias, G, out, arg, range_flag, s);                                         \
  }
  HITADV_GLM_FWD(64, 128)
  HITADV_GLM_FWD(128, 128)
  HITADV_GLM_FWD(128, 256)
#undef HITADV_GLM_FWD
  HITADV_LAUNCH_CHECK();
  return 0;
}

static int glm_bwd(const float *dOut, const float *out, const int32_t *arg, const uint16_t *Wb2, int64_t G, int ns, int Cin,
                   int Cout, const float *xmask, float *dX, int32_t *range_flag, void *stream) {
  if (!dOut || !out || !arg || !Wb2 || !dX || G <= 0 || !hitadv_group_linear_max_supported(Cin, Cout, ns) ||
      ((uintptr_t)dX & 15) || ((uintptr_t)Wb2 & 15) || ((uintptr_t)xmask & 15))
    return HITADV_E_ARG;
  hipStream_t s = (hipStream_t)stream;
#define HITADV_GLM_BWD(CI, CO)                                                                                          \
  if (Cin == CI && Cout == CO) {                                                                                        \
    if (xmask != nullptr) {                                                                                             \
      if (ns == 32) launch_bwd<CI, 32, CO, true>(dOut, out, arg, Wb2, G, dX, range_flag, xmask, s);                     \
      else launch_bwd<CI, 64, CO, true>(dOut, out, arg, Wb2, G, dX, range_flag, xmask, s);                              \
    } else {                                                                                                            \
      if (ns == 32) launch_bwd<CI, 32, CO, false>(dOut, out, arg, Wb2, G, dX, range_flag, nullptr, s);                  \
      else launch_bwd<CI, 64, CO, false>(dOut, out, arg, Wb2, G, dX, range_flag, nullptr, s);                           \
    }                                                                                                                   \
  }
  HITADV_GLM_BWD(64, 128)
  HITADV_GLM_BWD(128, 128)
  HITADV_GLM_BWD(128, 256)
#undef HITADV_GLM_BWD
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_group_linear_max_bwd(const float *dOut, const float *out, const int32_t *arg, const uint16_t *Wb2,
                                           int64_t G, int ns, int Cin, int Cout, float *dX, int32_t *range_flag, void *stream) {
  return glm_bwd(dOut, out, arg, Wb2, G, ns, Cin, Cout, nullptr, dX, range_flag, stream);
}

extern "C" int hitadv_group_linear_max_bwd_masked(const float *dOut, const float *out, const int32_t *arg, const uint16_t *Wb2,
                                                  int64_t G, int ns, int Cin, int Cout, const float *xmask, float *dX,
                                                  int32_t *range_flag, void *stream) {
  if (!xmask) return HITADV_E_ARG;
  return glm_bwd(dOut, out, arg, Wb2, G, ns, Cin, Cout, xmask, dX, range_flag, stream);
}
