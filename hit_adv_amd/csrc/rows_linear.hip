// Y[rows, COUT] = act(X[rows, CIN] W^T + bias) for MANY rows and a small square-ish layer (CIN, COUT in {64, 128}): the middle
// shared layers of PointNet++'s set-abstraction blocks (model/pointnet2_utils.py:197-201, conv -> bn -> relu on
// [B, C, nsample, npoint] = 0.5-1 M rows of 64 / 128 channels at cfg4) and their input gradients (dX = dY W: the same kernel on
// the transposed weights, no bias, no activation).  These products are bound by the read of X and the write of Y (537 MB at
// 1 M x 64 -> 64); the library's f32 GEMM moves them at 2.8 TB/s (189 / 230 us forward at cfg4's two levels).
//
// V1's scheme (csrc/victim_bf3.hip, FLAT form) on a flat stream of 64-row tiles: x split into two fp16 pieces on the way into
// LDS (hi pieces two per v_cvt_pk_f16_f32, lo pieces one v_fma_mixlo/hi each), W's pieces in registers for the whole kernel,
// three exact fp16 MFMAs per useful product into two fp32 accumulator sets (fp32-accurate), tiles double buffered with the loads
// two tiles ahead.  The result leaves through a double-buffered LDS tile so that every store instruction writes whole rows.
//   block = 8 waves: WC = COUT / 16 column tiles x WR = 8 / WC row groups; a wave owns one 16-column tile of 64 / WR rows.
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"
#include "hitadv.h"

namespace hitadv {

typedef float f32x4r __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8r __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f16x8r as_f16x8r(uint4 u) { return __builtin_bit_cast(f16x8r, u); }
constexpr int RL_TM = 64;           // rows per tile
constexpr float RL_SCALE = 2048.f;  // 2^11: the second piece is the residual scaled back into fp16's normal range

template <int CIN, int COUT>
__global__ __launch_bounds__(512) void rows_linear_k(const float *__restrict__ X, const uint16_t *__restrict__ W2,
                                                     const float *__restrict__ bias, long long rows, int tiles_per_block, int relu,
                                                     float *__restrict__ Y, int *range_flag) {
  constexpr int NSL = CIN / 32;
  constexpr int RS = 2 * CIN + 32;  // bytes per LDS row of one piece (conflict-free 16x16x32 A-fragment reads)
  constexpr int PIECE = RL_TM * RS;
  constexpr int G8 = CIN / 8;
  constexpr int ST = RL_TM * G8 / 512;  // 8-value groups staged per thread per tile
  constexpr int WC = COUT / 16, WR = 8 / WC, RT = 4 / WR;
  constexpr int LDO = COUT + 4;          // floats per row of the output tile
  constexpr int OUTB = RL_TM * LDO * 4;  // bytes of one output tile
  constexpr int NO = RL_TM * COUT / 4 / 512;  // float4 of the output tile per thread
  static_assert(ST >= 1 && WR >= 1 && RT >= 1 && NO >= 1, "shape");
  extern __shared__ __attribute__((aligned(16))) char sR[];  // 2 x 2 pieces x PIECE, then 2 output tiles
  char *const sOut = sR + 4 * PIECE;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l16 = lane & 15, g4 = lane >> 4;
  const int wc = wave % WC, wr = wave / WC;
  const long long ntiles_all = (rows + RL_TM - 1) / RL_TM;
  const long long t0 = (long long)blockIdx.x * tiles_per_block;
  const int ntiles = (int)max(0ll, min((long long)tiles_per_block, ntiles_all - t0));
  if (ntiles <= 0) return;
  const long long n0 = t0 * RL_TM, n1 = min(rows, n0 + (long long)ntiles * RL_TM);
  const int nfull = (int)((n1 - n0) / RL_TM);  // tiles wholly inside the matrix (all, or all but the last)

  // W2 is in V1's fragment order [piece][c / 16][k / 32][lane] x 16 bytes (hitadv_split_weights_f16x2 of W [COUT, CIN])
  uint4 w[2][NSL];
  {
    const uint4 *wp = reinterpret_cast<const uint4 *>(W2) + (size_t)wc * NSL * 64 + lane;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int j = 0; j < NSL; ++j) w[p][j] = wp[((size_t)p * (COUT / 16) * NSL + j) * 64];
  }
  const float bv = bias != nullptr ? bias[16 * wc + l16] : 0.f;

  // Whole tiles and the (possibly ragged) last one are separate instantiations: a run-time bounds test becomes a select per
  // VALUE (csrc/victim_bf3.hip), and vector instructions are not hidden behind anything here either.
  using Full = std::true_type;
  using Ragged = std::false_type;
  uint32_t soff[ST];
#pragma unroll
  for (int u = 0; u < ST; ++u) {
    const int e = threadIdx.x + 512 * u;
    soff[u] = (uint32_t)((e / G8) * CIN + 8 * (e % G8)) * 4u;
  }
  float4 stA[ST][2], stB[ST][2];
  auto fetch = [&](float4 (&st)[ST][2], int tile, auto full_c) {
    if constexpr (decltype(full_c)::value) {
      const char *tb = reinterpret_cast<const char *>(X) + (size_t)(n0 + (long long)tile * RL_TM) * CIN * 4;
#pragma unroll
      for (int u = 0; u < ST; ++u) {
        st[u][0] = *reinterpret_cast<const float4 *>(tb + soff[u]);
        st[u][1] = *reinterpret_cast<const float4 *>(tb + soff[u] + 16);
      }
    } else {
#pragma unroll
      for (int u = 0; u < ST; ++u) {
        const int e = threadIdx.x + 512 * u;
        const long long n = n0 + (long long)tile * RL_TM + e / G8;
        const float *sp = n < n1 ? X + (size_t)n * CIN + 8 * (e % G8) : X;
        st[u][0] = *reinterpret_cast<const float4 *>(sp);
        st[u][1] = *reinterpret_cast<const float4 *>(sp + 4);
      }
    }
  };
  PieceWatch big;
  auto stash = [&](const float4 (&st)[ST][2], int tile, auto full_c) {
    const int buf = tile & 1;
    const float nsc = -RL_SCALE;
#pragma unroll
    for (int u = 0; u < ST; ++u) {
      const int e = threadIdx.x + 512 * u;
      float a[8] = {st[u][0].x, st[u][0].y, st[u][0].z, st[u][0].w, st[u][1].x, st[u][1].y, st[u][1].z, st[u][1].w};
      if constexpr (!decltype(full_c)::value) {
        const bool in = n0 + (long long)tile * RL_TM + e / G8 < n1;  // rows past the end are zero in LDS (and not written out)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = in ? a[i] : 0.f;
      }
      uint32_t H[4], L[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float s0 = a[2 * p] * RL_SCALE, s1 = a[2 * p + 1] * RL_SCALE;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(H[p]) : "v"(a[2 * p]), "v"(a[2 * p + 1]));
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(L[p]) : "v"(H[p]), "s"(nsc), "v"(s0));
        asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(L[p]) : "v"(H[p]), "s"(nsc), "v"(s1));
        big.see_f16x2(H[p]);
      }
      char *dst = sR + (size_t)buf * 2 * PIECE + (e / G8) * RS + 16 * (e % G8);
      *reinterpret_cast<uint4 *>(dst) = make_uint4(H[0], H[1], H[2], H[3]);
      *reinterpret_cast<uint4 *>(dst + PIECE) = make_uint4(L[0], L[1], L[2], L[3]);
    }
  };
  const bool late = wave >= 4;
  // element i of acc[rt] = row 16 (RT wr + rt) + 4 g4 + i, column 16 wc + l16 of the tile
  auto compute = [&](int tile) {
    const char *base = sR + (size_t)(tile & 1) * 2 * PIECE + (16 * RT * wr + l16) * RS + 16 * g4;
    f32x4r acc[RT], accl[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      acc[rt] = f32x4r{0.f, 0.f, 0.f, 0.f};
      accl[rt] = f32x4r{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < NSL; ++j) {
      const f16x8r bhi = as_f16x8r(w[0][j]), blo = as_f16x8r(w[1][j]);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const f16x8r ahi = as_f16x8r(*reinterpret_cast<const uint4 *>(base + rt * 16 * RS + 64 * j));
        const f16x8r alo = as_f16x8r(*reinterpret_cast<const uint4 *>(base + PIECE + rt * 16 * RS + 64 * j));
        accl[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bhi, accl[rt], 0, 0, 0);
        accl[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, blo, accl[rt], 0, 0, 0);
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bhi, acc[rt], 0, 0, 0);
      }
    }
    float *so = reinterpret_cast<float *>(sOut + (size_t)(tile & 1) * OUTB) + (16 * RT * wr + 4 * g4) * LDO + 16 * wc + l16;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = fmaf(accl[rt][i], 1.f / RL_SCALE, acc[rt][i]) + bv;  // the two powers of two meet
        v = relu ? fmaxf(v, 0.f) : v;
        so[(16 * rt + i) * LDO] = v;
      }
  };
  // tile t leaves after the barrier that ends its step: whole rows per store instruction.  Two output tiles: the waves that are
  // ahead write tile t+1's results while the others still copy tile t out (one barrier per tile)
  auto leave = [&](int tile, auto full_c) {
    const float *so = reinterpret_cast<const float *>(sOut + (size_t)(tile & 1) * OUTB);
    const long long row0 = n0 + (long long)tile * RL_TM;
#pragma unroll
    for (int u = 0; u < NO; ++u) {
      const int e = threadIdx.x + 512 * u, r = e / (COUT / 4), c4 = e % (COUT / 4);
      const float4 v = *reinterpret_cast<const float4 *>(so + r * LDO + 4 * c4);
      if (decltype(full_c)::value || row0 + r < n1) *reinterpret_cast<float4 *>(Y + (size_t)(row0 + r) * COUT + 4 * c4) = v;
    }
  };
  auto steady = [&](int tile, float4 (&have)[ST][2], float4 (&next)[ST][2]) {  // tiles t, t+1, t+2 exist and are whole
    fetch(next, tile + 2, Full{});
    if (late) stash(have, tile + 1, Full{});
    compute(tile);
    if (!late) stash(have, tile + 1, Full{});
    __syncthreads();
    leave(tile, Full{});
  };
  auto step = [&](int tile, float4 (&have)[ST][2], float4 (&next)[ST][2]) {
    if (tile + 2 < nfull) return steady(tile, have, next);
    const bool more = tile + 1 < ntiles;
    if (tile + 2 < ntiles) fetch(next, tile + 2, Ragged{});
    if (more && late) stash(have, tile + 1, Ragged{});
    compute(tile);
    if (more && !late) stash(have, tile + 1, Ragged{});
    __syncthreads();
    leave(tile, Ragged{});
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

// The same layer with its INPUT produced on the way in: x[(b, i, s), :] = relu(U[b, idx[b,i,s], :] + V[b, i, :]) -- the first shared
// layer of the block after its split over the neighbour and the centre (csrc/grouping.hip::group_add_relu_fwd_k) -- so that the
// [B, S, ns, CIN] activation between the two layers (268 MB at cfg4's levels, written once and read once) never exists.  The
// backward pass does not need it either (group_add_relu_bwd recomputes its ReLU mask from U and V).  Same arithmetic as
// group_add_relu followed by rows_linear: the same bits.  Every tile is whole (rows a multiple of 64, a cloud a whole number of
// tiles: the launcher checks): the neighbour numbers of tile t+3 and the rows of tile t+2 are requested while tile t is multiplied.
template <int CIN, int COUT>
__global__ __launch_bounds__(512) void rows_linear_gather_k(const float *__restrict__ U, const float *__restrict__ V,
                                                            const int64_t *__restrict__ idx, int N, int S, int ns_shift,
                                                            const uint16_t *__restrict__ W2, const float *__restrict__ bias,
                                                            long long rows, int tiles_per_block, int relu, float *__restrict__ Y,
                                                            int *range_flag) {
  constexpr int NSL = CIN / 32;
  constexpr int RS = 2 * CIN + 32;
  constexpr int PIECE = RL_TM * RS;
  constexpr int G8 = CIN / 8;
  constexpr int ST = RL_TM * G8 / 512;
  constexpr int WC = COUT / 16, WR = 8 / WC, RT = 4 / WR;
  constexpr int LDO = COUT + 4;
  constexpr int OUTB = RL_TM * LDO * 4;
  constexpr int NO = RL_TM * COUT / 4 / 512;
  extern __shared__ __attribute__((aligned(16))) char sR[];
  char *const sOut = sR + 4 * PIECE;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l16 = lane & 15, g4 = lane >> 4;
  const int wc = wave % WC, wr = wave / WC;
  const long long ntiles_all = rows / RL_TM;
  const long long t0 = (long long)blockIdx.x * tiles_per_block;
  const int ntiles = (int)max(0ll, min((long long)tiles_per_block, ntiles_all - t0));
  if (ntiles <= 0) return;
  const long long n0 = t0 * RL_TM;
  const int last = ntiles - 1;
  const long long cloud_rows = (long long)S << ns_shift;

  uint4 w[2][NSL];
  {
    const uint4 *wp = reinterpret_cast<const uint4 *>(W2) + (size_t)wc * NSL * 64 + lane;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int j = 0; j < NSL; ++j) w[p][j] = wp[((size_t)p * (COUT / 16) * NSL + j) * 64];
  }
  const float bv = bias != nullptr ? bias[16 * wc + l16] : 0.f;

  struct Set {
    float4 u[ST][2], v[ST][2];
    bool ok[ST];
  };
  // the neighbour numbers of a tile's rows (one per staged group of this thread)
  auto load_idx = [&](long long (&j)[ST], int tile) {
#pragma unroll
    for (int q = 0; q < ST; ++q) j[q] = idx[n0 + (long long)tile * RL_TM + (threadIdx.x + 512 * q) / G8];
  };
  auto fetch = [&](Set &st, int tile, const long long (&j)[ST]) {
    const long long r0 = n0 + (long long)tile * RL_TM;
    const long long b = r0 / cloud_rows;  // wave-uniform: a cloud is a whole number of tiles
#pragma unroll
    for (int q = 0; q < ST; ++q) {
      const int e = threadIdx.x + 512 * q;
      const long long bi = (r0 + e / G8) >> ns_shift;
      st.ok[q] = j[q] >= 0 && j[q] < N;
      const float *up = U + ((size_t)(b * N + (st.ok[q] ? j[q] : 0))) * CIN + 8 * (e % G8);
      const float *vp = V + (size_t)bi * CIN + 8 * (e % G8);
      st.u[q][0] = *reinterpret_cast<const float4 *>(up);
      st.u[q][1] = *reinterpret_cast<const float4 *>(up + 4);
      st.v[q][0] = *reinterpret_cast<const float4 *>(vp);
      st.v[q][1] = *reinterpret_cast<const float4 *>(vp + 4);
    }
  };
  PieceWatch big;
  auto stash = [&](const Set &st, int tile) {
    const int buf = tile & 1;
    const float nsc = -RL_SCALE;
#pragma unroll
    for (int q = 0; q < ST; ++q) {
      const int e = threadIdx.x + 512 * q;
      const float uu[8] = {st.u[q][0].x, st.u[q][0].y, st.u[q][0].z, st.u[q][0].w, st.u[q][1].x, st.u[q][1].y, st.u[q][1].z, st.u[q][1].w};
      const float vv[8] = {st.v[q][0].x, st.v[q][0].y, st.v[q][0].z, st.v[q][0].w, st.v[q][1].x, st.v[q][1].y, st.v[q][1].z, st.v[q][1].w};
      float a[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = st.ok[q] ? fmaxf(uu[i] + vv[i], 0.f) : 0.f;  // group_add_relu_fwd_k's value
      uint32_t H[4], L[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float s0 = a[2 * p] * RL_SCALE, s1 = a[2 * p + 1] * RL_SCALE;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(H[p]) : "v"(a[2 * p]), "v"(a[2 * p + 1]));
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(L[p]) : "v"(H[p]), "s"(nsc), "v"(s0));
        asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(L[p]) : "v"(H[p]), "s"(nsc), "v"(s1));
        big.see_f16x2(H[p]);
      }
      char *dst = sR + (size_t)buf * 2 * PIECE + (e / G8) * RS + 16 * (e % G8);
      *reinterpret_cast<uint4 *>(dst) = make_uint4(H[0], H[1], H[2], H[3]);
      *reinterpret_cast<uint4 *>(dst + PIECE) = make_uint4(L[0], L[1], L[2], L[3]);
    }
  };
  const bool late = wave >= 4;
  auto compute = [&](int tile) {
    const char *base = sR + (size_t)(tile & 1) * 2 * PIECE + (16 * RT * wr + l16) * RS + 16 * g4;
    f32x4r acc[RT], accl[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      acc[rt] = f32x4r{0.f, 0.f, 0.f, 0.f};
      accl[rt] = f32x4r{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < NSL; ++j) {
      const f16x8r bhi = as_f16x8r(w[0][j]), blo = as_f16x8r(w[1][j]);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const f16x8r ahi = as_f16x8r(*reinterpret_cast<const uint4 *>(base + rt * 16 * RS + 64 * j));
        const f16x8r alo = as_f16x8r(*reinterpret_cast<const uint4 *>(base + PIECE + rt * 16 * RS + 64 * j));
        accl[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bhi, accl[rt], 0, 0, 0);
        accl[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, blo, accl[rt], 0, 0, 0);
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bhi, acc[rt], 0, 0, 0);
      }
    }
    float *so = reinterpret_cast<float *>(sOut + (size_t)(tile & 1) * OUTB) + (16 * RT * wr + 4 * g4) * LDO + 16 * wc + l16;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = fmaf(accl[rt][i], 1.f / RL_SCALE, acc[rt][i]) + bv;
        v = relu ? fmaxf(v, 0.f) : v;
        so[(16 * rt + i) * LDO] = v;
      }
  };
  auto leave = [&](int tile) {
    const float *so = reinterpret_cast<const float *>(sOut + (size_t)(tile & 1) * OUTB);
    const long long row0 = n0 + (long long)tile * RL_TM;
#pragma unroll
    for (int q = 0; q < NO; ++q) {
      const int e = threadIdx.x + 512 * q, r = e / (COUT / 4), c4 = e % (COUT / 4);
      *reinterpret_cast<float4 *>(Y + (size_t)(row0 + r) * COUT + 4 * c4) = *reinterpret_cast<const float4 *>(so + r * LDO + 4 * c4);
    }
  };
  // every step is the steady one: past the end of the block's tiles the last tile is requested again (and written to the LDS
  // buffer nobody reads any more)
  auto step = [&](int tile, Set &have, Set &next, const long long (&jc)[ST], long long (&jn)[ST]) {
    load_idx(jn, min(tile + 3, last));
    fetch(next, min(tile + 2, last), jc);
    if (late) stash(have, tile + 1);
    compute(tile);
    if (!late) stash(have, tile + 1);
    __syncthreads();
    leave(tile);
  };
  Set sa, sb;
  long long ja[ST], jb[ST];
  load_idx(ja, 0);
  fetch(sa, 0, ja);
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): W and tile 0 are complete before the loop (see victim_bf3.hip)
  stash(sa, 0);
  load_idx(ja, min(1, last));
  fetch(sa, min(1, last), ja);
  load_idx(jb, min(2, last));
  __syncthreads();
  for (int tile = 0; tile < ntiles; tile += 2) {
    step(tile, sa, sb, jb, ja);
    if (tile + 1 < ntiles) step(tile + 1, sb, sa, ja, jb);
  }
  if (big.beyond_fp16() && range_flag != nullptr) *range_flag = 1;
}

template <int CIN, int COUT>
static int launch_rows_linear(const float *X, const uint16_t *W2, const float *bias, long long rows, int relu, float *Y,
                              int32_t *range_flag, hipStream_t s) {
  const long long ntiles = (rows + RL_TM - 1) / RL_TM;
  // one block per CU (141 KB of LDS at 128 -> 128), at least eight tiles per block (every block first loads its slice of W)
  long long blocks = min(ntiles, 256ll);
  int tpb = (int)((ntiles + blocks - 1) / blocks);
  if (tpb < 8) tpb = (int)min(8ll, ntiles);
  blocks = (ntiles + tpb - 1) / tpb;
  constexpr int shm = 4 * RL_TM * (2 * CIN + 32) + 2 * RL_TM * (COUT + 4) * 4;
  HITADV_RAISE_LDS((&rows_linear_k<CIN, COUT>), shm);
  rows_linear_k<CIN, COUT><<<(unsigned)blocks, 512, shm, s>>>(X, W2, bias, rows, tpb, relu, Y, range_flag);
  HITADV_LAUNCH_CHECK();
  return 0;
}

}  // namespace hitadv

using namespace hitadv;

extern "C" int hitadv_rows_linear_supported(int Cin, int Cout) { return ((Cin == 64 || Cin == 128) && (Cout == 64 || Cout == 128)) ? 1 : 0; }

template <int CIN, int COUT>
static int launch_rows_linear_gather(const float *U, const float *V, const int64_t *idx, int N, int S, int ns_shift, const uint16_t *W2,
                                     const float *bias, long long rows, int relu, float *Y, int32_t *range_flag, hipStream_t s) {
  const long long ntiles = rows / RL_TM;
  constexpr int shm = 4 * RL_TM * (2 * CIN + 32) + 2 * RL_TM * (COUT + 4) * 4;
  // two blocks per CU where the LDS holds them (76 KB at 64 -> 64): this form waits on gathers, not on HBM: 96.9 -> 90.3 us at cfg4's
  // first level (the plain form, which streams, gains nothing from it)
  long long blocks = min(ntiles, 256ll * (shm <= 80 * 1024 ? 2 : 1));
  int tpb = (int)((ntiles + blocks - 1) / blocks);
  if (tpb < 8) tpb = (int)min(8ll, ntiles);
  blocks = (ntiles + tpb - 1) / tpb;
  HITADV_RAISE_LDS((&rows_linear_gather_k<CIN, COUT>), shm);
  rows_linear_gather_k<CIN, COUT><<<(unsigned)blocks, 512, shm, s>>>(U, V, idx, N, S, ns_shift, W2, bias, rows, tpb, relu, Y, range_flag);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_group_add_relu_linear_supported(int C, int Cout, int S, int ns) {
  return (hitadv_rows_linear_supported(C, Cout) && (ns == 16 || ns == 32 || ns == 64) && S > 0 && ((long long)S * ns) % RL_TM == 0) ? 1 : 0;
}

extern "C" int hitadv_group_add_relu_linear(const float *U, const float *V, const int64_t *idx, int B, int N, int S, int ns, int C,
                                            const uint16_t *W2, const float *bias, int Cout, int relu, float *Y, int32_t *range_flag,
                                            void *stream) {
  if (!U || !V || !idx || !W2 || !Y || B <= 0 || N <= 0 || !hitadv_group_add_relu_linear_supported(C, Cout, S, ns) ||
      (((uintptr_t)U | (uintptr_t)V | (uintptr_t)W2 | (uintptr_t)Y) & 15))
    return HITADV_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long long rows = (long long)B * S * ns;
  const int sh = ns == 16 ? 4 : (ns == 32 ? 5 : 6);
  if (C == 64 && Cout == 64) return launch_rows_linear_gather<64, 64>(U, V, idx, N, S, sh, W2, bias, rows, relu, Y, range_flag, s);
  if (C == 64 && Cout == 128) return launch_rows_linear_gather<64, 128>(U, V, idx, N, S, sh, W2, bias, rows, relu, Y, range_flag, s);
  if (C == 128 && Cout == 64) return launch_rows_linear_gather<128, 64>(U, V, idx, N, S, sh, W2, bias, rows, relu, Y, range_flag, s);
  return launch_rows_linear_gather<128, 128>(U, V, idx, N, S, sh, W2, bias, rows, relu, Y, range_flag, s);
}

extern "C" int hitadv_rows_linear(const float *X, const uint16_t *W2, const float *bias, int64_t rows, int Cin, int Cout, int relu,
                                  float *Y, int32_t *range_flag, void *stream) {
  if (!X || !W2 || !Y || rows <= 0 || !hitadv_rows_linear_supported(Cin, Cout) || ((uintptr_t)X & 15) || ((uintptr_t)W2 & 15) ||
      ((uintptr_t)Y & 15))
    return HITADV_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (Cin == 64 && Cout == 64) return launch_rows_linear<64, 64>(X, W2, bias, rows, relu, Y, range_flag, s);
  if (Cin == 64 && Cout == 128) return launch_rows_linear<64, 128>(X, W2, bias, rows, relu, Y, range_flag, s);
  if (Cin == 128 && Cout == 64) return launch_rows_linear<128, 64>(X, W2, bias, rows, relu, Y, range_flag, s);
  return launch_rows_linear<128, 128>(X, W2, bias, rows, relu, Y, range_flag, s);
}
