// Backward of "shared linear layer -> max over points" (PointNet's 128->1024 layer + global max-pool).
//
// With y[b,n,j] = x[b,n,:].W[j,:] + bias[j] and g[b,j] = max_n y[b,n,j], the gradient w.r.t. x is
//     dX[b,n,:] = sum_{j : argmax[b,j] == n} dg[b,j] * W[j,:]
// i.e. only B*Cout rows of W are ever added, to at most Cout distinct points per cloud.  torch's
// autograd materialises the [B*N,Cout] gradient (zero fill + scatter), and runs a dense GEMM over it;
// this kernel does the sparse sum directly: the arg-max table of the cloud sits in LDS, matches are
// found with ballot and summed in a fixed order (no atomics); see linear_max_bwd_k for the work split.
#include "common.hpp"
#include "hitadv.h"

namespace hitadv {

constexpr int LM_MAXC = 8;   // Cin <= 64 * LM_MAXC
constexpr int LM_ROWS = 8;   // destination points per block (small: hot points spread over more CUs)
constexpr int LM_UN = 8;     // rows of W kept in flight per wave

// One block per 32 destination points of a cloud.  Most points own no channel (a few hundred of the
// 1024 win the max-pool), and a few own dozens, so per point: skip if no match (LDS bitmap), otherwise
// the 4 waves split the channel range, each keeps LM_UN rows of W in flight, and the 4 partial sums are
// added in wave order -> deterministic, and a hot point costs matches/32 dependent L2 round trips.
__global__ __launch_bounds__(256) void linear_max_bwd_k(const float *__restrict__ dg, const float *__restrict__ W,
                                                        const int64_t *__restrict__ idx,
                                                        const float *__restrict__ act_out, int N, int Cout, int Cin,
                                                        float *__restrict__ dX) {
  extern __shared__ int sidx[];  // Cout entries, then 4*Cin floats of partial sums
  __shared__ int has[LM_ROWS];
  float *part = reinterpret_cast<float *>(sidx + Cout);
  const int b = blockIdx.y, n0 = blockIdx.x * LM_ROWS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < LM_ROWS) has[threadIdx.x] = 0;
  __syncthreads();
  for (int j = threadIdx.x; j < Cout; j += 256) {
    const int n = (int)idx[(size_t)b * Cout + j];
    sidx[j] = n;
    if (n >= n0 && n < n0 + LM_ROWS) has[n - n0] = 1;
  }
  __syncthreads();
  const float *dgb = dg + (size_t)b * Cout;
  const float *aob = act_out ? act_out + (size_t)b * Cout : nullptr;  // ReLU'd maxima: gradient passes where > 0
  const int nc = (Cin + 63) >> 6;
  const int jper = (((Cout + 3) >> 2) + 63) & ~63;  // channels per wave, multiple of 64
  const int jlo = wave * jper, jhi = min(Cout, jlo + jper);
  for (int r = 0; r < LM_ROWS; ++r) {
    const int n = n0 + r;
    if (n >= N) break;
    float *o = dX + ((size_t)b * N + n) * Cin;
    if (!has[r]) {  // block-uniform
      for (int k = threadIdx.x; k < Cin; k += 256) o[k] = 0.f;
      continue;
    }
    float acc[LM_MAXC];
#pragma unroll
    for (int c = 0; c < LM_MAXC; ++c) acc[c] = 0.f;
    for (int j0 = jlo; j0 < jhi; j0 += 64) {
      const int j = j0 + lane;
      unsigned long long m = __ballot(j < jhi && sidx[j] == n);
      while (m) {
        int jj[LM_UN];
        float g[LM_UN];
#pragma unroll
        for (int u = 0; u < LM_UN; ++u) {
          const bool on = m != 0ull;
          jj[u] = on ? j0 + __builtin_ctzll(m) : 0;
          m = on ? (m & (m - 1)) : 0ull;
          g[u] = on ? ((aob == nullptr || aob[jj[u]] > 0.f) ? dgb[jj[u]] : 0.f) : 0.f;
        }
#pragma unroll
        for (int c = 0; c < LM_MAXC; ++c) {
          if (c < nc) {
            const int k = lane + 64 * c;
            float w[LM_UN];
#pragma unroll
            for (int u = 0; u < LM_UN; ++u) w[u] = k < Cin ? W[(size_t)jj[u] * Cin + k] : 0.f;
#pragma unroll
            for (int u = 0; u < LM_UN; ++u) acc[c] = fmaf(g[u], w[u], acc[c]);
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < LM_MAXC; ++c)
      if (c < nc) {
        const int k = lane + 64 * c;
        if (k < Cin) part[wave * Cin + k] = acc[c];
      }
    __syncthreads();
    for (int k = threadIdx.x; k < Cin; k += 256)
      o[k] = ((part[k] + part[Cin + k]) + part[2 * Cin + k]) + part[3 * Cin + k];
    __syncthreads();
  }
}

// Max over the point axis of y[B,N,C] (points-major activations), with the arg-max and the lowest
// point index on ties.  HBM-read-bound: float4 per lane along C (4 KiB per block-row), the point range
// split over blockIdx.y so that a B x C = 32k-output reduction still fills 256 CUs; partials are merged
// in split order by max_over_points_merge.
constexpr int MP_SPLIT = 16;
__global__ __launch_bounds__(256) void max_over_points_k(const float *__restrict__ y, int N, int C,
                                                         float *__restrict__ pval, int32_t *__restrict__ pidx) {
  const int b = blockIdx.z, s = blockIdx.y;
  const int c0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c0 >= C) return;
  const int per = (N + MP_SPLIT - 1) / MP_SPLIT;
  const int n0 = s * per, n1 = min(N, n0 + per);
  float4 best = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff());
  int i0 = n0, i1 = n0, i2 = n0, i3 = n0;
  const float *p = y + ((size_t)b * N + n0) * C + c0;
#pragma unroll 4
  for (int n = n0; n < n1; ++n, p += C) {
    const float4 v = *reinterpret_cast<const float4 *>(p);
    if (v.x > best.x) { best.x = v.x; i0 = n; }
    if (v.y > best.y) { best.y = v.y; i1 = n; }
    if (v.z > best.z) { best.z = v.z; i2 = n; }
    if (v.w > best.w) { best.w = v.w; i3 = n; }
  }
  const size_t o = ((size_t)b * MP_SPLIT + s) * C + c0;
  *reinterpret_cast<float4 *>(pval + o) = best;
  *reinterpret_cast<int4 *>(pidx + o) = make_int4(i0, i1, i2, i3);
}

__global__ __launch_bounds__(256) void max_over_points_merge(const float *__restrict__ pval,
                                                             const int32_t *__restrict__ pidx, int C, int S,
                                                             const float *__restrict__ bias, int relu,
                                                             float *__restrict__ out, int64_t *__restrict__ idx,
                                                             long long total) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % C);
  const long long b = e / C;
  float best = -__builtin_inff();
  int bi = 0;
  for (int s = 0; s < S; ++s) {
    const size_t o = ((size_t)b * S + s) * C + c;
    const float v = pval[o];
    if (v > best || s == 0) { best = v; bi = pidx[o]; }
  }
  if (bias != nullptr) best = best + bias[c];  // rounding is monotonic: max_n(y_n + b) == max_n(y_n) + b
  if (relu) best = best > 0.f ? best : 0.f;    // max and ReLU commute
  out[e] = best;
  idx[e] = max(bi, 0);
}


// ---------------------------------------------------------------------------------------------------
// Fused  y = x[B*N,CIN] @ Wt[CIN,Cout]  ->  max / arg-max over the N points of each cloud, on the f32
// matrix cores (v_mfma_f32_32x32x2_f32: exact f32, a k-ordered fmaf chain).  The [B*N,Cout] activation
// (134 MB for PointNet's 128->1024 layer at B=32) never exists: not written, not re-read.
//
//   block   = 8 waves (two per SIMD); one cloud, one split of its points, 256 output channels (32 per wave)
//   W       : the wave's 32 columns x CIN rows live in VGPRs for the whole kernel (CIN/2 registers)
//   x       : 64-point tiles, global -> registers -> LDS (row stride CIN+4 floats: conflict-free
//             ds_read_b128), double buffered, one barrier per tile; every wave reads the same tile
//   compute : per tile and wave 2 (rows) x NCB (columns) accumulators of 32x32, CIN/2 MFMA steps each
//   epilogue: accumulator layout = column on the lane, 16 rows in registers -> running (max, first
//             arg-max) per lane; the two lane halves are merged once at the end.  Two accumulator sets
//             alternate: the scan of tile t runs between the MFMA groups of tile t+1.
// Partials [B,S,Cout] go through max_over_points_merge (bias, ReLU, split order = ascending points).
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LF_TM = 64;
// One accumulator set (138 VGPRs) rather than two (256 + spills): alone it is 3 % slower (71.1 vs 68.9 us at B=32), but
// it leaves half of the register file to kernels of other streams -- with three attacks in flight 16.8 vs 16.1 clouds/s.
#ifndef LM_PIPE_DEFAULT
#define LM_PIPE_DEFAULT false
#endif
constexpr bool LM_PIPE = LM_PIPE_DEFAULT;
// Two LDS tiles (67.6 KB).  One tile (33.8 KB, an extra barrier per tile) measured the same within noise, alone (71.6 us)
// and with three attacks in flight (17.57 vs 17.66 clouds/s), so the double buffer stays.
#ifndef LM_DBUF_DEFAULT
#define LM_DBUF_DEFAULT true
#endif
constexpr bool LM_DBUF = LM_DBUF_DEFAULT;

template <int CIN, int NCB, int NT, bool PIPE>
__global__ __launch_bounds__(NT) void linear_max_fwd_k(const float *__restrict__ X, const float *__restrict__ Wt,
                                                        int B_, int N, int Cout, int rows_per_split, int S, int ncg,
                                                        float *pval, int32_t *pidx, const float *__restrict__ bias,
                                                        int relu, float *__restrict__ out, int64_t *__restrict__ idx,
                                                        int *tickets) {
  constexpr int LDA = CIN + 4;
  constexpr int KS = CIN / 2;            // MFMA steps per output tile
  constexpr int F4_ROW = CIN / 4;        // float4 per row of x
  constexpr int ST = LF_TM * F4_ROW / NT;  // float4 staged per thread per tile
  extern __shared__ float4 sA4[];        // 2 x LF_TM x LDA floats
  float *sA = reinterpret_cast<float *>(sA4);
  // XCD-aware block order: consecutive workgroup ids go to consecutive XCDs (8, each with its own L2), and the NCG
  // column-group blocks of one (cloud, split) all stream the SAME x tiles -> they are given ids that are congruent
  // mod 8, so the tiles come from HBM once and from that XCD's L2 three more times (PMC: 70 MB -> see DESIGN.md).
  int cg, s, b;
  {
    const int NCG = ncg, id = blockIdx.x, nrg = S * B_;
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
  const int r = lane & 31, h = lane >> 5;
  const int col0 = cg * 256 + wave * 32 * NCB;  // NT/64 waves x 32*NCB columns = 256 per block
  const bool active = col0 < Cout;  // wave-uniform
  const int n0 = s * rows_per_split, n1 = min(N, n0 + rows_per_split);
  const int ntiles = (n1 - n0 + LF_TM - 1) / LF_TM;
  X += (size_t)b * N * CIN;

  // step t of the K loop consumes k = 8*(t/4) + 4*h + t%4: a lane's four consecutive steps are one float4 of x
  float w[NCB][KS];
  const float *wp = Wt + (active ? col0 : 0) + r;  // idle waves (Cout not a multiple of 256) load in range, use nothing
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int t = 0; t < KS; ++t) w[cb][t] = wp[(size_t)(8 * (t >> 2) + 4 * h + (t & 3)) * Cout + 32 * cb];

  float4 st[ST];
  auto fetch = [&](int tile) {
#pragma unroll
    for (int u = 0; u < ST; ++u) {
      const int e = threadIdx.x + NT * u;
      const int n = n0 + tile * LF_TM + e / F4_ROW;
      st[u] = n < n1 ? *reinterpret_cast<const float4 *>(X + (size_t)n * CIN + 4 * (e % F4_ROW))
                     : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int u = 0; u < ST; ++u) {
      const int e = threadIdx.x + NT * u;
      *reinterpret_cast<float4 *>(sA + (size_t)buf * LF_TM * LDA + (e / F4_ROW) * LDA + 4 * (e % F4_ROW)) = st[u];
    }
  };

  float bv[NCB];
  int bi[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    bv[cb] = -__builtin_inff();
    bi[cb] = -1;
  }
  // accumulator element e of tile (rb, cb): row 32*rb + (e&3) + 8*(e>>2) + 4*h, column 32*cb + r.
  // The scan costs matrix time (VALU instructions do not overlap f32 MFMAs on this chip, tools/tune/
  // mfma_valu_overlap.hip), so it is three instructions per value: compare, keep the value, keep a wave-uniform CODE
  // (tile, rb, e) of where it came from -- the point index is decoded from the code once, after the last tile.
  // (a tile's 32 values are scanned into a tile-local best whose code rb*16+e is an inline constant; the tile number
  // joins once per tile)
  float tv[NCB];
  int tc[NCB];
  auto update = [&](const f32x16 (&acc)[2][NCB], int tile, int rb, int e, bool ragged) {
    bool live = true;
    if (ragged) live = n0 + tile * LF_TM + 4 * h + 32 * rb + (e & 3) + 8 * (e >> 2) < n1;  // zero-filled rows stay out
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const float v = live ? acc[rb][cb][e] : -__builtin_inff();
      if (rb == 0 && e == 0) {
        tv[cb] = v;
        tc[cb] = 0;
      } else {
        const bool g = v > tv[cb];
        tv[cb] = g ? v : tv[cb];
        tc[cb] = g ? rb * 16 + e : tc[cb];
      }
      if (rb == 1 && e == 15) {  // earlier tiles hold earlier points: they keep ties
        const bool g = tv[cb] > bv[cb];
        bv[cb] = g ? tv[cb] : bv[cb];
        bi[cb] = g ? tile * 32 + tc[cb] : bi[cb];
      }
    }
  };
  auto decode = [&]() {  // code -> point index; nothing won (all rows -inf): the split's first point
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int c = bi[cb];
      bi[cb] = c < 0 ? n0 : n0 + (c >> 5) * LF_TM + 4 * h + 32 * ((c >> 4) & 1) + (c & 3) + 8 * ((c & 15) >> 2);
    }
  };
  auto epilogue = [&](const f32x16 (&acc)[2][NCB], int tile) {
    if (n0 + (tile + 1) * LF_TM > n1) {  // wave-uniform: only a split's last tile can be ragged
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int e = 0; e < 16; ++e) update(acc, tile, rb, e, true);
    } else {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int e = 0; e < 16; ++e) update(acc, tile, rb, e, false);
    }
  };
  // One tile of MFMAs into `cur`; the max / arg-max scan of the PREVIOUS tile's accumulators (`prev`, never ragged)
  // is spread between the MFMA groups so that the VALU work issues in the shadow of the matrix pipe.
  constexpr int EPJ = 32 / (CIN / 8);  // (rb, e) pairs scanned per K group
  auto tile_step = [&](f32x16 (&cur)[2][NCB], int buf, const f32x16 (&prev)[2][NCB], int prev_tile, bool have_prev) {
    const float *a = sA + (size_t)buf * LF_TM * LDA + r * LDA + 4 * h;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NCB; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) cur[i][j][e] = 0.f;
    // the A operands of K group j+1 are read from LDS while the MFMAs of group j run: left to itself the compiler reuses
    // one register set for every group and waits out an LDS round trip per eight MFMAs (20 % of the tile)
    float4 pa0[2], pa1[2];
    pa0[0] = *reinterpret_cast<const float4 *>(a);
    pa1[0] = *reinterpret_cast<const float4 *>(a + 32 * LDA);
#pragma unroll
    for (int j = 0; j < CIN / 8; ++j) {
      if (j + 1 < CIN / 8) {
        pa0[(j + 1) & 1] = *reinterpret_cast<const float4 *>(a + 8 * (j + 1));
        pa1[(j + 1) & 1] = *reinterpret_cast<const float4 *>(a + 32 * LDA + 8 * (j + 1));
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the reads above this group's MFMAs
      const float4 a0 = pa0[j & 1], a1 = pa1[j & 1];
      const float a0v[4] = {a0.x, a0.y, a0.z, a0.w};
      const float a1v[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int t = 4 * j + i;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) cur[0][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v[i], w[cb][t], cur[0][cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) cur[1][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v[i], w[cb][t], cur[1][cb], 0, 0, 0);
      }
      if (have_prev) {
#pragma unroll
        for (int u = 0; u < EPJ; ++u) {
          const int q = EPJ * j + u;
          update(prev, prev_tile, q >> 4, q & 15, false);
        }
      }
    }
  };

  f32x16 accA[2][NCB], accB[2][NCB];
  fetch(0);
  stash(0);
  __syncthreads();
  if (!PIPE) {  // one accumulator set: the scan of a tile follows its own MFMAs (another wave of the SIMD fills the gap)
    for (int tile = 0; tile < ntiles; ++tile) {
      const bool more = tile + 1 < ntiles;
      if (more) fetch(tile + 1);
      if (active) {
        tile_step(accA, LM_DBUF ? (tile & 1) : 0, accB, 0, false);
        epilogue(accA, tile);
      }
      if (!LM_DBUF) __syncthreads();  // single LDS tile: every wave is done reading before it is overwritten
      if (more) stash(LM_DBUF ? ((tile + 1) & 1) : 0);
      __syncthreads();
    }
  } else {
  {  // tile 0: nothing to scan yet
    const bool more = 1 < ntiles;
    if (more) fetch(1);
    if (active) tile_step(accA, 0, accB, 0, false);
    if (more) stash(1);
    __syncthreads();
  }
  int tile = 1;
  while (true) {
    if (tile >= ntiles) {
      if (active) epilogue(accA, tile - 1);
      break;
    }
    bool more = tile + 1 < ntiles;
    if (more) fetch(tile + 1);
    if (active) tile_step(accB, 1, accA, tile - 1, true);
    if (more) stash(0);
    __syncthreads();
    ++tile;
    if (tile >= ntiles) {
      if (active) epilogue(accB, tile - 1);
      break;
    }
    more = tile + 1 < ntiles;
    if (more) fetch(tile + 1);
    if (active) tile_step(accA, 0, accB, tile - 1, true);
    if (more) stash(1);
    __syncthreads();
    ++tile;
  }
  }
  decode();
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {  // the other half of the wave holds the same column, other rows
    const float ov = __shfl_xor(bv[cb], 32, HITADV_WAVE);
    const int oi = __shfl_xor(bi[cb], 32, HITADV_WAVE);
    if (ov > bv[cb] || (ov == bv[cb] && oi < bi[cb])) { bv[cb] = ov; bi[cb] = oi; }
    if (active && h == 0) {
      const int c = col0 + 32 * cb + r;
      if (S == 1 && tickets != nullptr) {  // nothing to merge: finish here
        float v = bv[cb] + (bias ? bias[c] : 0.f);  // rounding is monotonic: max_n(y_n + b) == max_n(y_n) + b
        out[(size_t)b * Cout + c] = relu ? (v > 0.f ? v : 0.f) : v;  // max and ReLU commute
        idx[(size_t)b * Cout + c] = max(bi[cb], 0);
      } else {
        const size_t o = ((size_t)b * S + s) * Cout + c;
        __hip_atomic_store(&pval[o], bv[cb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&pidx[o], bi[cb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (tickets == nullptr || S == 1) return;  // partials only: the caller merges them (max_over_points_merge)
  // The S splits of a (cloud, column group) meet here: the last block to draw the group's ticket merges the partials in
  // split order (= ascending points, so ties keep the first point), adds the bias, applies the ReLU and writes the
  // result.  Same fence-free protocol as fc_layer_k (csrc/pointnet.hip): write-through stores, vmcnt(0), barrier, ticket.
  __shared__ int s_last;
  if (!handoff_last_arriver(tickets, b * ncg + cg, S, &s_last)) return;
  const int c = cg * 256 + threadIdx.x;
  if (threadIdx.x < 256 && c < Cout) {
    float best = 0.f;
    int bidx = 0;
    for (int q = 0; q < S; ++q) {
      const size_t o = ((size_t)b * S + q) * Cout + c;
      const float v = __hip_atomic_load(&pval[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int vi = __hip_atomic_load(&pidx[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (q == 0 || v > best) { best = v; bidx = vi; }
    }
    best += bias ? bias[c] : 0.f;
    out[(size_t)b * Cout + c] = relu ? (best > 0.f ? best : 0.f) : best;
    idx[(size_t)b * Cout + c] = max(bidx, 0);
  }
  if (threadIdx.x == 0) __hip_atomic_store(&tickets[b * ncg + cg], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static void linear_max_split(int B, int N, int Cout, int *S, int *rows) {
  const int colgroups = (Cout + 255) / 256;
  // One block per CU and as many 64-point tiles per block as that allows: every block first loads its 32 KB/wave of
  // W into registers (~4 us), so fewer, longer blocks win (measured at B=32: 256 blocks 75.6 us, 512: 79.9, 1024: 86.5).
  int want = (256 + B * colgroups - 1) / (B * colgroups);
  const int maxs = (N + LF_TM - 1) / LF_TM;
  want = want < 1 ? 1 : (want > maxs ? maxs : want);
  int per = (N + want - 1) / want;
  per = (per + LF_TM - 1) / LF_TM * LF_TM;
  *rows = per;
  *S = (N + per - 1) / per;
}

}  // namespace hitadv

using namespace hitadv;

extern "C" int64_t hitadv_max_over_points_scratch(int B, int C) { return (int64_t)B * MP_SPLIT * C; }

extern "C" int hitadv_max_over_points(const float *y, int B, int N, int C, const float *bias, int relu,
                                      float *part_val, int32_t *part_idx, float *out, int64_t *idx, void *stream) {
  if (!y || !part_val || !part_idx || !out || !idx || B <= 0 || N <= 0 || C <= 0 || (C & 3) ||
      ((uintptr_t)y & 15))
    return HITADV_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((C / 4 + 255) / 256, MP_SPLIT, B);
  max_over_points_k<<<grid, 256, 0, s>>>(y, N, C, part_val, part_idx);
  const long long total = (long long)B * C;
  max_over_points_merge<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(part_val, part_idx, C, MP_SPLIT, bias, relu, out, idx,
                                                                         total);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_linear_max_bwd(const float *dg, const float *W, const int64_t *idx, const float *act_out, int B,
                                     int N, int Cout, int Cin, float *dX, void *stream) {
  if (!dg || !W || !idx || !dX || B <= 0 || N <= 0 || Cout <= 0 || Cin <= 0 || Cin > 64 * LM_MAXC || Cout > 16384)
    return HITADV_E_ARG;
  dim3 grid((N + LM_ROWS - 1) / LM_ROWS, B);
  linear_max_bwd_k<<<grid, 256, (size_t)Cout * sizeof(int) + (size_t)4 * Cin * sizeof(float), (hipStream_t)stream>>>(dg, W, idx, act_out, N, Cout, Cin, dX);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t hitadv_linear_max_fwd_scratch(int B, int N, int Cout) {
  if (B <= 0 || N <= 0 || Cout <= 0) return 0;
  int S, rows;
  linear_max_split(B, N, Cout, &S, &rows);
  return (int64_t)B * S * Cout;
}

extern "C" int hitadv_linear_max_fwd(const float *X, const float *Wt, const float *bias, int B, int N, int Cin, int Cout,
                                     int relu, float *part_val, int32_t *part_idx, float *out, int64_t *idx,
                                     int32_t *tickets, void *stream) {
  if (!X || !Wt || !part_val || !part_idx || !out || !idx || B <= 0 || N <= 0 || Cout <= 0 || (Cout & 63) ||
      (Cin != 64 && Cin != 128) || ((uintptr_t)X & 15))
    return HITADV_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  int S, rows;
  linear_max_split(B, N, Cout, &S, &rows);
  const int ncg = (Cout + 255) / 256;
  dim3 grid((unsigned)(ncg * S * B));
  const size_t shm = (size_t)(LM_DBUF ? 2 : 1) * LF_TM * (Cin + 4) * sizeof(float);
  // 8 waves x 32 columns per block: two waves per SIMD keep the matrix pipe busy across each other's LDS waits and
  // epilogue scans (measured at B=32: 69.1 us; 4 waves x 64 columns, one wave per SIMD: 73.0 us)
  // 8 waves x 32 columns per block.  Measured at B=32 (us incl. merge): two accumulator sets 68.9; one set 71.1 (chosen,
  // see LM_PIPE); 4 waves x 64 columns (one wave per SIMD) 73.0; 512 blocks 75.4; forced to 128 VGPRs 74.9-76.7.
  if (Cin == 128) {
    HITADV_RAISE_LDS((&linear_max_fwd_k<128, 1, 512, LM_PIPE>), 2 * LF_TM * 132 * 4);
    linear_max_fwd_k<128, 1, 512, LM_PIPE><<<grid, 512, shm, s>>>(X, Wt, B, N, Cout, rows, S, ncg, part_val, part_idx, bias,
                                                                relu, out, idx, tickets);
  } else {
    linear_max_fwd_k<64, 1, 512, true><<<grid, 512, shm, s>>>(X, Wt, B, N, Cout, rows, S, ncg, part_val, part_idx, bias, relu,
                                                               out, idx, tickets);
  }
  if (tickets == nullptr) {
    const long long total = (long long)B * Cout;
    max_over_points_merge<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(part_val, part_idx, Cout, S, bias, relu, out,
                                                                           idx, total);
  }
  HITADV_LAUNCH_CHECK();
  return 0;
}
