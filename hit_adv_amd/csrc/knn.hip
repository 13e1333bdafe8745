// K nearest neighbours (ascending, ties -> lower index; distances in one of the forms of common.hpp) and its backward.
//
//   K4  knn_select<KB>  K <= 32, M <= 2048: one lane per query and per quarter of the references (4 waves share 64
//                       queries), references resident in LDS (float4).  Selection without a divergent insert: the KB
//                       smallest DISTANCES are kept sorted in VGPRs by a branch-free v_med3_f32 chain, the INDEX of
//                       every accepted candidate is appended to a per-lane log in LDS, and the few logged candidates
//                       that survive are matched to their slots once, at the end.  fp32-VALU-bound.
//       knn_topk<KB>    every other size: sorted (distance, index) insertion list per lane in VGPRs.
#include <stdlib.h>

#include "common.hpp"
#include "hitadv.h"

namespace hitadv {

// diagnostic: how often a selection kernel of this file replaced a sentinel / out-of-range index (sane_index) since the
// library was loaded, per site (hitadv_debug_knn_sane_hits); 0 on every finite input
__device__ unsigned int g_knn_sane_hits[8];
__device__ __forceinline__ int sane_index_at(int site, int i, int limit, int fallback) {
  if ((unsigned)i >= (unsigned)limit) atomicAdd(&g_knn_sane_hits[site], 1u);
  return sane_index(i, limit, fallback);
}

constexpr int KNN_RCH = 1024;

// One block = 64 queries x 4 waves.  Every wave scans a quarter of each staged reference chunk for the
// SAME 64 queries (one query per lane) and keeps its own sorted KB-list in VGPRs; the four lists then
// meet in LDS and wave 0 runs a 4-way merge (K steps) per query.  Splitting the reference range, not
// the queries, is what quadruples the number of waves (2048 at B=32, N=1024) without shrinking a
// query's work below one lane.  Ties: equal distances are ordered by reference index, in the lists
// (ascending scan + stable insertion) and in the merge (explicit index compare).
template <int KB, int FORM, typename IdxT>
__global__ __launch_bounds__(256) void knn_topk(const float *__restrict__ q, const float *__restrict__ p,
                                                int N, int M, int K, float *__restrict__ dists,
                                                IdxT *__restrict__ idx) {
  __shared__ float4 sref[KNN_RCH];
  extern __shared__ float smerge[];  // [4][KB][64] distances, then [4][KB][64] indices
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const bool live = i < N;
  const float *qp = q + ((size_t)b * N + (live ? i : N - 1)) * 3;
  const float qx = qp[0], qy = qp[1], qz = qp[2];
  const float rq = sq_norm<FORM>(qx, qy, qz);
  p += (size_t)b * M * 3;
  float d[KB];
  int ix[KB];
#pragma unroll
  for (int t = 0; t < KB; ++t) {
    d[t] = __builtin_inff();
    ix[t] = 0x7fffffff;
  }
  constexpr int BUF = KB <= 4 ? 1 : (KB <= 12 ? 4 : 8);  // pending-candidate depth (1 = insert directly)
  float pd[BUF];
  int pj[BUF];
  int np = 0;
#pragma unroll
  for (int t = 0; t < BUF; ++t) {
    pd[t] = __builtin_inff();
    pj[t] = 0x7fffffff;
  }
  auto insert = [&](float c, int j) {
#pragma unroll
    for (int t = KB - 1; t > 0; --t) {
      const bool sh = c < d[t - 1];
      const bool wr = c < d[t];
      const float nd = sh ? d[t - 1] : c;
      const int ni = sh ? ix[t - 1] : j;
      d[t] = wr ? nd : d[t];
      ix[t] = wr ? ni : ix[t];
    }
    const bool w0 = c < d[0];
    d[0] = w0 ? c : d[0];
    ix[0] = w0 ? j : ix[0];
  };
  auto flush = [&]() {  // oldest first: equal distances keep ascending reference index
#pragma unroll
    for (int t = BUF - 1; t >= 0; --t)
      if (np > t) insert(pd[t], pj[t]);
    np = 0;
  };
  for (int c0 = 0; c0 < M; c0 += KNN_RCH) {
    const int cnt = min(KNN_RCH, M - c0);
    __syncthreads();
    for (int r = threadIdx.x; r < cnt; r += 256) {
      const float *s = p + (size_t)(c0 + r) * 3;
      sref[r] = make_float4(s[0], s[1], s[2], sq_norm<FORM>(s[0], s[1], s[2]));
    }
    __syncthreads();
    const int per = KNN_RCH / 4;
    const int lo = wave * per, hi = min(lo + per, cnt);
    for (int r = lo; r < hi; ++r) {
      const float4 v = sref[r];
      const float c = pair_dist<FORM>(qx, qy, qz, rq, v.x, v.y, v.z, v.w);
      // Accepted candidates (closer than the current KB-th best) are parked in a BUF-deep per-lane shift
      // register; the expensive sorted insert runs only when some lane's register is full.  The insert body is
      // executed by the whole wave whenever ANY lane needs it, so batching it per BUF accepted candidates of the
      // fastest-filling lane (instead of per reference) is what removes most of its cost.
      const bool acc = c < d[KB - 1];
      if (BUF == 1) {
        if (acc) insert(c, c0 + r);
      } else {
#pragma unroll
        for (int t = BUF - 1; t > 0; --t) {
          pd[t] = acc ? pd[t - 1] : pd[t];
          pj[t] = acc ? pj[t - 1] : pj[t];
        }
        pd[0] = acc ? c : pd[0];
        pj[0] = acc ? c0 + r : pj[0];
        np += acc ? 1 : 0;
        if (__ballot(np == BUF)) flush();
      }
    }
  }
  flush();
  float *md = smerge;
  int *mi = reinterpret_cast<int *>(smerge + 4 * KB * 64);
#pragma unroll
  for (int t = 0; t < KB; ++t) {
    md[(wave * KB + t) * 64 + lane] = d[t];
    mi[(wave * KB + t) * 64 + lane] = ix[t];
  }
  __syncthreads();
  if (wave != 0 || !live) return;
  int pos[4] = {0, 0, 0, 0};
  float hd[4];
  int hi4[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    hd[w] = md[(w * KB) * 64 + lane];
    hi4[w] = mi[(w * KB) * 64 + lane];
  }
  float *od = dists + ((size_t)b * N + i) * K;
  IdxT *oi = idx + ((size_t)b * N + i) * K;
  for (int t = 0; t < K; ++t) {
    int bw = 0;
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const bool take = hd[w] < hd[bw] || (hd[w] == hd[bw] && hi4[w] < hi4[bw]);
      bw = take ? w : bw;
    }
    float bd = hd[0];
    int bi = hi4[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      bd = bw == w ? hd[w] : bd;
      bi = bw == w ? hi4[w] : bi;
    }
    od[t] = bd;
    oi[t] = (IdxT)sane_index_at(0, bi, M, t);
    // advance the winning list
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (bw == w) {
        pos[w] += 1;
        const bool more = pos[w] < KB;
        const int a = (w * KB + (more ? pos[w] : KB - 1)) * 64 + lane;
        hd[w] = more ? md[a] : __builtin_inff();
        hi4[w] = more ? mi[a] : 0x7fffffff;
      }
    }
  }
}

// Two references at a time on the packed-f32 VALU (v_pk_add / v_pk_mul / v_pk_fma: two IEEE fp32 results per instruction,
// each rounded exactly as the scalar instruction would): the expression and its rounding points are those of pair_dist.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int FORM>
__device__ __forceinline__ f32x2 pair_dist2(float qx, float qy, float qz, float rq, f32x2 px, f32x2 py, f32x2 pz, f32x2 rp) {
  const f32x2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
  if (FORM == 0) {
    const f32x2 dx = qx2 - px, dy = qy2 - py, dz = qz2 - pz;
    return (dx * dx + dy * dy) + dz * dz;
  }
  const f32x2 zz = __builtin_elementwise_fma(qz2, pz, __builtin_elementwise_fma(qy2, py, qx2 * px));
  const f32x2 m2 = {-2.0f, -2.0f}, rq2 = {rq, rq};
  if (FORM == 1) return __builtin_elementwise_fma(m2, zz, rq2 + rp);
  if (FORM == 3) return __builtin_elementwise_fma(m2, zz, rq2) + rp;
  return __builtin_elementwise_fma(m2, zz, rp) + rq2;
}

// ------------------------------------------------------------------------------------ K4 (K <= 32, M <= 2048)
// Why not a sorted (distance, index) insertion list: its body (4 VALU instructions per slot) is executed by the whole wave
// whenever ANY of the 64 lanes accepts a candidate, which for K >= 5 is at almost every reference -- the r01 kernel spent
// ~110 instructions per reference at K = 17.  Here a reference costs the distance, KB v_med3_f32 (new L[s] =
// med3(L[s-1], c, L[s]) keeps the KB smallest distances sorted, no branch, no index) and one masked 2-byte LDS store.
//   * acceptance is strict (c < L[KB-1]) and the scan ascends in reference index, so of equal distances the lower
//     indices are the ones that get in;
//   * a lane's log holds every candidate it ever accepted, in ascending index; the final K best are among them (the
//     threshold only falls).  When a log fills up (KS_CAP >= 2 KB entries) it is compacted in place to the entries that
//     can still matter (distance <= current threshold: at most KB-1 below it and KB equal to it);
//   * at the end the surviving entries are matched to the slots of the sorted distance list (equal distances: in log =
//     index order), distances re-evaluated from the LDS-resident references with the same expression, bit for bit;
//   * the four waves' lists meet in LDS and wave 0 merges them (K steps, index compare on equal distances).
// NW waves of a block split the reference range for the same 64 queries (NW = 8 from 512 references on: 4096 short waves
// instead of 2048 give every SIMD four waves to switch between -- the scan is a chain of dependent VALU instructions).
__host__ __device__ constexpr int ks_cap(int KB, int NW) {  // >= 2 KB - 1 survivors of a compaction + one group of 4
  return NW == 8 ? (KB <= 8 ? 48 : (KB <= 12 ? 56 : (KB <= 20 ? 64 : 72))) : (KB <= 8 ? 48 : (KB <= 16 ? 80 : 96));
}
__host__ __device__ constexpr size_t ks_union_bytes(int KB, int NW) {
  return (size_t)NW * 64 * (ks_cap(KB, NW) * 2 > KB * 8 ? ks_cap(KB, NW) * 2 : KB * 8);
}

template <int KB, int FORM, int NW>
__global__ __launch_bounds__(NW * 64) void knn_select(const float *__restrict__ q, const float *__restrict__ p, int N,
                                                      int M, int K, float *__restrict__ dists,
                                                      void *__restrict__ idx_out, int idx_is_i64) {
  constexpr int CAP = ks_cap(KB, NW);
  static_assert(CAP >= 2 * KB + 3, "a compaction must leave room for one group of four candidates");
  extern __shared__ __attribute__((aligned(16))) char ks_smem[];
  // references live in LDS as PAIRS: sref[2 m] = (x0, x1, y0, y1), sref[2 m + 1] = (z0, z1, n0, n1) of references 2 m, 2 m + 1
  // (n = |p|^2 in the kernel's form).  Every wave's range is a multiple of 4 references; the slots past M hold a sentinel
  // whose distance to anything is +inf, so the scan needs no bounds test: +inf is never below a threshold.
  const int per = (((M + NW - 1) / NW) + 3) & ~3;
  const int Mpad = per * NW;
  float4 *sref = reinterpret_cast<float4 *>(ks_smem);                                       // [Mpad]
  unsigned short *slog = reinterpret_cast<unsigned short *>(ks_smem + (size_t)Mpad * 16);   // [NW][CAP][64]
  float *md = reinterpret_cast<float *>(ks_smem + (size_t)Mpad * 16);                       // after the scan: [NW][KB][64]
  int *mi = reinterpret_cast<int *>(md + NW * KB * 64);
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = blockIdx.x * 64 + lane;
  const bool live = i < N;
  const float *qp = q + ((size_t)b * N + (live ? i : N - 1)) * 3;
  const float qx = qp[0], qy = qp[1], qz = qp[2];
  const float rq = sq_norm<FORM>(qx, qy, qz);
  p += (size_t)b * M * 3;
  {
    float *sf = reinterpret_cast<float *>(sref);
    for (int r = threadIdx.x; r < Mpad; r += NW * 64) {
      const float *s = p + (size_t)min(r, M - 1) * 3;
      const bool in = r < M;
      // sentinel: direct form x = +inf (dx * dx = +inf); Gram forms p = 0, |p|^2 = +inf
      const float x = in ? s[0] : (FORM == 0 ? __builtin_inff() : 0.f), y = in ? s[1] : 0.f, z = in ? s[2] : 0.f;
      const float n = in ? sq_norm<FORM>(x, y, z) : __builtin_inff();
      float *o = sf + (size_t)(r >> 1) * 8 + (r & 1);
      o[0] = x;
      o[2] = y;
      o[4] = z;
      o[6] = n;
    }
  }
  __syncthreads();
  float L[KB];
  int I[KB];
#pragma unroll
  for (int t = 0; t < KB; ++t) {
    L[t] = __builtin_inff();
    I[t] = 0x7fffffff;
  }
  const int lo = wave * per, hi = lo + per;  // a multiple of 4 references, sentinels included
  auto dist_of = [&](int j) {  // one reference (compaction, slot matching): the same expression, scalar
    const float *o = reinterpret_cast<const float *>(sref) + (size_t)(j >> 1) * 8 + (j & 1);
    return pair_dist<FORM>(qx, qy, qz, rq, o[0], o[2], o[4], o[6]);
  };
  auto dist4 = [&](int r0, float (&c)[4]) {  // references r0 .. r0 + 3 (r0 % 4 == 0)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float4 a = sref[r0 + 2 * u], bq = sref[r0 + 2 * u + 1];
      const f32x2 d = pair_dist2<FORM>(qx, qy, qz, rq, f32x2{a.x, a.y}, f32x2{a.z, a.w}, f32x2{bq.x, bq.y}, f32x2{bq.z, bq.w});
      c[2 * u] = d.x;
      c[2 * u + 1] = d.y;
    }
  };
  if (KB == 1) {  // nearest neighbour: the index rides along (strict <, ascending scan: lowest index on ties)
    for (int r0 = lo; r0 < hi; r0 += 4) {
      float c[4];
      dist4(r0, c);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool acc = c[u] < L[0];
        L[0] = acc ? c[u] : L[0];
        I[0] = acc ? r0 + u : I[0];
      }
    }
  } else {
    unsigned short *mylog = slog + (size_t)wave * CAP * 64 + lane;
    int cnt = 0;
    // keep the logged candidates that can still be among the K best, in order.  Eight entries at a time: their indices, then
    // their references, are read as independent LDS loads (one round trip each for the group, not per entry)
    auto compact = [&]() {
      const float tau = L[KB - 1];
      int w = 0;
      int most = cnt;
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) most = max(most, __shfl_xor(most, m, 64));
      most = __builtin_amdgcn_readfirstlane(most);
      for (int e0 = 0; e0 < most; e0 += 8) {
        int jj[8];
        float cc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) jj[u] = e0 + u < cnt ? (int)mylog[min(e0 + u, CAP - 1) * 64] : 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) cc[u] = dist_of(jj[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (e0 + u < cnt && cc[u] <= tau) {
            mylog[w * 64] = (unsigned short)jj[u];  // w <= e0 + u: never ahead of the entries still to be read
            ++w;
          }
        }
      }
      cnt = w;
    };
    for (int r0 = lo; r0 < hi; r0 += 4) {
      // four independent distance chains first (their LDS reads are in flight before the first log store) ...
      float c[4];
      dist4(r0, c);
      if (__builtin_amdgcn_ballot_w64(cnt > CAP - 4)) compact();  // room for this group in every lane's log
      // ... then the four insertions: a branch-free median chain on the distances, a masked 2-byte store of the index
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool acc = c[u] < L[KB - 1];
        const float cu = c[u];
#pragma unroll
        for (int t = KB - 1; t > 0; --t) L[t] = __builtin_amdgcn_fmed3f(L[t - 1], cu, L[t]);
        L[0] = __builtin_amdgcn_fmed3f(-__builtin_inff(), cu, L[0]);  // = min(c, L[0]) in one instruction
        // the index goes to the log's next slot whatever happens (no exec-mask juggling); an acceptance claims the slot
        mylog[cnt * 64] = (unsigned short)(r0 + u);
        cnt += acc ? 1 : 0;
      }
    }
    // the survivors, matched to their slots (entries of equal distance fill equal slots in log = index order)
    compact();
    int most = cnt;  // wave-uniform trip count: the longest surviving log (<= 2 KB - 1)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) most = max(most, __shfl_xor(most, m, 64));
    most = __builtin_amdgcn_readfirstlane(most);
    for (int e0 = 0; e0 < most; e0 += 4) {
      int jj[4];
      float cc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) jj[u] = e0 + u < cnt ? (int)mylog[min(e0 + u, CAP - 1) * 64] : 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) cc[u] = dist_of(jj[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        bool placed = !(e0 + u < cnt);
#pragma unroll
        for (int t = 0; t < KB; ++t) {
          const bool hit = !placed && cc[u] == L[t] && I[t] == 0x7fffffff;
          I[t] = hit ? jj[u] : I[t];
          placed = placed || hit;
        }
      }
    }
    __syncthreads();  // every wave is done with its log: the area becomes the merge buffers
  }
#pragma unroll
  for (int t = 0; t < KB; ++t) {
    md[(wave * KB + t) * 64 + lane] = L[t];
    mi[(wave * KB + t) * 64 + lane] = I[t];
  }
  __syncthreads();
  if (wave != 0 || !live) return;
  int pos[NW];
  float hd[NW];
  int hi4[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    pos[w] = 0;
    hd[w] = md[(w * KB) * 64 + lane];
    hi4[w] = mi[(w * KB) * 64 + lane];
  }
  float *od = dists + ((size_t)b * N + i) * K;
  int64_t *o64 = reinterpret_cast<int64_t *>(idx_out) + ((size_t)b * N + i) * K;
  int32_t *o32 = reinterpret_cast<int32_t *>(idx_out) + ((size_t)b * N + i) * K;
  for (int t = 0; t < K; ++t) {
    int bw = 0;
    float bd = hd[0];
    int bi = hi4[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const bool take = hd[w] < bd || (hd[w] == bd && hi4[w] < bi);
      bw = take ? w : bw;
      bd = take ? hd[w] : bd;
      bi = take ? hi4[w] : bi;
    }
    od[t] = bd;
    bi = sane_index_at(1, bi, M, t);
    if (idx_is_i64) o64[t] = bi;
    else o32[t] = bi;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      if (bw == w) {
        pos[w] += 1;
        const bool more = pos[w] < KB;
        const int a = (w * KB + (more ? pos[w] : KB - 1)) * 64 + lane;
        hd[w] = more ? md[a] : __builtin_inff();
        hi4[w] = more ? mi[a] : 0x7fffffff;
      }
    }
  }
}

// K smallest (or largest) entries of every row of a materialised matrix P[B*N, M], ascending
// (descending) with ties -> lower column index: the selection half of DGCNN's feature-space kNN
// (model/dgcnn_cls.py:7-13: Gram matrix by GEMM, then topk).  One lane per row; a 64-row x 64-column tile is
// staged through LDS so that the global reads are coalesced along the row (each wave reads 64 x 256 B
// segments) and the per-lane reads are conflict-free (row pitch 65 words).
template <int KB, bool LARGEST>
__global__ __launch_bounds__(64) void topk_rows(const float *__restrict__ P, long long rows, int M, int K,
                                                float *__restrict__ vals, int64_t *__restrict__ idx) {
  __shared__ float tile[64 * 65];
  const int lane = threadIdx.x;
  const long long r0 = (long long)blockIdx.x * 64;
  const long long row = r0 + lane;
  const bool live = row < rows;
  float d[KB];
  int ix[KB];
#pragma unroll
  for (int t = 0; t < KB; ++t) {
    d[t] = __builtin_inff();
    ix[t] = 0x7fffffff;
  }
  for (int c0 = 0; c0 < M; c0 += 64) {
    const int cnt = min(64, M - c0);
    __syncthreads();
    for (int rr = 0; rr < 64; ++rr) {
      const long long gr = r0 + rr;
      if (gr < rows && lane < cnt) tile[rr * 65 + lane] = P[gr * M + c0 + lane];
    }
    __syncthreads();
    for (int cc = 0; cc < cnt; ++cc) {
      float c = tile[lane * 65 + cc];
      if (LARGEST) c = -c;
      if (c < d[KB - 1]) {
        const int j = c0 + cc;
#pragma unroll
        for (int t = KB - 1; t > 0; --t) {
          const bool sh = c < d[t - 1];
          const bool wr = c < d[t];
          const float nd = sh ? d[t - 1] : c;
          const int ni = sh ? ix[t - 1] : j;
          d[t] = wr ? nd : d[t];
          ix[t] = wr ? ni : ix[t];
        }
        const bool w0 = c < d[0];
        d[0] = w0 ? c : d[0];
        ix[0] = w0 ? j : ix[0];
      }
    }
  }
  if (live) {
#pragma unroll
    for (int t = 0; t < KB; ++t)
      if (t < K) {
        vals[row * K + t] = LARGEST ? -d[t] : d[t];
        idx[row * K + t] = sane_index_at(2, ix[t], M, t);
      }
  }
}

template <typename IdxT>
__global__ __launch_bounds__(256) void knn_bwd_q(const float *__restrict__ q, const float *__restrict__ p,
                                                 const IdxT *__restrict__ idx, const float *__restrict__ g,
                                                 int N, int M, int K, float *__restrict__ grad_q) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float *qp = q + ((size_t)b * N + i) * 3;
  const float qx = qp[0], qy = qp[1], qz = qp[2];
  p += (size_t)b * M * 3;
  float ax = 0.f, ay = 0.f, az = 0.f;
  for (int t = 0; t < K; ++t) {
    const int j = sane_index_at(3, (int)idx[((size_t)b * N + i) * K + t], M, 0);  // (a caller's table: never a wild read)
    const float g2 = 2.0f * g[((size_t)b * N + i) * K + t];
    ax = ax + g2 * (qx - p[j * 3]);
    ay = ay + g2 * (qy - p[j * 3 + 1]);
    az = az + g2 * (qz - p[j * 3 + 2]);
  }
  float *o = grad_q + ((size_t)b * N + i) * 3;
  o[0] = ax;
  o[1] = ay;
  o[2] = az;
}

// grad_p[j] = -sum_{(i,t): idx[i,t]==j} 2 g[i,t] (q_i - p_j): "owner computes", no atomics.  A block owns 64 reference points
// (lane = reference); the (idx, g, q_i) triples stream through LDS in query order and the block's four waves each scan every
// fourth group of sixteen entries -- four broadcast 16-byte LDS reads in flight instead of one read and one branch per entry,
// which made the first version wait an LDS round trip 6144 times per lane (454 us at B = 32, N = 1024, K = 6 on 128 blocks).
// A lane's hits are added in entry order and the four waves' sums in wave order: a fixed summation order.
constexpr int KB_ENT = 4096;
template <typename IdxT>
__global__ __launch_bounds__(256) void knn_bwd_p(const float *__restrict__ q, const float *__restrict__ p,
                                                 const IdxT *__restrict__ idx, const float *__restrict__ g,
                                                 int N, int M, int K, float *__restrict__ grad_p) {
  __shared__ __attribute__((aligned(16))) int sidx[KB_ENT + 16];
  __shared__ float sg[KB_ENT];
  __shared__ float sq[3 * 1024];
  __shared__ float part[3][3][64];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const bool live = j < M;
  const float *pp = p + ((size_t)b * M + (live ? j : M - 1)) * 3;
  const float px = pp[0], py = pp[1], pz = pp[2];
  const int qch = min(KB_ENT / K, 1024);  // queries per chunk
  float ax = 0.f, ay = 0.f, az = 0.f;
  for (int i0 = 0; i0 < N; i0 += qch) {
    const int nq = min(qch, N - i0);
    __syncthreads();
    for (int e = threadIdx.x; e < nq * K; e += 256) {
      sidx[e] = (int)idx[((size_t)b * N + i0) * K + e];
      sg[e] = 2.0f * g[((size_t)b * N + i0) * K + e];
    }
    for (int e = threadIdx.x; e < nq * 3; e += 256) sq[e] = q[((size_t)b * N + i0) * 3 + e];
    if (threadIdx.x < 16) sidx[nq * K + threadIdx.x] = -1;  // whole groups of 16 entries are read
    __syncthreads();
    const int ne = nq * K;
    for (int e0 = 16 * wave; e0 < ne; e0 += 64) {
      const int4 v0 = *reinterpret_cast<const int4 *>(sidx + e0), v1 = *reinterpret_cast<const int4 *>(sidx + e0 + 4);
      const int4 v2 = *reinterpret_cast<const int4 *>(sidx + e0 + 8), v3 = *reinterpret_cast<const int4 *>(sidx + e0 + 12);
      const int vv[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
      bool any = false;
#pragma unroll
      for (int u = 0; u < 16; ++u) any = any || vv[u] == j;
      if (any) {  // a hit is rare: K per reference on average
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (vv[u] == j) {
            const int e = e0 + u, i = e / K;
            const float g2 = sg[e];
            ax = ax - g2 * (sq[i * 3] - px);
            ay = ay - g2 * (sq[i * 3 + 1] - py);
            az = az - g2 * (sq[i * 3 + 2] - pz);
          }
      }
    }
  }
  if (wave > 0) part[wave - 1][0][lane] = ax, part[wave - 1][1][lane] = ay, part[wave - 1][2][lane] = az;
  __syncthreads();
  if (wave == 0 && live) {
#pragma unroll
    for (int w = 0; w < 3; ++w) ax += part[w][0][lane], ay += part[w][1][lane], az += part[w][2][lane];
    float *o = grad_p + ((size_t)b * M + j) * 3;
    o[0] = ax;
    o[1] = ay;
    o[2] = az;
  }
}

template <int FORM, typename IdxT>
static int launch_knn_topk(const float *q, const float *p, int B, int N, int M, int K, float *dists, IdxT *idx,
                           hipStream_t s) {
  dim3 grid((N + 63) / 64, B);
#define HITADV_KNN_CASE(KB)                                                                          \
  if (K <= KB) {                                                                                     \
    const size_t shm = (size_t)8 * KB * 64 * sizeof(float);                                          \
    if (shm > 48 * 1024) HITADV_RAISE_LDS((&knn_topk<KB, FORM, IdxT>), shm);                         \
    knn_topk<KB, FORM, IdxT><<<grid, 256, shm, s>>>(q, p, N, M, K, dists, idx);                      \
    return 0;                                                                                        \
  }
  if (FORM == 0) {
    HITADV_KNN_CASE(4)
    HITADV_KNN_CASE(8)
    HITADV_KNN_CASE(16)
    HITADV_KNN_CASE(32)
  } else {
    HITADV_KNN_CASE(8)
  }
  HITADV_KNN_CASE(64)
#undef HITADV_KNN_CASE
  return HITADV_E_ARG;
}

template <int FORM, int NW>
static int launch_knn_select(const float *q, const float *p, int B, int N, int M, int K, float *dists, void *idx,
                             int idx_is_i64, hipStream_t s) {
  dim3 grid((N + 63) / 64, B);
  const size_t refs = (size_t)((((M + NW - 1) / NW + 3) & ~3) * NW) * 16;  // every wave's range padded to a multiple of 4
#define HITADV_KS_CASE(KB)                                                                                     \
  if (K <= KB) {                                                                                               \
    const size_t shm = refs + ks_union_bytes(KB, NW);                                                          \
    HITADV_RAISE_LDS((&knn_select<KB, FORM, NW>), 32768 + 512 + (int)ks_union_bytes(KB, NW));                  \
    knn_select<KB, FORM, NW><<<grid, NW * 64, shm, s>>>(q, p, N, M, K, dists, idx, idx_is_i64);                \
    return 0;                                                                                                  \
  }
  if (FORM == 0) {
    HITADV_KS_CASE(1)
    HITADV_KS_CASE(4)
    HITADV_KS_CASE(5)
    HITADV_KS_CASE(6)
    HITADV_KS_CASE(8)
    HITADV_KS_CASE(12)
    HITADV_KS_CASE(16)
    HITADV_KS_CASE(17)
    HITADV_KS_CASE(20)
    HITADV_KS_CASE(24)
  } else if (FORM == 2) {  // KNNDist calls with K = k + 1 = 5 or 6 (util/dist_utils.py:156)
    HITADV_KS_CASE(5)
    HITADV_KS_CASE(6)
    HITADV_KS_CASE(16)
  } else {
    HITADV_KS_CASE(8)
  }
  HITADV_KS_CASE(32)
#undef HITADV_KS_CASE
  return HITADV_E_ARG;
}

}  // namespace hitadv

using namespace hitadv;

extern "C" int hitadv_knn_points(const float *q, const float *p, int B, int N, int M, int K, int form, float *dists,
                                 void *idx, int idx_is_i64, void *stream) {
  if (!q || !p || !dists || !idx || B <= 0 || N <= 0 || M <= 0 || K <= 0 || K > 64 || K > M)
    return HITADV_E_ARG;
  if (form != HITADV_FORM_DIRECT && form != HITADV_FORM_GRAM_KNN && form != HITADV_FORM_SQUARE_DISTANCE) return HITADV_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (form == HITADV_FORM_SQUARE_DISTANCE) {  // PCT's knn_point: K = 32 over 1024 / 512 points (model/pct_cls.py:48-53)
    if (K <= 32 && M <= 2048)
      rc = launch_knn_select<3, 4>(q, p, B, N, M, K, dists, idx, idx_is_i64, s);
    else
      rc = idx_is_i64 ? launch_knn_topk<3, int64_t>(q, p, B, N, M, K, dists, (int64_t *)idx, s)
                      : launch_knn_topk<3, int32_t>(q, p, B, N, M, K, dists, (int32_t *)idx, s);
  } else if (K <= 32 && M <= 2048) {
    // measured at B = 32, 1024 x 1024 (profiles/r02_kbench.json): 8 waves per block win for short lists (K = 1: 14 vs 29 us,
    // K = 6: 37 vs 43), 4 waves for long ones (K = 17: 74 vs 107 -- the 8-way merge and the doubled log area cost more
    // than the extra waves hide).  Two queries per lane (128-query blocks, as K2 does with four) was also measured: it
    // halves the number of waves and lost -- K = 6: 50 us, K = 17: 107 us -- thread-level parallelism is what this scan needs.
    static const int force_nw = [] { const char *e = getenv("HITADV_KNN_NW"); return e ? atoi(e) : 0; }();  // tuning only
    if (force_nw == 8 || (force_nw != 4 && M >= 512 && K <= 8))
      rc = form == HITADV_FORM_DIRECT ? launch_knn_select<0, 8>(q, p, B, N, M, K, dists, idx, idx_is_i64, s)
                                      : launch_knn_select<2, 8>(q, p, B, N, M, K, dists, idx, idx_is_i64, s);
    else
      rc = form == HITADV_FORM_DIRECT ? launch_knn_select<0, 4>(q, p, B, N, M, K, dists, idx, idx_is_i64, s)
                                      : launch_knn_select<2, 4>(q, p, B, N, M, K, dists, idx, idx_is_i64, s);
  } else if (form == HITADV_FORM_DIRECT) {
    rc = idx_is_i64 ? launch_knn_topk<0, int64_t>(q, p, B, N, M, K, dists, (int64_t *)idx, s)
                    : launch_knn_topk<0, int32_t>(q, p, B, N, M, K, dists, (int32_t *)idx, s);
  } else {
    rc = idx_is_i64 ? launch_knn_topk<2, int64_t>(q, p, B, N, M, K, dists, (int64_t *)idx, s)
                    : launch_knn_topk<2, int32_t>(q, p, B, N, M, K, dists, (int32_t *)idx, s);
  }
  if (rc) return rc;
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_topk_rows(const float *P, int64_t rows, int M, int K, int largest, float *vals, int64_t *idx,
                                void *stream) {
  if (!P || !vals || !idx || rows <= 0 || M <= 0 || K <= 0 || K > 64 || K > M) return HITADV_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((rows + 63) / 64);
#define HITADV_TOPK_CASE(KB)                                                           \
  if (K <= KB) {                                                                       \
    if (largest)                                                                       \
      topk_rows<KB, true><<<grid, 64, 0, s>>>(P, rows, M, K, vals, idx);               \
    else                                                                               \
      topk_rows<KB, false><<<grid, 64, 0, s>>>(P, rows, M, K, vals, idx);              \
    HITADV_LAUNCH_CHECK();                                                             \
    return 0;                                                                          \
  }
  HITADV_TOPK_CASE(4)
  HITADV_TOPK_CASE(8)
  HITADV_TOPK_CASE(16)
  HITADV_TOPK_CASE(24)
  HITADV_TOPK_CASE(32)
  HITADV_TOPK_CASE(48)
  HITADV_TOPK_CASE(64)
#undef HITADV_TOPK_CASE
  return HITADV_E_ARG;
}

extern "C" int hitadv_debug_knn_sane_hits(unsigned int *host8) {
  return host8 && hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_knn_sane_hits), 8 * sizeof(unsigned int)) == hipSuccess ? 0 : HITADV_E_ARG;
}

extern "C" int hitadv_knn_points_bwd(const float *q, const float *p, const void *idx, int idx_is_i64,
                                     const float *g_dists, int B, int N, int M, int K, float *grad_q,
                                     float *grad_p, void *stream) {
  if (!q || !p || !idx || !g_dists || B <= 0 || N <= 0 || M <= 0 || K <= 0 || K > 64) return HITADV_E_ARG;
  if (!grad_q && !grad_p) return HITADV_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (grad_q) {
    dim3 grid((N + 255) / 256, B);
    if (idx_is_i64)
      knn_bwd_q<int64_t><<<grid, 256, 0, s>>>(q, p, (const int64_t *)idx, g_dists, N, M, K, grad_q);
    else
      knn_bwd_q<int32_t><<<grid, 256, 0, s>>>(q, p, (const int32_t *)idx, g_dists, N, M, K, grad_q);
  }
  if (grad_p) {
    dim3 grid((M + 63) / 64, B);
    if (idx_is_i64)
      knn_bwd_p<int64_t><<<grid, 256, 0, s>>>(q, p, (const int64_t *)idx, g_dists, N, M, K, grad_p);
    else
      knn_bwd_p<int32_t><<<grid, 256, 0, s>>>(q, p, (const int32_t *)idx, g_dists, N, M, K, grad_p);
  }
  HITADV_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// k nearest neighbours in FEATURE space (DGCNN's dynamic graph, model/dgcnn_cls.py:7-13) without the [B,N,N] score
// matrix: scores  s_ij = (-|x_i|^2 + 2 x_i.x_j) - |x_j|^2  (the reference's expression, larger = closer) come out of the
// matrix cores tile by tile and go straight into per-lane sorted lists.
//   block = 8 waves = 128 queries of one cloud; reference points stream through LDS 128 (D = 64) or 64 (D = 128) at a time
//   (32-point tiles, double buffered, shared by the waves); waves w and w+4 hold the same 32 queries' features in registers as the MFMA
//   B operand and take the even / odd tile, so the accumulator has the QUERY on the lane and 16 reference points in
//   registers: a lane scans its 16 scores against its own list (KB entries, compile-time indices, no scratch).  Each
//   query is served by four lanes (two waves x the halves of a wave, disjoint reference rows), so a list sees a quarter
//   of the candidates; the four lists are merged at the end.  Ties -> lower index.
//   Measured at B=32, N=1024, K=5 with f32 MFMAs (tools/tune/knnfeat_tune.hip): D=64 74 us, of which 47 the bare MFMA +
//   staging loop and 27 the selection (VALU time adds to MFMA time on this chip, see below); D=128 111 us (87 + 24).  The
//   scores now come from the bf16 matrix cores at fp32 accuracy (three-piece split, below).
namespace hitadv {

typedef float f32x16_k __attribute__((ext_vector_type(16)));

// Scores on the bf16 matrix cores at fp32 accuracy (the three-piece split of csrc/victim_bf3.hip: both operands carried as
// three bf16 numbers that sum to the fp32 value exactly, six MFMAs per 16 values of k in an fp32 accumulator): 2.7x less
// matrix time than v_mfma_f32_32x32x2_f32.  The 32x32 output shape is kept, so the scan below is unchanged.
//   references: fp32 -> registers -> split -> three bf16 LDS images per buffer (row stride 2 D + 16 bytes: conflict-free
//               ds_read_b128), KF_STEP(D) = 128 (D = 64) or 64 (D = 128) points per step, double buffered;
//   queries   : the lane's query row as the B operand, three pieces x D/16 slices in registers for the whole kernel.
__host__ __device__ constexpr int kf_step(int D) { return D <= 64 ? 128 : 64; }
__host__ __device__ constexpr int kf_row_bytes(int D) { return 2 * D + 16; }
// LDS bytes of knn_feat_k: the double-buffered reference images (2 x 3 x KF_STEP rows) or, after the scan, the 4 x 128 lists
__host__ __device__ constexpr int knn_feat_main_bytes(int D, int KB) {
  return 2 * 3 * kf_step(D) * kf_row_bytes(D) > 128 * 4 * KB * 2 * 4 ? 2 * 3 * kf_step(D) * kf_row_bytes(D) : 128 * 4 * KB * 2 * 4;
}
__host__ __device__ constexpr int knn_feat_lds_bytes(int D, int KB) {
  return knn_feat_main_bytes(D, KB) + 2 * kf_step(D) * 4;
}

template <int D, int KB>
__global__ __launch_bounds__(512) void knn_feat_k(const float *__restrict__ X, const float *__restrict__ xx, int N, int K,
                                                  int64_t *__restrict__ idx) {
  constexpr int STEP = kf_step(D), SUB = STEP / 64;  // reference points per step; 32-point tiles per wave per step
  constexpr int RS = kf_row_bytes(D), PIECE = STEP * RS, NSL = D / 16;
  constexpr int ST = STEP * (D / 8) / 512;  // groups of 8 values staged per thread per step
  extern __shared__ float4 knn_feat_sm[];  // knn_feat_lds_bytes<D,KB>(): max(reference images, final merge) + |x|^2
  char *sB = reinterpret_cast<char *>(knn_feat_sm);
  float *sR = reinterpret_cast<float *>(knn_feat_sm);  // the merge buffers alias the images once the scan is over
  float(*sXX)[STEP] = reinterpret_cast<float(*)[STEP]>(sB + knn_feat_main_bytes(D, KB));
  const int b = blockIdx.y, q0 = blockIdx.x * 128;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int qg = wave & 3, half = wave >> 2;  // query group; which tile of a step
  const int r = lane & 31, h = lane >> 5;
  X += (size_t)b * N * D;
  xx += (size_t)b * N;
  const int q = q0 + 32 * qg + r;         // this lane's query
  const int qc = q < N ? q : N - 1;       // clamp for loads
  uint4 qw[3][NSL];  // B operand: column = the query, k = 16 j + 8 h .. + 7, three pieces
#pragma unroll
  for (int j = 0; j < NSL; ++j) {
    const float *src = X + (size_t)qc * D + 16 * j + 8 * h;
    split3x8(*reinterpret_cast<const float4 *>(src), *reinterpret_cast<const float4 *>(src + 4), qw[0][j], qw[1][j], qw[2][j]);
  }
  const float qxx = xx[qc];
  float lv[KB];
  int li[KB];
#pragma unroll
  for (int t = 0; t < KB; ++t) {
    lv[t] = -__builtin_inff();
    li[t] = 0x7fffffff;
  }
#ifndef KF_BUF
#define KF_BUF 3  // measured with the bf16x3 scores (D = 64 / 128, K = 5): depth 1: 54.1 / 74.6 us, 2: 48.8 / 70.0, 3: 48.2 / 70.5, 4: 49.8 / 73.9, 6: 62 / 83
#endif
#ifndef KF_MODE
#define KF_MODE 0  // tuning only (tools/tune/knnfeat_tune.hip): 1 = never flush, 2 = no selection at all
#endif
  constexpr int BUF = KF_BUF;  // pending-candidate depth
  float pv[BUF];
  int pj[BUF];
  int np = 0;
#pragma unroll
  for (int t = 0; t < BUF; ++t) {
    pv[t] = -__builtin_inff();
    pj[t] = 0x7fffffff;
  }
  auto insert = [&](float s, int j) {  // behind every entry >= s: equal scores keep ascending reference index
#pragma unroll
    for (int t = KB - 1; t > 0; --t) {
      const bool sh = s > lv[t - 1];
      const bool wr = s > lv[t];
      const float nv = sh ? lv[t - 1] : s;
      const int ni = sh ? li[t - 1] : j;
      lv[t] = wr ? nv : lv[t];
      li[t] = wr ? ni : li[t];
    }
    const bool w0 = s > lv[0];
    lv[0] = w0 ? s : lv[0];
    li[0] = w0 ? j : li[0];
  };
  auto flush = [&]() {  // oldest first
#pragma unroll
    for (int t = BUF - 1; t >= 0; --t)
      if (np > t) insert(pv[t], pj[t]);
    np = 0;
  };
  const int ntiles = (N + 31) / 32, nsteps = (N + STEP - 1) / STEP;
  float4 st[ST][2];
  float stx = 0.f;
  auto fetch = [&](int step) {
#pragma unroll
    for (int u = 0; u < ST; ++u) {
      const int e = threadIdx.x + 512 * u;
      const int n = min(step * STEP + e / (D / 8), N - 1);  // rows past N read row N - 1: their |x|^2 of +inf keeps them out
      const float *src = X + (size_t)n * D + 8 * (e % (D / 8));
      st[u][0] = *reinterpret_cast<const float4 *>(src);
      st[u][1] = *reinterpret_cast<const float4 *>(src + 4);
    }
    if (threadIdx.x < STEP) stx = step * STEP + threadIdx.x < N ? xx[step * STEP + threadIdx.x] : __builtin_inff();
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int u = 0; u < ST; ++u) {
      const int e = threadIdx.x + 512 * u;
      uint4 p1, p2, p3;
      split3x8(st[u][0], st[u][1], p1, p2, p3);
      char *dst = sB + (size_t)buf * 3 * PIECE + (e / (D / 8)) * RS + 16 * (e % (D / 8));
      *reinterpret_cast<uint4 *>(dst) = p1;
      *reinterpret_cast<uint4 *>(dst + PIECE) = p2;
      *reinterpret_cast<uint4 *>(dst + 2 * PIECE) = p3;
    }
    if (threadIdx.x < STEP) sXX[buf][threadIdx.x] = stx;
  };
  fetch(0);
  stash(0);
  __syncthreads();
  // MFMA and VALU instructions of one wave do not overlap on gfx950 (tools/tune/mfma_valu_overlap.hip), so the selection's
  // VALU work is paid on top of the matrix time whatever the schedule; it simply follows the tile's products.
  for (int step = 0; step < nsteps; ++step) {
    const bool more = step + 1 < nsteps;
    if (more) fetch(step + 1);
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
      const int trow = 32 * (2 * sub + half);  // this wave's tile inside the step: waves w / w+4 alternate
      const int tile = (STEP / 32) * step + 2 * sub + half;
      if (tile < ntiles) {
        const char *a = sB + (size_t)(step & 1) * 3 * PIECE + (trow + r) * RS + 16 * h;
        f32x16_k acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
          const bf16x8 f0 = as_bf16x8(*reinterpret_cast<const uint4 *>(a + 32 * j));
          const bf16x8 f1 = as_bf16x8(*reinterpret_cast<const uint4 *>(a + PIECE + 32 * j));
          const bf16x8 f2 = as_bf16x8(*reinterpret_cast<const uint4 *>(a + 2 * PIECE + 32 * j));
          const bf16x8 w0 = as_bf16x8(qw[0][j]), w1 = as_bf16x8(qw[1][j]), w2 = as_bf16x8(qw[2][j]);
          // smallest terms first
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, w2, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f2, w0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, w1, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, w1, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, w0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, w0, acc, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {  // reference row of element e: (e&3) + 8*(e>>2) + 4*h, ascending in e
          const int rr = (e & 3) + 8 * (e >> 2) + 4 * h;
          const float s = (2.0f * acc[e] - qxx) - sXX[step & 1][trow + rr];  // -inf for rows past N
#if KF_MODE == 2
          lv[0] = fmaxf(lv[0], s);
#else
          // accepted candidates are parked in a BUF-deep per-lane shift register; the sorted insert (executed by the
          // whole wave whenever ANY lane needs it) runs once per BUF acceptances of the fastest-filling lane
          const bool accept = s > lv[KB - 1];
#pragma unroll
          for (int t = BUF - 1; t > 0; --t) {
            pv[t] = accept ? pv[t - 1] : pv[t];
            pj[t] = accept ? pj[t - 1] : pj[t];
          }
          pv[0] = accept ? s : pv[0];
          pj[0] = accept ? tile * 32 + rr : pj[0];
          np += accept ? 1 : 0;
#if KF_MODE == 0
          if (__ballot(np == BUF)) flush();
#else
          np = np == BUF ? 0 : np;
#endif
#endif
        }
      }
    }
    if (more) stash((step + 1) & 1);
    __syncthreads();
  }
  flush();
  // merge the four lists of every query (LDS: the reference images are dead)
  float *mv = sR;                                  // [128 queries][4 lists][KB]
  int *mi = reinterpret_cast<int *>(sR + 128 * 4 * KB);
  const int slot = ((qg * 32 + r) * 4 + 2 * half + h) * KB;
#pragma unroll
  for (int t = 0; t < KB; ++t) {
    mv[slot + t] = lv[t];
    mi[slot + t] = li[t];
  }
  __syncthreads();
  if (half == 0 && h == 0 && q < N) {
    const int s0 = (qg * 32 + r) * 4 * KB;
    int p[4] = {0, 0, 0, 0};
    int64_t *o = idx + ((size_t)b * N + q) * K;
    for (int t = 0; t < K; ++t) {
      float bv = -__builtin_inff();
      int bi = 0x7fffffff, bl = 0;
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        const float v = p[l] < KB ? mv[s0 + l * KB + p[l]] : -__builtin_inff();
        const int i = p[l] < KB ? mi[s0 + l * KB + p[l]] : 0x7fffffff;
        if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; bl = l; }
      }
      o[t] = sane_index_at(4, bi, N, t);
#pragma unroll
      for (int l = 0; l < 4; ++l) p[l] += bl == l ? 1 : 0;
    }
  }
}

// xx[row] = sum_k X[row,k]^2 in a fixed order: 16 lanes per row, each the ascending sum of its D / 16 values (float4 loads),
// then a four-step butterfly.  (torch spends a multiply pass and a reduction pass over the feature tensor on this,
// 22 us per DGCNN layer at B = 32 against the 49-72 us of the neighbour search itself.)
__global__ __launch_bounds__(256) void row_sqnorm_k(const float *__restrict__ X, long long rows, int D, float *__restrict__ xx) {
  const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int l = threadIdx.x & 15;
  float s = 0.f;
  if (row < rows) {
    const float4 *x4 = reinterpret_cast<const float4 *>(X + row * D);
    for (int q = l; q < D / 4; q += 16) {
      const float4 v = x4[q];
      s = fmaf(v.w, v.w, fmaf(v.z, v.z, fmaf(v.y, v.y, fmaf(v.x, v.x, s))));
    }
  }
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) s += __shfl_xor(s, m, HITADV_WAVE);
  if (row < rows && l == 0) xx[row] = s;
}

}  // namespace hitadv

extern "C" int hitadv_row_sqnorm(const float *X, int64_t rows, int D, float *xx, void *stream) {
  if (!X || !xx || rows <= 0 || D <= 0 || (D & 3) || ((uintptr_t)X & 15)) return HITADV_E_ARG;
  hitadv::row_sqnorm_k<<<(unsigned)((rows + 15) / 16), 256, 0, (hipStream_t)stream>>>(X, rows, D, xx);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_knn_features(const float *X, const float *xx, int B, int N, int D, int K, int64_t *idx,
                                   void *stream) {
  if (!X || !xx || !idx || B <= 0 || N <= 0 || K <= 0 || K > 20 || K > N || (D != 64 && D != 128) || ((uintptr_t)X & 15))
    return HITADV_E_ARG;
  dim3 grid((N + 127) / 128, B);
  hipStream_t s = (hipStream_t)stream;
#define HITADV_KNN_FEAT(DD, KK)                                                                                          \
  do {                                                                                                                 \
    constexpr int shm = hitadv::knn_feat_lds_bytes(DD, KK);                                                            \
    HITADV_RAISE_LDS((&hitadv::knn_feat_k<DD, KK>), shm);                                                              \
    hitadv::knn_feat_k<DD, KK><<<grid, 512, shm, s>>>(X, xx, N, K, idx);                                                \
  } while (0)
  // list length = K rounded up to an instantiated size: a shorter list means a tighter threshold and a cheaper insert
  if (D == 64) {
    if (K <= 5) HITADV_KNN_FEAT(64, 5);
    else if (K <= 8) HITADV_KNN_FEAT(64, 8);
    else HITADV_KNN_FEAT(64, 20);
  } else {
    if (K <= 5) HITADV_KNN_FEAT(128, 5);
    else if (K <= 8) HITADV_KNN_FEAT(128, 8);
    else HITADV_KNN_FEAT(128, 20);
  }
#undef HITADV_KNN_FEAT
  HITADV_LAUNCH_CHECK();
  return 0;
}
