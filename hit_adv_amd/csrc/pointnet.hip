// PointNet victim (model/feature_models.py:71-230 in eval mode, BatchNorm folded) as hand-written kernels:
// the per-point MLP stacks in front of the three 128->1024 layers, their input-gradient chains, and the
// small FC stacks, all on the f32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, k-ordered fmaf chain).
//
//   rowmlp_fwd<STAGE>   64 points of one cloud per block; the chain of shared layers runs tile-resident:
//                       layer output (accumulator layout: column on the lane) -> LDS -> next layer's A operand.
//        STAGE 0 (STN3d)   x -> relu(s1) -> relu(s2)                                 feature_models.py:168-171
//        STAGE 1 (encoder) x @ T3 -> relu(e1) -> relu(t1) -> relu(t2)                :119-128, :210-213
//        STAGE 2 (encoder) h1 @ T64 -> relu(e2)                                      :131-137
//   rowmlp_bwd<STAGE>   the same tiles backwards (input gradient only; weights are constants of the attack):
//                       ReLU masks from the saved activations, per-cloud transform gradients as per-tile
//                       partials (summed in tile order by sum_partials -> deterministic).
//   fc_layer            out[B,NOUT] = act(in[B,K] @ Wt[K,NOUT] + bias), 32 clouds per MFMA row block, K split
//                       over the waves of a block, partial tiles added in wave order.  With `mask` the input is
//                       gated by (mask > 0): the same kernel is the backward of a ReLU'd FC layer.
//
// Weights: Wt = [Cin][Cout] (forward operand), Wr = [Cout][Cin] (backward operand); both are kept by the caller.
#include <stdlib.h>

#include "common.hpp"
#include "deform_body.hpp"
#include "hitadv.h"

namespace hitadv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int PM_TM = 64;    // points per block tile
constexpr int PM_L64 = 68;   // LDS row stride of a 64-wide tile (conflict-free ds_read_b128)
constexpr int PM_L128 = 132;

// K index consumed by MFMA step t on lane half h: four consecutive steps of a lane are one float4 of A.
__device__ __forceinline__ int kmap(int t, int h) { return 8 * (t >> 2) + 4 * h + (t & 3); }

__device__ __forceinline__ void zero(f32x16 &a) {
#pragma unroll
  for (int e = 0; e < 16; ++e) a[e] = 0.f;
}

// B operand of a [.,K] x [K,32] product, held in registers: element (k, c) is TRANSB ? W[c*ldw + k] : W[k*ldw + c].
// Loaded at the top of a kernel so that its latency overlaps the tile loads.
template <int K, bool TRANSB>
__device__ __forceinline__ void load_w(const float *__restrict__ W, int ldw, int col, int r, int h, float (&w)[K / 2]) {
#pragma unroll
  for (int t = 0; t < K / 2; ++t)
    w[t] = TRANSB ? W[(size_t)(col + r) * ldw + kmap(t, h)] : W[(size_t)kmap(t, h) * ldw + col + r];
}

// acc[rb] += A[row_base + 32*rb .. +32)[0..K) @ Bm   (A: LDS tile, row stride lda floats; Bm: load_w registers)
template <int K, int NRB>
__device__ __forceinline__ void mfma_apply(const float *sA, int lda, int row_base, const float (&w)[K / 2],
                                           f32x16 (&acc)[NRB], int r, int h) {
#pragma unroll
  for (int j = 0; j < K / 8; ++j) {
    float av[NRB][4];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      const float4 a = *reinterpret_cast<const float4 *>(sA + (row_base + 32 * rb + r) * lda + 8 * j + 4 * h);
      av[rb][0] = a.x; av[rb][1] = a.y; av[rb][2] = a.z; av[rb][3] = a.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb)
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[rb][i], w[4 * j + i], acc[rb], 0, 0, 0);
  }
}

// accumulator element e of a 32x32 tile sits at row (e&3) + 8*(e>>2) + 4*h, column r
__device__ __forceinline__ int acc_row(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

// Coalesced float4 fetch of a [rows x W] tile of a points-major matrix into registers (rows past `rows` -> 0) ...
template <int W>
__device__ __forceinline__ void fetch_tile(const float *__restrict__ src, int rows, float4 (&v)[PM_TM * W / 4 / 256]) {
#pragma unroll
  for (int u = 0; u < PM_TM * W / 4 / 256; ++u) {
    const int e = threadIdx.x + 256 * u;
    const int n = e / (W / 4), c4 = e % (W / 4);
    v[u] = n < rows ? *reinterpret_cast<const float4 *>(src + (size_t)n * W + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
// ... and its store into an LDS tile of row stride ld.
template <int W>
__device__ __forceinline__ void stash_tile(const float4 (&v)[PM_TM * W / 4 / 256], float *dst, int ld) {
#pragma unroll
  for (int u = 0; u < PM_TM * W / 4 / 256; ++u) {
    const int e = threadIdx.x + 256 * u;
    *reinterpret_cast<float4 *>(dst + (e / (W / 4)) * ld + 4 * (e % (W / 4))) = v[u];
  }
}

struct RowMlpFwd {
  const float *x;    // [B,3,N]      (stage 0, 1)
  const float *T;    // [B,9] (stage 1) / [B,64,64] (stage 2)
  const float *hin;  // [B*N,64]     (stage 2: h1)
  const float *W0, *b0;  // 3->64   Wt [3,64]     (stage 0: s1, stage 1: e1)
  const float *W1, *b1;  // 64->64  Wt [64,64]    (stage 1: t1)
  const float *W2, *b2;  // 64->128 Wt [64,128]   (s2 / t2 / e2)
  float *xp;         // [B*N,3]   transformed points         (stage 1)
  float *o0;         // [B*N,64]  a1 (stage 0) / h1 (stage 1) / h1 @ T64 (stage 2)
  float *o1;         // [B*N,64]  relu(t1)                   (stage 1)
  float *o2;         // [B*N,128] relu(last layer)
  int N;
  // stage 1, optional: the input transform itself, T3[b] = F5[b,:256] @ W6[256,9] + b6 (STN3d's fc3, :186-190, identity
  // folded into b6), evaluated by every block of the cloud instead of by a launch of its own; tile 0 writes it to Tout
  const float *F5, *W6, *b6;
  float *Tout;
  // stage 0, optional: the input itself is HiT-ADV's deformation of d_ori (csrc/deform_body.hpp), evaluated by this block
  // for its 64 points instead of by a launch of its own; the deformed points go to d_adv [B,3,N] (what `x` would have held)
  // and 1 / sum_j k to d_inv, for the backward pass
  const float *d_ori, *d_central, *d_perturb, *d_sigma;
  float *d_adv, *d_inv;
  int d_C;
  // mode 2: o2 is written as PACKED PIECES, one 32-bit word per value = (fp16 hi | fp16 lo << 16) with lo the residual scaled by
  // 2^11 -- what the fp16x2 128 -> 1024 kernel consumes directly (hitadv_linear_max_fwd_f16x2_packed) and what the backward
  // chain (mode 2) reads as the layer's ReLU mask (word != 0); range_flag (may be NULL) is raised by a value beyond fp16's range
  int pack_o2;
  int *range_flag;
};

template <int STAGE>
__global__ __launch_bounds__(256) void rowmlp_fwd_k(RowMlpFwd a) {
  __shared__ float4 sA4[PM_TM * PM_L64 / 4], sB4[PM_TM * PM_L64 / 4];
  __shared__ float sX[PM_TM * 3], sXp[PM_TM * 3];
  float *sA = reinterpret_cast<float *>(sA4), *sB = reinterpret_cast<float *>(sB4);
  const int b = blockIdx.y, n0 = blockIdx.x * PM_TM, N = a.N;
  const int rows = min(PM_TM, N - n0);
  const size_t row0 = (size_t)b * N + n0;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int rb = wave & 1, cb = wave >> 1;

  // every global operand is requested up front; the layer chain below then only waits on LDS and the matrix cores
  float w2[32], w1[32];
  load_w<64, false>(a.W2, 128, 32 * wave, r, h, w2);
  const float bias2 = a.b2[32 * wave + r];
  float bias1 = 0.f;
  if (STAGE == 1) {
    load_w<64, false>(a.W1, 64, 32 * cb, r, h, w1);
    bias1 = a.b1[32 * cb + r];
  }
  if (STAGE == 2) load_w<64, false>(a.T + (size_t)b * 4096, 64, 32 * cb, r, h, w1);

  if (STAGE < 2) {
    const int c0 = threadIdx.x & 63;
    const float w00 = a.W0[c0], w01 = a.W0[64 + c0], w02 = a.W0[128 + c0], bb = a.b0[c0];
    if (STAGE == 0 && a.d_ori != nullptr) {  // block-uniform
      deform_fwd_body<256>(a.d_ori, a.d_central, a.d_perturb, a.d_sigma, N, a.d_C, a.d_adv, a.d_inv, b, blockIdx.x, sX);
    } else if (threadIdx.x < 192) {
      const int c = threadIdx.x >> 6, n = threadIdx.x & 63;
      sX[n * 3 + c] = n < rows ? a.x[((size_t)b * 3 + c) * N + n0 + n] : 0.f;
    }
    __shared__ float sTp[4][9], sT[9];
    const bool ownT = STAGE == 1 && a.F5 != nullptr;  // block-uniform
    if (ownT) {  // 256 threads = the 256 inputs of fc3: a fixed butterfly per wave, then the four waves in order
      const float f = a.F5[(size_t)b * 256 + threadIdx.x];
      float p[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) p[q] = f * a.W6[threadIdx.x * 9 + q];
#pragma unroll
      for (int m = 1; m < 64; m <<= 1)
#pragma unroll
        for (int q = 0; q < 9; ++q) p[q] += __shfl_xor(p[q], m, HITADV_WAVE);
      if (lane < 9) {
        float v = p[0];
#pragma unroll
        for (int q = 1; q < 9; ++q) v = lane == q ? p[q] : v;
        sTp[wave][lane] = v;
      }
    }
    __syncthreads();
    if (ownT) {
      if (threadIdx.x < 9) {
        const float v = ((sTp[0][threadIdx.x] + sTp[1][threadIdx.x]) + sTp[2][threadIdx.x]) + sTp[3][threadIdx.x] + a.b6[threadIdx.x];
        sT[threadIdx.x] = v;
        if (blockIdx.x == 0) a.Tout[(size_t)b * 9 + threadIdx.x] = v;
      }
      __syncthreads();
    }
    const float *xin = sX;
    if (STAGE == 1) {  // x' = x @ T3   (torch.bmm(x, trans), :124)
      if (threadIdx.x < 192) {
        const int n = threadIdx.x / 3, j = threadIdx.x % 3;
        const float *T = ownT ? sT : a.T + (size_t)b * 9;
        const float v = fmaf(sX[n * 3 + 2], T[6 + j], fmaf(sX[n * 3 + 1], T[3 + j], sX[n * 3] * T[j]));
        sXp[threadIdx.x] = v;
        if (a.xp != nullptr && n < rows) a.xp[row0 * 3 + threadIdx.x] = v;
      }
      __syncthreads();
      xin = sXp;
    }
    {  // 3 -> 64, ReLU: one column per lane, 16 rows per thread
      const int q = threadIdx.x >> 6;
#pragma unroll 4
      for (int i = 0; i < 16; ++i) {
        const int n = q * 16 + i;
        float v = fmaf(xin[n * 3 + 2], w02, fmaf(xin[n * 3 + 1], w01, fmaf(xin[n * 3], w00, bb)));
        v = v > 0.f ? v : 0.f;
        sA[n * PM_L64 + c0] = v;
        if (n < rows) a.o0[(row0 + n) * 64 + c0] = v;
      }
    }
    __syncthreads();
  } else {  // h1' = h1 @ T64   (torch.bmm(x, trans_feat), :131)
    float4 t[4];
    fetch_tile<64>(a.hin + row0 * 64, rows, t);
    stash_tile<64>(t, sB, PM_L64);
    __syncthreads();
    f32x16 acc[1];
    zero(acc[0]);
    mfma_apply<64, 1>(sB, PM_L64, 32 * rb, w1, acc, r, h);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int n = 32 * rb + acc_row(e, h), c = 32 * cb + r;
      sA[n * PM_L64 + c] = acc[0][e];
      if (a.o0 != nullptr && n < rows) a.o0[(row0 + n) * 64 + c] = acc[0][e];
    }
    __syncthreads();
  }

  const float *sIn = sA;
  if (STAGE == 1) {  // t1: 64 -> 64, ReLU
    f32x16 acc[1];
    zero(acc[0]);
    mfma_apply<64, 1>(sA, PM_L64, 32 * rb, w1, acc, r, h);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int n = 32 * rb + acc_row(e, h), c = 32 * cb + r;
      float v = acc[0][e] + bias1;
      v = v > 0.f ? v : 0.f;
      sB[n * PM_L64 + c] = v;
      if (n < rows) a.o1[(row0 + n) * 64 + c] = v;
    }
    __syncthreads();
    sIn = sB;
  }
  {  // 64 -> 128, ReLU: wave w owns columns 32w..32w+31 for all 64 rows
    f32x16 acc[2];
    zero(acc[0]);
    zero(acc[1]);
    mfma_apply<64, 2>(sIn, PM_L64, 0, w2, acc, r, h);
    const int c = 32 * wave + r;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = 32 * q + acc_row(e, h);
        float v = acc[q][e] + bias2;
        v = v > 0.f ? v : 0.f;
        if (n < rows) a.o2[(row0 + n) * 128 + c] = v;
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same chains on the fp16 matrix cores (mode 1 of the entry points; what `view.matrix_mode = 'fp16x2'` selects).
// v_mfma_f32_32x32x2_f32 takes 64 cycles for 4096 flop; v_mfma_f32_32x32x16_f16 32 cycles for 32768.  With every operand
// as TWO fp16 pieces (a = a1 + 2^-11 a2 + r, |r| <= 2^-24 |a|: csrc/victim_bf3.hip) a product costs three of the latter --
// 5.3x less matrix time -- at an error of fp32's own unit roundoff per term.  At B = 32 these kernels pay latency; in a
// 128-cloud stack (attack_many) half of their time was the f32 matrix pipe.  The C / D layout of the 32x32x16 form is that
// of the 32x32x2 form (column on the lane, acc_row(e, h) rows), so the epilogues are the same code; the A operand is read
// from LDS tiles of fp16 pieces (row strides 144 / 272 bytes: conflict-free ds_read_b128 for lane -> (row r, half h)),
// the B operand is split in registers once per block.
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
constexpr float PM_SC = F16X2_PIECE_SCALE;   // 2^11 (csrc/arith.hpp)
constexpr int PM_LH64 = 144;      // bytes per row of a 64-wide fp16 piece tile
constexpr int PM_LH128 = 272;

// The residual as ONE fused multiply-add with the fp16 piece as a half-precision source operand (v_fma_mix*): 2048 v is
// exact, hi (-2048) + 2048 v = 2048 (v - hi) is exact before its single rounding to fp16 -- the same bits as converting hi
// back, subtracting, scaling and converting (five instructions per value; the splits were a quarter of the backward kernels'
// vector instructions).
// (split_pair itself: csrc/arith.hpp -- plain C++, also compiled for the host by tests/test_arith_host.py)
// eight values: the hi pieces two per v_cvt_pk_f16_f32, the lo pieces written into the halves of their words by
// v_fma_mixlo / mixhi reading the packed hi pieces in place (2.5 instructions per value; the compiler's own form converts every
// hi piece twice, once alone for the residual and once packed)
__device__ __forceinline__ void split8v(const float (&v)[8], uint4 &hi, uint4 &lo) {
  uint32_t H[4], L[4];
  const float nsc = -PM_SC;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float s0 = v[2 * p] * PM_SC, s1 = v[2 * p + 1] * PM_SC;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(H[p]) : "v"(v[2 * p]), "v"(v[2 * p + 1]));
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(L[p]) : "v"(H[p]), "s"(nsc), "v"(s0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(L[p]) : "v"(H[p]), "s"(nsc), "v"(s1));
  }
  hi = make_uint4(H[0], H[1], H[2], H[3]);
  lo = make_uint4(L[0], L[1], L[2], L[3]);
}
// one value into the two piece tiles (row-major halves, row stride ld bytes)
__device__ __forceinline__ void put_pieces(char *th, char *tl, int ld, int n, int c, float v) {
  _Float16 x, y;
  split_pair(v, x, y);
  *reinterpret_cast<_Float16 *>(th + n * ld + 2 * c) = x;
  *reinterpret_cast<_Float16 *>(tl + n * ld + 2 * c) = y;
}

// B operand of a [.,K] x [K,32] product as pieces: step s, element i of lane (r, h) is W(k = 16 s + 8 h + i, col + r)
template <int K, bool TRANSB>
__device__ __forceinline__ void load_w16(const float *__restrict__ W, int ldw, int col, int r, int h, uint4 (&wh)[K / 16],
                                         uint4 (&wl)[K / 16]) {
  float v[K / 16][8];
#pragma unroll
  for (int s = 0; s < K / 16; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = 16 * s + 8 * h + i;
      v[s][i] = TRANSB ? W[(size_t)(col + r) * ldw + k] : W[(size_t)k * ldw + col + r];
    }
#pragma unroll
  for (int s = 0; s < K / 16; ++s) split8v(v[s], wh[s], wl[s]);
}

// the same fragments, one 16-deep slice at a time (8 values in flight instead of K / 2: for a reload inside a loop that holds
// other things in registers; its latency is paid once per cloud)
template <int K, bool TRANSB>
__device__ __forceinline__ void load_w16_seq(const float *__restrict__ W, int ldw, int col, int r, int h, uint4 (&wh)[K / 16],
                                             uint4 (&wl)[K / 16]) {
#pragma unroll
  for (int s = 0; s < K / 16; ++s) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = 16 * s + 8 * h + i;
      v[i] = TRANSB ? W[(size_t)(col + r) * ldw + k] : W[(size_t)k * ldw + col + r];
    }
    split8v(v, wh[s], wl[s]);
    asm volatile("" ::: "memory");  // (keeps the slices' loads from being hoisted together again)
  }
}

// acc[rb] += A[row_base + 32 rb .. +32)[0..K) @ B: three MFMAs per 16 values of k (lo x hi and hi x lo into accl, hi x hi into acc)
template <int K, int NRB>
__device__ __forceinline__ void mfma_apply16(const char *th, const char *tl, int ld, int row_base, const uint4 (&wh)[K / 16],
                                             const uint4 (&wl)[K / 16], f32x16 (&acc)[NRB], f32x16 (&accl)[NRB], int r, int h) {
#pragma unroll
  for (int s = 0; s < K / 16; ++s) {
    const h8v bh = __builtin_bit_cast(h8v, wh[s]), bl = __builtin_bit_cast(h8v, wl[s]);
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      const int o = (row_base + 32 * rb + r) * ld + 2 * (16 * s + 8 * h);
      const h8v ah = __builtin_bit_cast(h8v, *reinterpret_cast<const uint4 *>(th + o));
      const h8v al = __builtin_bit_cast(h8v, *reinterpret_cast<const uint4 *>(tl + o));
      accl[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accl[rb], 0, 0, 0);
      accl[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accl[rb], 0, 0, 0);
      acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[rb], 0, 0, 0);
    }
  }
}
__device__ __forceinline__ float joined(const f32x16 &hi, const f32x16 &lo, int e) { return fmaf(lo[e], 1.f / PM_SC, hi[e]); }

template <int STAGE>
__global__ __launch_bounds__(256) void rowmlp_fwd16_k(RowMlpFwd a) {
  __shared__ __attribute__((aligned(16))) char tA[2][PM_TM * PM_LH64], tB[2][PM_TM * PM_LH64];  // [piece][row][k]
  __shared__ float sX[PM_TM * 3], sXp[PM_TM * 3];
  const int b = blockIdx.y, n0 = blockIdx.x * PM_TM, N = a.N;
  const int rows = min(PM_TM, N - n0);
  const size_t row0 = (size_t)b * N + n0;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int rb = wave & 1, cb = wave >> 1;

  uint4 w2h[4], w2l[4], w1h[4], w1l[4];
  load_w16<64, false>(a.W2, 128, 32 * wave, r, h, w2h, w2l);
  const float bias2 = a.b2[32 * wave + r];
  float bias1 = 0.f;
  if (STAGE == 1) {
    load_w16<64, false>(a.W1, 64, 32 * cb, r, h, w1h, w1l);
    bias1 = a.b1[32 * cb + r];
  }
  if (STAGE == 2) load_w16<64, false>(a.T + (size_t)b * 4096, 64, 32 * cb, r, h, w1h, w1l);

  if (STAGE < 2) {
    const int c0 = threadIdx.x & 63;
    const float w00 = a.W0[c0], w01 = a.W0[64 + c0], w02 = a.W0[128 + c0], bb = a.b0[c0];
    if (STAGE == 0 && a.d_ori != nullptr) {  // block-uniform
      deform_fwd_body<256>(a.d_ori, a.d_central, a.d_perturb, a.d_sigma, N, a.d_C, a.d_adv, a.d_inv, b, blockIdx.x, sX);
    } else if (threadIdx.x < 192) {
      const int c = threadIdx.x >> 6, n = threadIdx.x & 63;
      sX[n * 3 + c] = n < rows ? a.x[((size_t)b * 3 + c) * N + n0 + n] : 0.f;
    }
    __shared__ float sTp[4][9], sT[9];
    const bool ownT = STAGE == 1 && a.F5 != nullptr;  // block-uniform
    if (ownT) {  // exactly the f32 kernel's evaluation of STN3d's last layer (same bits: T3 is shared with the backward pass)
      const float f = a.F5[(size_t)b * 256 + threadIdx.x];
      float p[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) p[q] = f * a.W6[threadIdx.x * 9 + q];
#pragma unroll
      for (int m = 1; m < 64; m <<= 1)
#pragma unroll
        for (int q = 0; q < 9; ++q) p[q] += __shfl_xor(p[q], m, HITADV_WAVE);
      if (lane < 9) {
        float v = p[0];
#pragma unroll
        for (int q = 1; q < 9; ++q) v = lane == q ? p[q] : v;
        sTp[wave][lane] = v;
      }
    }
    __syncthreads();
    if (ownT) {
      if (threadIdx.x < 9) {
        const float v = ((sTp[0][threadIdx.x] + sTp[1][threadIdx.x]) + sTp[2][threadIdx.x]) + sTp[3][threadIdx.x] + a.b6[threadIdx.x];
        sT[threadIdx.x] = v;
        if (blockIdx.x == 0) a.Tout[(size_t)b * 9 + threadIdx.x] = v;
      }
      __syncthreads();
    }
    // the cloud itself: a coordinate that is not finite would be mapped to 0 by the first ReLU below and never be seen again
    // (the reference propagates it into NaN logits): raise the range flag, the caller then refuses / re-runs the pass
    if (STAGE == 0 && a.range_flag != nullptr && threadIdx.x < 192) {
      RangeWatch xw;
      xw.see(sX[threadIdx.x]);
      if (xw.beyond_fp16()) *a.range_flag = 1;
    }
    const float *xin = sX;
    if (STAGE == 1) {  // x' = x @ T3   (torch.bmm(x, trans), :124)
      if (threadIdx.x < 192) {
        const int n = threadIdx.x / 3, j = threadIdx.x % 3;
        const float *T = ownT ? sT : a.T + (size_t)b * 9;
        const float v = fmaf(sX[n * 3 + 2], T[6 + j], fmaf(sX[n * 3 + 1], T[3 + j], sX[n * 3] * T[j]));
        sXp[threadIdx.x] = v;
        if (a.xp != nullptr && n < rows) a.xp[row0 * 3 + threadIdx.x] = v;
      }
      __syncthreads();
      xin = sXp;
    }
    {  // 3 -> 64, ReLU on the VALU (exact f32, as the f32 kernel): one column per lane, 16 rows per thread
      const int q = threadIdx.x >> 6;
#pragma unroll 4
      for (int i = 0; i < 16; ++i) {
        const int n = q * 16 + i;
        float v = fmaf(xin[n * 3 + 2], w02, fmaf(xin[n * 3 + 1], w01, fmaf(xin[n * 3], w00, bb)));
        v = v > 0.f ? v : 0.f;
        put_pieces(tA[0], tA[1], PM_LH64, n, c0, v);
        if (n < rows) a.o0[(row0 + n) * 64 + c0] = v;
      }
    }
    __syncthreads();
  } else {  // h1' = h1 @ T64   (torch.bmm(x, trans_feat), :131)
    float4 t[4];
    fetch_tile<64>(a.hin + row0 * 64, rows, t);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = threadIdx.x + 256 * u;
      const int n = e / 16, c = 4 * (e % 16);
      put_pieces(tB[0], tB[1], PM_LH64, n, c, t[u].x);
      put_pieces(tB[0], tB[1], PM_LH64, n, c + 1, t[u].y);
      put_pieces(tB[0], tB[1], PM_LH64, n, c + 2, t[u].z);
      put_pieces(tB[0], tB[1], PM_LH64, n, c + 3, t[u].w);
    }
    __syncthreads();
    f32x16 acc[1], accl[1];
    zero(acc[0]);
    zero(accl[0]);
    mfma_apply16<64, 1>(tB[0], tB[1], PM_LH64, 32 * rb, w1h, w1l, acc, accl, r, h);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int n = 32 * rb + acc_row(e, h), c = 32 * cb + r;
      float v = joined(acc[0], accl[0], e);
      // the pieces are the split of the fp32 value that is STORED: without this the compiler folds the fma above and the
      // conversion to fp16 into one v_fma_mixlo_f16 (a single rounding of the exact fma: another hi piece in 1 of ~8000
      // values; found in round 5 when rowmlp_stream_k, which splits the rounded value, differed in 0.4 % of the rows)
      asm volatile("" : "+v"(v));
      put_pieces(tA[0], tA[1], PM_LH64, n, c, v);
      if (a.o0 != nullptr && n < rows) a.o0[(row0 + n) * 64 + c] = v;
    }
    __syncthreads();
  }

  const char *inH = tA[0], *inL = tA[1];
  if (STAGE == 1) {  // t1: 64 -> 64, ReLU
    f32x16 acc[1], accl[1];
    zero(acc[0]);
    zero(accl[0]);
    mfma_apply16<64, 1>(tA[0], tA[1], PM_LH64, 32 * rb, w1h, w1l, acc, accl, r, h);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int n = 32 * rb + acc_row(e, h), c = 32 * cb + r;
      float v = joined(acc[0], accl[0], e) + bias1;
      v = v > 0.f ? v : 0.f;
      put_pieces(tB[0], tB[1], PM_LH64, n, c, v);
      if (n < rows) a.o1[(row0 + n) * 64 + c] = v;
    }
    __syncthreads();
    inH = tB[0];
    inL = tB[1];
  }
  {  // 64 -> 128, ReLU: wave w owns columns 32w..32w+31 for all 64 rows, one 32-row block after the other (two accumulator
     // sets per block: with both blocks in flight the kernel needs 173 registers and loses its third wave per SIMD)
    const int c = 32 * wave + r;
    RangeWatch big;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x16 acc[1], accl[1];
      zero(acc[0]);
      zero(accl[0]);
      mfma_apply16<64, 1>(inH, inL, PM_LH64, 32 * q, w2h, w2l, acc, accl, r, h);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = 32 * q + acc_row(e, h);
        const float pre = joined(acc[0], accl[0], e) + bias2;
        float v = pre > 0.f ? pre : 0.f;
        if (a.pack_o2) {  // block-uniform
          _Float16 x, y;
          split_pair(v, x, y);
          big.see_relu(pre, v);
          const uint32_t wd = (uint32_t)__builtin_bit_cast(uint16_t, x) | ((uint32_t)__builtin_bit_cast(uint16_t, y) << 16);
          if (n < rows) reinterpret_cast<uint32_t *>(a.o2)[(row0 + n) * 128 + c] = wd;
        } else if (n < rows) {
          a.o2[(row0 + n) * 128 + c] = v;
        }
      }
    }
    if (a.pack_o2 && a.range_flag != nullptr && big.beyond_fp16()) *a.range_flag = 1;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// rowmlp_stream_k (round 5): the SAME layers, the same arithmetic per element -- every value it writes is the bit pattern
// rowmlp_fwd16_k writes (tests/test_gpu_kernels.py::test_rowmlp_stream_equals_the_tile_kernel) -- with the data movement of
// csrc/rows_linear.hip.  What the counters said about rowmlp_fwd16_k at the stacked launch size (256 clouds,
// profiles/r05_loop_traffic.json): 45.5 / 26.7 / 20.1 M vector instructions per launch = 100 % / 65 % / 72 % of the kernel's
// duration at four cycles each on the chip's 1024 SIMDs -- it is bound by its VECTOR INSTRUCTIONS, not by its stores
// (2.3-3.7 TB/s) and not by latency: ~1600 per wave and tile, of which a quarter re-loads and re-splits the block's weights
// for every 64 points, and most of the rest is per-VALUE work: one fp16 conversion pair, two 2-byte LDS stores, one
// 4-byte global store with its own 64-bit address for every value a layer produces.  Here:
//   * a workgroup streams `tpb` consecutive 64-point tiles: the weights' pieces are loaded and split ONCE per workgroup
//     (per cloud for the per-cloud 64 x 64 transform);
//   * the products are taken TRANSPOSED (A = the weights, B = the activation: D[channel][point]; the same products summed in
//     the same order over k, so the same bits: tools/tune/mfma_transpose_probe.hip): a lane's accumulator then holds runs of
//     four consecutive CHANNELS of one point -- pieces are made two per v_cvt_pk_f16_f32 and leave as 8-byte LDS stores,
//     values as 16-byte ones;
//   * every result goes to an LDS tile first and leaves as WHOLE ROWS, 16 bytes per lane (MI355X_MICROARCH.md, stores);
//   * the 3 -> 64 layer keeps its exact-f32 chain, sixteen consecutive channels per thread with the wave's slice of W in SGPRs.
// LDS (two workgroups per CU): the piece images P1 (first 64-wide activation) and P2 (second, stages 1 / 2), fp32 tiles O64
// (/ O64b, stage 1) and the 128-wide result tile O128, which ALIASES what is dead by the time it is written:
//   stage 0:  P1 | [O64 ......... O128]                      52 KB   (O64 has left before the barrier in front of the epilogue;
//                                                                     the deformation's own LDS is in there too, between tiles)
//   stage 1:  [P1 | O64 = O128] | P2 | O64b                  72 KB   (P1 and O64 are dead after the middle layer's barrier)
//   stage 2:  P1 | P2 | [O64 ......... O128]                 71 KB
constexpr int RS_P = PM_TM * PM_LH64;          // bytes of one 64-wide piece image (64 rows x 144 B)
constexpr int RS_O64 = PM_TM * PM_L64 * 4;     // bytes of an fp32 64 x 64 tile (row stride 68 floats)
constexpr int RS_O128 = PM_TM * PM_L128 * 4;   // bytes of a 64 x 128 tile of 32-bit words (row stride 132)
static_assert(2 * RS_P + RS_O64 >= RS_O128, "stage 1: the 128-wide tile aliases P1 + O64");
static_assert(deform_fwd_lds_float4<256>() * 16 <= RS_O128, "stage 0: the deformation's LDS is the result tile's");
template <int STAGE>
__host__ __device__ constexpr int rowmlp_stream_lds() {
  return STAGE == 0 ? 2 * RS_P + RS_O128 : (STAGE == 1 ? 4 * RS_P + 2 * RS_O64 : 4 * RS_P + RS_O128);
}

// D[32 channels of this wave's block][32 points of block pb] += W^T x^T over k = 0..63: weights as the A operand (registers),
// the activation's piece rows as the B operand.  Per 16 values of k the three MFMAs of mfma_apply16, in its order.
__device__ __forceinline__ void mfma_cols16(const char *th, const char *tl, int pb, const uint4 (&wh)[4], const uint4 (&wl)[4],
                                            f32x16 &acc, f32x16 &accl, int r, int h) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const h8v bh = __builtin_bit_cast(h8v, wh[s]), bl = __builtin_bit_cast(h8v, wl[s]);
    const int o = (32 * pb + r) * PM_LH64 + 2 * (16 * s + 8 * h);
    const h8v ah = __builtin_bit_cast(h8v, *reinterpret_cast<const uint4 *>(th + o));
    const h8v al = __builtin_bit_cast(h8v, *reinterpret_cast<const uint4 *>(tl + o));
    accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, accl, 0, 0, 0);
    accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, accl, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc, 0, 0, 0);
  }
}

// four values -> their fp16 pieces, two words each (hi pieces two per conversion, lo pieces by v_fma_mix: split8v's bits,
// which are split_pair's: tools/tune/split_probe.hip)
__device__ __forceinline__ void split4v(const float (&v)[4], uint2 &hi, uint2 &lo) {
  uint32_t H[2], L[2];
  const float nsc = -PM_SC;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float s0 = v[2 * p] * PM_SC, s1 = v[2 * p + 1] * PM_SC;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(H[p]) : "v"(v[2 * p]), "v"(v[2 * p + 1]));
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(L[p]) : "v"(H[p]), "s"(nsc), "v"(s0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(L[p]) : "v"(H[p]), "s"(nsc), "v"(s1));
  }
  hi = make_uint2(H[0], H[1]);
  lo = make_uint2(L[0], L[1]);
}

template <int STAGE>
__global__ __launch_bounds__(256, 2) void rowmlp_stream_k(RowMlpFwd a, int tiles_per_cloud, int total_tiles, int tpb) {
  extern __shared__ __attribute__((aligned(16))) char sRS[];
  char *const P1h = sRS, *const P1l = sRS + RS_P;
  char *const P2h = sRS + (STAGE == 1 ? 2 * RS_P + RS_O64 : 2 * RS_P), *const P2l = P2h + RS_P;  // (stages 1, 2)
  float *const O64 = reinterpret_cast<float *>(sRS + (STAGE == 2 ? 4 * RS_P : 2 * RS_P));
  float *const O64b = reinterpret_cast<float *>(sRS + 4 * RS_P + RS_O64);  // STAGE 1 only
  uint32_t *const O128 = STAGE == 1 ? reinterpret_cast<uint32_t *>(sRS) : reinterpret_cast<uint32_t *>(O64);
  __shared__ float sX[PM_TM * 3], sXp[PM_TM * 3];
  __shared__ float sTp[4][9], sT[9];
  __shared__ __attribute__((aligned(16))) float sB2[128], sB1[64];
  const int N = a.N;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int t_begin = blockIdx.x * tpb, t_end = min(total_tiles, t_begin + tpb);
  if (t_begin >= t_end) return;

  // the wave's weights, once: the last layer's 32 channels 32 wave .. +31, the middle layer's 32 (wave & 1) .. +31
  uint4 w2h[4], w2l[4], w1h[4], w1l[4];
  load_w16<64, false>(a.W2, 128, 32 * wave, r, h, w2h, w2l);
  if (threadIdx.x < 128) sB2[threadIdx.x] = a.b2[threadIdx.x];
  if (STAGE == 1 && threadIdx.x < 64) sB1[threadIdx.x] = a.b1[threadIdx.x];
  const int cb = wave & 1, pbm = wave >> 1;  // middle layer: channel block, point block
  if (STAGE == 1) load_w16<64, false>(a.W1, 64, 32 * cb, r, h, w1h, w1l);
  // 3 -> 64: this wave's sixteen channels 16 wave .. +15 (wave-uniform addresses: the compiler keeps them in SGPRs)
  float w0[3][16], bb0[16];
  if (STAGE < 2) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      w0[0][c] = a.W0[16 * wave + c];
      w0[1][c] = a.W0[64 + 16 * wave + c];
      w0[2][c] = a.W0[128 + 16 * wave + c];
      bb0[c] = a.b0[16 * wave + c];
    }
  }
  RangeWatch big;
  int cloud = -1;
  float4 hin_next[4];
  auto fetch_hin = [&](int t) {  // STAGE 2: the next tile's 64 x 64 rows, requested a tile ahead
    const int b = t / tiles_per_cloud, n0 = (t % tiles_per_cloud) * PM_TM;
    const int rows = min(PM_TM, N - n0);
    fetch_tile<64>(a.hin + ((size_t)b * N + n0) * 64, rows, hin_next);
  };
  if (STAGE == 2) fetch_hin(t_begin);

  for (int t = t_begin; t < t_end; ++t) {
    const int b = t / tiles_per_cloud, tile = t % tiles_per_cloud, n0 = tile * PM_TM;
    const int rows = min(PM_TM, N - n0);
    const size_t row0 = (size_t)b * N + n0;
    const bool new_cloud = b != cloud;  // block-uniform
    cloud = b;
    // ------------------------------------------------------------------ phase 0: the tile's input
    if (STAGE < 2) {
      if (STAGE == 0 && a.d_ori != nullptr) {
        // (its LDS is the result tile's: dead between the tiles)
        deform_fwd_body_in<256>(reinterpret_cast<float4 *>(O128), a.d_ori, a.d_central, a.d_perturb, a.d_sigma, N, a.d_C, a.d_adv, a.d_inv, b,
                                tile, sX);
      } else if (threadIdx.x < 192) {
        const int c = threadIdx.x >> 6, n = threadIdx.x & 63;
        sX[n * 3 + c] = n < rows ? a.x[((size_t)b * 3 + c) * N + n0 + n] : 0.f;
      }
      const bool ownT = STAGE == 1 && a.F5 != nullptr;  // block-uniform
      if (ownT && new_cloud) {  // STN3d's last layer for this cloud: rowmlp_fwd16_k's evaluation, once per cloud and workgroup
        const float f = a.F5[(size_t)b * 256 + threadIdx.x];
        float p[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) p[q] = f * a.W6[threadIdx.x * 9 + q];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1)
#pragma unroll
          for (int q = 0; q < 9; ++q) p[q] += __shfl_xor(p[q], m, HITADV_WAVE);
        if (lane < 9) {
          float v = p[0];
#pragma unroll
          for (int q = 1; q < 9; ++q) v = lane == q ? p[q] : v;
          sTp[wave][lane] = v;
        }
      }
      __syncthreads();
      if (ownT && new_cloud) {
        if (threadIdx.x < 9) {
          const float v = ((sTp[0][threadIdx.x] + sTp[1][threadIdx.x]) + sTp[2][threadIdx.x]) + sTp[3][threadIdx.x] + a.b6[threadIdx.x];
          sT[threadIdx.x] = v;
          if (tile == 0) a.Tout[(size_t)b * 9 + threadIdx.x] = v;
        }
        __syncthreads();
      }
      if (STAGE == 0 && a.range_flag != nullptr && threadIdx.x < 192) {  // a cloud that is not finite (see rowmlp_fwd16_k)
        RangeWatch xw;
        xw.see(sX[threadIdx.x]);
        if (xw.beyond_fp16()) *a.range_flag = 1;
      }
      const float *xin = sX;
      if (STAGE == 1) {  // x' = x @ T3
        if (threadIdx.x < 192) {
          const int n = threadIdx.x / 3, j = threadIdx.x % 3;
          const float *T = ownT ? sT : a.T + (size_t)b * 9;
          const float v = fmaf(sX[n * 3 + 2], T[6 + j], fmaf(sX[n * 3 + 1], T[3 + j], sX[n * 3] * T[j]));
          sXp[threadIdx.x] = v;
          if (a.xp != nullptr && n < rows) a.xp[row0 * 3 + threadIdx.x] = v;
        }
        __syncthreads();
        xin = sXp;
      }
      {  // 3 -> 64, ReLU: point `lane`, channels 16 wave .. +15
        const float x0 = xin[lane * 3], x1 = xin[lane * 3 + 1], x2 = xin[lane * 3 + 2];
        float v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const float y = fmaf(x2, w0[2][c], fmaf(x1, w0[1][c], fmaf(x0, w0[0][c], bb0[c])));
          v[c] = y > 0.f ? y : 0.f;
        }
        float *so = O64 + lane * PM_L64 + 16 * wave;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4 *>(so + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float v8[8] = {v[8 * q], v[8 * q + 1], v[8 * q + 2], v[8 * q + 3], v[8 * q + 4], v[8 * q + 5], v[8 * q + 6], v[8 * q + 7]};
          uint4 hi, lo;
          split8v(v8, hi, lo);
          *reinterpret_cast<uint4 *>(P1h + lane * PM_LH64 + 2 * (16 * wave + 8 * q)) = hi;
          *reinterpret_cast<uint4 *>(P1l + lane * PM_LH64 + 2 * (16 * wave + 8 * q)) = lo;
        }
      }
    } else {  // STAGE 2: h1's rows (in registers since the last tile) -> pieces; the cloud's 64 x 64 transform as weights
      if (new_cloud) load_w16_seq<64, false>(a.T + (size_t)b * 4096, 64, 32 * cb, r, h, w1h, w1l);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = threadIdx.x + 256 * u;
        const int n = e / 16, c = 4 * (e % 16);
        const float v4[4] = {hin_next[u].x, hin_next[u].y, hin_next[u].z, hin_next[u].w};
        uint2 hi, lo;
        split4v(v4, hi, lo);
        *reinterpret_cast<uint2 *>(P1h + n * PM_LH64 + 2 * c) = hi;
        *reinterpret_cast<uint2 *>(P1l + n * PM_LH64 + 2 * c) = lo;
      }
      if (t + 1 < t_end) fetch_hin(t + 1);
    }
    __syncthreads();
    // ------------------------------------------------------------------ phase 1: the first activation leaves; the middle layer
    auto leave64 = [&](const float *src, float *dst) {  // an fp32 64 x 64 tile -> global, whole rows
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = threadIdx.x + 256 * u;
        const int n = e >> 4, c4 = e & 15;
        const float4 v = *reinterpret_cast<const float4 *>(src + n * PM_L64 + 4 * c4);
        if (n < rows) *reinterpret_cast<float4 *>(dst + (row0 + n) * 64 + 4 * c4) = v;
      }
    };
    if (STAGE < 2) leave64(O64, a.o0);
    const char *inH = P1h, *inL = P1l;
    if (STAGE >= 1) {  // STAGE 1: t1 = relu(W1 e1 + b1); STAGE 2: h1' = h1 @ T64 (no bias, no ReLU)
      f32x16 acc, accl;
      zero(acc);
      zero(accl);
      mfma_cols16(P1h, P1l, pbm, w1h, w1l, acc, accl, r, h);
      const int n = 32 * pbm + r;
      float *so = (STAGE == 1 ? O64b : O64) + n * PM_L64 + 32 * cb + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v4[4];
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (STAGE == 1) bv = *reinterpret_cast<const float4 *>(sB1 + 32 * cb + 8 * g + 4 * h);
        const float bq[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = joined(acc, accl, 4 * g + i);
          if (STAGE == 1) {
            v = v + bq[i];
            v = v > 0.f ? v : 0.f;
          }
          v4[i] = v;
        }
        *reinterpret_cast<float4 *>(so + 8 * g) = make_float4(v4[0], v4[1], v4[2], v4[3]);
        uint2 hi, lo;
        split4v(v4, hi, lo);
        *reinterpret_cast<uint2 *>(P2h + n * PM_LH64 + 2 * (32 * cb + 8 * g + 4 * h)) = hi;
        *reinterpret_cast<uint2 *>(P2l + n * PM_LH64 + 2 * (32 * cb + 8 * g + 4 * h)) = lo;
      }
      __syncthreads();
      if (STAGE == 1) leave64(O64b, a.o1);
      else if (a.o0 != nullptr) leave64(O64, a.o0);
      inH = P2h;
      inL = P2l;
    }
    // ------------------------------------------------------------------ phase 2: 64 -> 128, ReLU, one 32-point block at a time
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x16 acc2, accl2;
      zero(acc2);
      zero(accl2);
      mfma_cols16(inH, inL, q, w2h, w2l, acc2, accl2, r, h);
      // stages 0 / 2: O64 (under the result tile) has been read by everybody (stage 1: its barrier is the middle layer's)
      if (STAGE != 1 && q == 0) __syncthreads();
      const int n = 32 * q + r;
      uint32_t *so = O128 + n * PM_L128 + 32 * wave + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v4[4];
        const float4 bv = *reinterpret_cast<const float4 *>(sB2 + 32 * wave + 8 * g + 4 * h);
        const float bq[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float pre = joined(acc2, accl2, 4 * g + i) + bq[i];
          v4[i] = pre > 0.f ? pre : 0.f;
          if (a.pack_o2) big.see_relu(pre, v4[i]);
        }
        uint4 w;
        if (a.pack_o2) {  // block-uniform: one word per value, fp16 hi | fp16 lo << 16
          uint2 hi, lo;
          split4v(v4, hi, lo);
          w.x = (hi.x & 0xffffu) | (lo.x << 16);
          w.y = (hi.x >> 16) | (lo.x & 0xffff0000u);
          w.z = (hi.y & 0xffffu) | (lo.y << 16);
          w.w = (hi.y >> 16) | (lo.y & 0xffff0000u);
        } else {
          w = make_uint4(__float_as_uint(v4[0]), __float_as_uint(v4[1]), __float_as_uint(v4[2]), __float_as_uint(v4[3]));
        }
        *reinterpret_cast<uint4 *>(so + 8 * g) = w;
      }
    }
    __syncthreads();
    {  // the 64 x 128 tile leaves, 512 bytes per row
      uint32_t *dst = reinterpret_cast<uint32_t *>(a.o2);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = threadIdx.x + 256 * u;
        const int n = e >> 5, c4 = e & 31;
        const uint4 v = *reinterpret_cast<const uint4 *>(O128 + n * PM_L128 + 4 * c4);
        if (n < rows) *reinterpret_cast<uint4 *>(dst + (row0 + n) * 128 + 4 * c4) = v;
      }
    }
    __syncthreads();  // the tile's LDS is free again
  }
  if (a.pack_o2 && a.range_flag != nullptr && big.beyond_fp16()) *a.range_flag = 1;
}

// dA2[D,32 columns of this wave] = S[D,M] @ W3r[list,:]: the gather of the max-pool gradient as MFMAs.  S[i,k] = g_k if
// list entry k routes to the i-th winning point of the tile (one non-zero per column, built on the fly); the list is
// padded with zero-gradient entries to a multiple of 32, so the loop body is branch-free.  TWO: more than 32 winning
// points -> a second row block.
template <bool TWO>
__device__ __forceinline__ void gather_rows(const int2 *list, int M, const float *__restrict__ Wc, int r, int h,
                                            f32x16 (&acc)[2]) {
  if (M <= 0) return;
  // Two register sets that swap roles every batch of 32 list entries: while one batch's 16 MFMAs run, the W3r rows of
  // the next are in flight into the other set.  Batches go in PAIRS and every batch requests its successor
  // unconditionally (the list carries 128 zero-gradient padding entries: they add exact zeros): a request under a
  // condition, or a register copy from a "next" set into a "current" one, makes the compiler wait for the rows it has
  // just requested before the MFMAs that do not need them.
  int2 enA[16], enB[16];
  float bvA[16], bvB[16];
  auto request = [&](int2 (&en)[16], float (&bv)[16], int q) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      en[t] = list[q + 2 * t + h];
      bv[t] = Wc[(size_t)(en[t].x & 0xffff) * 128];
    }
  };
  auto multiply = [&](const int2 (&en)[16], const float (&bv)[16]) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int i = en[t].x >> 16;
      const float g = __int_as_float(en[t].y);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(i == r ? g : 0.f, bv[t], acc[0], 0, 0, 0);
      if (TWO) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(i == 32 + r ? g : 0.f, bv[t], acc[1], 0, 0, 0);
    }
  };
  request(enA, bvA, 0);
  for (int q = 0; q < M; q += 64) {
    request(enB, bvB, q + 32);
    multiply(enA, bvA);
    request(enA, bvA, q + 64);
    multiply(enB, bvB);
  }
}

// Backward of a stage, starting from the gradient at the max-pooled output of its 128->Cout layer: the block
// first GATHERS its 64 points' rows of  dA2[n,:] = sum_{j : argmax[b,j] == n} dg[b,j] * W3r[j,:]  straight into the
// LDS tile (the max routes each channel's gradient to one point; channels are taken in ascending order -> no
// atomics, bitwise reproducible), then runs the chain.
struct RowMlpBwd {
  const float *dg;     // [B,Cout]  gradient at the max-pooled (and ReLU'd, if gmask) output
  const float *gmask;  // [B,Cout]  the ReLU'd forward output (gradient passes where > 0) or NULL
  const int64_t *idx;  // [B,Cout]  arg-max point of every channel
  const float *W3r;    // [Cout,128]
  const float *A2;    // [B*N,128] saved relu(64->128) output (mask)
  const float *W2r;   // [128,64]
  const float *A1;    // [B*N,64]  stage 0: a1s, stage 1: relu(t1) (mask)
  const float *W1r;   // [64,64]   stage 1: t1
  const float *H1;    // [B*N,64]  stage 1: mask of e1's ReLU; stage 2: left operand of dT64
  const float *dH1in; // [B*N,64]  stage 1: gradient arriving at h1 from stage 2
  const float *W0r;   // [64,3]    stage 0: s1, stage 1: e1
  const float *T;     // stage 1: [B,9], stage 2: [B,64,64]
  const float *x;     // [B,3,N]   stage 1 (left operand of dT3)
  const float *dPin;  // [B,3,N]   stage 0: gradient arriving at the points from stage 1
  float *dTpart;      // stage 1: [B,tiles,9]; stage 2: [B,tiles,64,64]
  float *out;         // stage 0: dX [B,3,N]; stage 1: dPts [B,3,N]; stage 2: dH1 [B*N,64] (ONLY the rows in pres_out)
  const unsigned long long *pres_in;  // [B,tiles] bit n: row n of the incoming gradient (dH1in / dPin) is non-zero;
                                      // NULL: every row may be (stage 2 has no incoming gradient: NULL = none)
  unsigned long long *pres_out;       // [B,tiles] rows of this tile that receive any gradient in this stage, or NULL
  int N, Cout;
  int a2_packed;                      // mode 2: A2 holds packed fp16 pieces (rowmlp_fwd16_k, pack_o2), not fp32 values
  int *overflow;                      // [B,tiles] (two-word tiles only) 1: the tile has more than 64 winning points, see FIXUP
  const int *fix_parent;              // one-word launch as the SECOND launch of the two-word form (round 5): [B,tiles/2] = the first
                                      // launch's overflow table; a block whose 128-point parent tile is not marked leaves at once
};

constexpr int BW_CH = 4;  // Cout <= 256 * BW_CH
#ifndef BW_MFMA_GATHER_DEFAULT
#define BW_MFMA_GATHER_DEFAULT true
#endif
constexpr bool BW_MFMA_GATHER = BW_MFMA_GATHER_DEFAULT;  // see the gather step of rowmlp_bwd_k

// Diagnostic build only (-DHITADV_STAMPS, tools/v3_phases.py): wall-clock stamps (s_memrealtime, 10 ns) of the phases of
// every block, read back through hitadv_debug_v3_stamps.  The product build has no stamp and no such symbol.
#ifdef HITADV_STAMPS
__device__ unsigned long long g_v3_stamp[3 * 1024 * 8];
// stamps stay in registers until the block is done: a global store in front of a barrier would make the barrier wait for
// every load in flight (release semantics) and change what is being measured
#define V3_STAMP_DECL unsigned long long v3_st[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define V3_STAMP(i) v3_st[i] = __builtin_amdgcn_s_memrealtime()
#define V3_STAMP_FLUSH()                                                                                      \
  do {                                                                                                        \
    if (threadIdx.x == 0 && blockIdx.y * gridDim.x + blockIdx.x < 1024)                                       \
      for (int i_ = 0; i_ < 8; ++i_) g_v3_stamp[(STAGE * 1024 + blockIdx.y * gridDim.x + blockIdx.x) * 8 + i_] = v3_st[i_]; \
  } while (0)
#else
#define V3_STAMP_DECL
#define V3_STAMP(i)
#define V3_STAMP_FLUSH()
#endif

// Registers: stage 1 holds the most masks and incoming gradients at once; at three blocks per CU (168 registers) it spills,
// and a spill is a scratch access behind the same counter as the loads it was meant to overlap.  Two blocks per CU is what
// a 16-tile x 32-cloud grid puts on a CU anyway.
template <int STAGE>
__global__ __launch_bounds__(256, STAGE == 1 ? 2 : 3) void rowmlp_bwd_k(RowMlpBwd a) {
  // LDS footprint kept at 51 KB (sD + sE) so that blocks of other kernels fit beside two of these on a CU: the second
  // 64-wide tile (sF) reuses sD, which is dead once every wave has finished the 128-deep product
  __shared__ float4 sD4[PM_TM * PM_L128 / 4], sEF4[PM_TM * PM_L64 / 4];
  __shared__ float sX[PM_TM * 3], sG[PM_TM * 3];
  __shared__ int s_cnt[BW_CH][4];
  __shared__ unsigned long long s_present;  // points of this tile that receive any gradient
  __shared__ int s_rowmap[PM_TM];           // compact index -> point
  float *sD = reinterpret_cast<float *>(sD4), *sE = reinterpret_cast<float *>(sEF4), *sF = sD;
  int2 *list = reinterpret_cast<int2 *>(sEF4);  // [256 * BW_CH + 128] (channel | point << 16, gradient bits); dead before sE is written
  const int b = blockIdx.y, tile = blockIdx.x, ntiles = gridDim.x, n0 = tile * PM_TM, N = a.N, Cout = a.Cout;
  const int rows = min(PM_TM, N - n0);
  const size_t row0 = (size_t)b * N + n0;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int rb = wave & 1, cb = wave >> 1;

  // A max-pool routes every channel's gradient to ONE point, so only a few points of a tile (~6-12 of 64 in the bench
  // workload) receive any gradient -- and every operation of the chain is row-wise: a point with a zero gradient row
  // stays zero through every layer.  The whole stage therefore runs on the COMPACTED rows: the points hit by this
  // stage's gather plus those the previous stage reported (pres_in), renumbered 0..D-1.  One 32-row block instead of
  // two halves the matrix work (waves with rb >= R skip it), masks and incoming gradients are fetched for D rows only,
  // and zero rows add exact zeros, so the result is bit for bit that of the dense chain.
  const unsigned long long rowmask = rows >= 64 ? ~0ull : ((1ull << rows) - 1ull);
  const unsigned long long incoming =
      (STAGE == 2 ? 0ull : (a.pres_in != nullptr ? a.pres_in[(size_t)b * ntiles + tile] : ~0ull)) & rowmask;
  V3_STAMP_DECL;
  V3_STAMP(0);
  if (threadIdx.x == 0) s_present = incoming;
  __syncthreads();
  // ---- the arg-max table of the cloud
  int mn[BW_CH];
  float mg[BW_CH];
  // (all loads of this kernel are unconditional, from addresses clamped into range: a load under a condition -- or one
  // whose only use is a select, which the optimiser turns back into a load under a condition -- is compiled as a branch
  // with its own wait, and a handful of them become as many dependent global round trips (tools/v3_phases.py).  Compacted
  // rows D..32R-1 therefore carry the masks of row D-1: they multiply rows of the gradient tile that are exact zeros.)
  {
    int64_t ti[BW_CH];
    float tg[BW_CH], tm[BW_CH];
#pragma unroll
    for (int ch = 0; ch < BW_CH; ++ch) {
      const size_t o = (size_t)b * Cout + min(ch * 256 + (int)threadIdx.x, Cout - 1);
      ti[ch] = a.idx[o];
      tg[ch] = a.dg[o];
      tm[ch] = (a.gmask != nullptr ? a.gmask : a.dg)[o];  // a pointer select, not a load under a condition
    }
    const bool gated = a.gmask != nullptr;
#pragma unroll
    for (int ch = 0; ch < BW_CH; ++ch) {
      const bool in = ch * 256 + (int)threadIdx.x < Cout;
      mn[ch] = in ? (int)ti[ch] - n0 : -1;
      mg[ch] = (in && (!gated || tm[ch] > 0.f)) ? tg[ch] : 0.f;
    }
  }
  if (STAGE == 1 && threadIdx.x < 192) {
    const int c = threadIdx.x >> 6, n = threadIdx.x & 63;
    sX[n * 3 + c] = n < rows ? a.x[((size_t)b * 3 + c) * N + n0 + n] : 0.f;
  }
  // ---- which channels route their gradient into this tile: ordered compaction (ascending channel) into one list
  unsigned rank[BW_CH];
#pragma unroll
  for (int ch = 0; ch < BW_CH; ++ch) {
    const int n = mn[ch];
    const bool hit = n >= 0 && n < rows && mg[ch] != 0.f;
    const unsigned long long m = __ballot(hit);
    if (lane == 0) s_cnt[ch][wave] = __popcll(m);
    if (hit) atomicOr(&s_present, 1ull << n);  // integer OR: order-independent
    mn[ch] = hit ? n : -1;
    rank[ch] = __popcll(m & ((1ull << lane) - 1ull));
  }
  __syncthreads();
  V3_STAMP(1);
  const unsigned long long present = s_present;
  const int D = __popcll(present);
  const int R = (D + 31) >> 5;  // 32-row blocks of compacted points: 0, 1 or 2
  if (threadIdx.x == 0 && a.pres_out != nullptr) a.pres_out[(size_t)b * ntiles + tile] = present;
  if (D == 0) {  // block-uniform: nothing arrives in this tile
    if (STAGE == 2) {
      float4 *o = reinterpret_cast<float4 *>(a.dTpart + ((size_t)b * ntiles + tile) * 4096);
      for (int e = threadIdx.x; e < 1024; e += 256) o[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      if (STAGE == 1 && threadIdx.x < 9) a.dTpart[((size_t)b * ntiles + tile) * 9 + threadIdx.x] = 0.f;
      if (threadIdx.x < 192) {
        const int c = threadIdx.x >> 6, n = threadIdx.x & 63;
        if (n < rows) {
          const size_t o = ((size_t)b * 3 + c) * N + n0 + n;
          a.out[o] = STAGE == 0 ? 0.f + a.dPin[o] : 0.f;
        }
      }
    }
    return;
  }
  if (threadIdx.x < PM_TM && ((present >> threadIdx.x) & 1ull))
    s_rowmap[__popcll(present & ((1ull << threadIdx.x) - 1ull))] = threadIdx.x;
  int M = 0;
#pragma unroll
  for (int ch = 0; ch < BW_CH; ++ch)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w == wave && mn[ch] >= 0)
        list[M + rank[ch]] = make_int2((ch * 256 + (int)threadIdx.x) | (__popcll(present & ((1ull << mn[ch]) - 1ull)) << 16),
                                       __float_as_int(mg[ch]));
      M += s_cnt[ch][w];
    }
  if (threadIdx.x < 128) list[M + threadIdx.x] = make_int2(0, 0);  // zero-gradient padding: batches go in pairs and look one ahead
  __syncthreads();
  V3_STAMP(2);
  // ---- gather on the matrix cores:  dA2[D,128] = S[D,M] @ W3r[list,:]  with S[i,k] = g_k if channel k routes to
  //      the i-th compacted point (one non-zero per column, built on the fly from the list).  Wave w owns columns
  //      32w..32w+31; K runs over the list in order -> a fixed fmaf chain per output, and a point that wins hundreds
  //      of channels costs M/2 MFMAs per wave instead of a serial chain of dependent row adds.
  //      BW_MFMA_GATHER = false selects the plain form instead -- two waves, a column per lane, one fmaf per list entry
  //      into the LDS tile, the W3r rows of the next eight entries in flight: a fifth of the instruction slots, but a
  //      dependent LDS round trip per entry.  Measured: 28.8 instead of 24.2 us per launch alone, and the same 19.0
  //      clouds/s with three attacks in flight (the gather is worth 3.8 % there either way), so the MFMA form stays.
  f32x16 gacc[2];
  if (BW_MFMA_GATHER) {
    zero(gacc[0]);
    zero(gacc[1]);
    if (D > 32)  // block-uniform
      gather_rows<true>(list, M, a.W3r + 32 * wave + r, r, h, gacc);
    else
      gather_rows<false>(list, M, a.W3r + 32 * wave + r, r, h, gacc);
  } else {
    for (int e = threadIdx.x; e < 32 * R * 32; e += 256)
      *reinterpret_cast<float4 *>(sD + (e >> 5) * PM_L128 + 4 * (e & 31)) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (threadIdx.x < 128) {
      const float *Wc = a.W3r + threadIdx.x;  // this lane's column of W3r
      float *col = sD + threadIdx.x;
      float wq[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) wq[u] = Wc[(list[u].x & 0xffff) * 128];  // the list is padded with 32 zero-gradient entries
      for (int k0 = 0; k0 < M; k0 += 8) {
        float wn[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wn[u] = Wc[(list[k0 + 8 + u].x & 0xffff) * 128];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int2 en = list[k0 + u];
          float *p = col + (en.x >> 16) * PM_L128;
          *p = fmaf(__int_as_float(en.y), wq[u], *p);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) wq[u] = wn[u];
      }
    }
  }
  V3_STAMP(3);
  // ---- everything the chain will need from global memory, requested while the gather's last MFMAs run (the gather
  //      itself wants the registers: 32 rows of W3r in flight per lane, double buffered).  Rows are the compacted ones.
  const bool act = rb < R;  // wave-uniform: this wave's 32-row block holds compacted points
  float w2[64], w1[32];
  if (act) load_w<128, false>(a.W2r, 64, 32 * cb, r, h, w2);
  float m1v[16], mhv[16], dhv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int i = 32 * rb + acc_row(e, h);
    m1v[e] = STAGE != 2 ? a.A1[(row0 + s_rowmap[min(i, D - 1)]) * 64 + 32 * cb + r] : 0.f;
  }
  float4 a2[8];  // ReLU mask of the 64->128 layer, compacted rows
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int e = threadIdx.x + 256 * u;
    const int i = e >> 5;
    a2[u] = *reinterpret_cast<const float4 *>(a.A2 + (row0 + s_rowmap[min(i, D - 1)]) * 128 + 4 * (e & 31));
  }
  float4 h1t[4];
  if (STAGE == 2) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = threadIdx.x + 256 * u;
      const int i = e >> 4;
      h1t[u] = *reinterpret_cast<const float4 *>(a.H1 + (row0 + s_rowmap[min(i, D - 1)]) * 64 + 4 * (e & 15));
    }
  }
  if (BW_MFMA_GATHER) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = 32 * q + acc_row(e, h);
        if (q < R) sD[i * PM_L128 + 32 * wave + r] = gacc[q][e];  // rows D..32R-1 are zeros of the accumulator
      }
  }
  __syncthreads();
  V3_STAMP(4);
#pragma unroll
  for (int u = 0; u < 8; ++u) {  // ReLU of the 64->128 layer
    const int e = threadIdx.x + 256 * u;
    if ((e >> 5) < 32 * R) {
      float4 *d = reinterpret_cast<float4 *>(sD + (e >> 5) * PM_L128 + 4 * (e & 31));
      float4 v = *d;
      v.x = a2[u].x > 0.f ? v.x : 0.f; v.y = a2[u].y > 0.f ? v.y : 0.f;
      v.z = a2[u].z > 0.f ? v.z : 0.f; v.w = a2[u].w > 0.f ? v.w : 0.f;
      *d = v;
    }
  }
  __syncthreads();
  if (act) {  // through the 64->128 layer: [32R,128] @ W2r[128,64]
    // what only the layers behind this product need is requested now: its 64 MFMAs cover the round trip, and the
    // registers of the gather and of the mask pass are free again (requested earlier, stage 1 spilled 35 VGPRs)
    if (STAGE == 1) load_w<64, false>(a.W1r, 64, 32 * cb, r, h, w1);
    if (STAGE == 2) load_w<64, true>(a.T + (size_t)b * 4096, 64, 32 * cb, r, h, w1);
    if (STAGE == 1) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = 32 * rb + acc_row(e, h);
        // (these two stay loads under a condition on purpose: only the D compacted rows are fetched, and fetching all 32
        // measured slower -- tools/v3_phases.py, stage 1: 6.3 instead of 3.9 us for this phase)
        const bool in = i < D;
        const int p = s_rowmap[min(i, D - 1)];
        const size_t o = (row0 + p) * 64 + 32 * cb + r;
        const float hv = a.H1[o], dv = a.dH1in[o];  // rows outside pres_in were never written: read, then dropped
        mhv[e] = in ? hv : 0.f;
        dhv[e] = (in && ((incoming >> p) & 1ull)) ? dv : 0.f;
      }
    }
    f32x16 acc[1];
    zero(acc[0]);
    mfma_apply<128, 1>(sD, PM_L128, 32 * rb, w2, acc, r, h);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float v = acc[0][e];
      if (STAGE != 2) v = m1v[e] > 0.f ? v : 0.f;
      sE[(32 * rb + acc_row(e, h)) * PM_L64 + 32 * cb + r] = v;
    }
  }
  __syncthreads();
  V3_STAMP(5);
  if (STAGE == 2) {  // sD is dead: it now takes the h1 rows (left operand of the transform gradient)
    stash_tile<64>(h1t, sF, PM_L64);
    __syncthreads();
  }

  if (STAGE == 2) {
    // (a) dT64 partial of this tile:  sum_n h1[n,i] * d[n,j]   (A = h1^T, B = d, K = the compacted points, ascending)
    {
      f32x16 acc;
      zero(acc);
      for (int t = 0; t < 16 * R; ++t) {
        const int n = 2 * t + h;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sF[n * PM_L64 + 32 * rb + r], sE[n * PM_L64 + 32 * cb + r], acc, 0, 0, 0);
      }
      float *o = a.dTpart + ((size_t)b * ntiles + tile) * 4096;
#pragma unroll
      for (int e = 0; e < 16; ++e) o[(32 * rb + acc_row(e, h)) * 64 + 32 * cb + r] = acc[e];
    }
    // (b) dH1 = d @ T64^T, written for the compacted points only (the next stage reads exactly those rows)
    if (act) {
      f32x16 acc[1];
      zero(acc[0]);
      mfma_apply<64, 1>(sE, PM_L64, 32 * rb, w1, acc, r, h);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = 32 * rb + acc_row(e, h);
        if (i < D) a.out[(row0 + s_rowmap[i]) * 64 + 32 * cb + r] = acc[0][e];
      }
    }
    V3_STAMP(7);
    V3_STAMP_FLUSH();
    return;
  }

  const float *sIn = sE;
  if (STAGE == 1) {  // through t1, add the gradient arriving at h1 from the encoder, through e1's ReLU
    if (act) {
      f32x16 acc[1];
      zero(acc[0]);
      mfma_apply<64, 1>(sE, PM_L64, 32 * rb, w1, acc, r, h);
#pragma unroll
      for (int e = 0; e < 16; ++e)
        sF[(32 * rb + acc_row(e, h)) * PM_L64 + 32 * cb + r] = mhv[e] > 0.f ? acc[0][e] + dhv[e] : 0.f;
    }
    __syncthreads();
    sIn = sF;
  }
  if (wave < 3) {  // 64 -> 3 backwards:  g[i,c] = sum_k d[i,k] * W0r[k,c]; c is the wave: W0r comes through scalar loads
    const int c = wave, i = lane;
    float v = 0.f;
    if (i < D) {
#pragma unroll
      for (int k4 = 0; k4 < 16; ++k4) {
        const float4 d = *reinterpret_cast<const float4 *>(sIn + i * PM_L64 + 4 * k4);
        v = fmaf(d.x, a.W0r[(4 * k4) * 3 + c], v);
        v = fmaf(d.y, a.W0r[(4 * k4 + 1) * 3 + c], v);
        v = fmaf(d.z, a.W0r[(4 * k4 + 2) * 3 + c], v);
        v = fmaf(d.w, a.W0r[(4 * k4 + 3) * 3 + c], v);
      }
    }
    sG[i * 3 + c] = v;  // compacted row i
  }
  __syncthreads();
  V3_STAMP(6);
  if (STAGE == 0) {
    if (wave < 3) {
      const int c = wave, n = lane;
      if (n < rows) {
        const size_t o = ((size_t)b * 3 + c) * N + n0 + n;
        const float v = ((present >> n) & 1ull) ? sG[__popcll(present & ((1ull << n) - 1ull)) * 3 + c] : 0.f;
        a.out[o] = v + a.dPin[o];
      }
    }
  } else {
    const float *T = a.T + (size_t)b * 9;
    if (wave < 3) {  // dPts[n,i] = sum_j g[n,j] * T3[i,j]
      const int i = wave, n = lane;
      if (n < rows) {
        float v = 0.f;
        if ((present >> n) & 1ull) {
          const float *g = sG + __popcll(present & ((1ull << n) - 1ull)) * 3;
          v = fmaf(g[2], T[i * 3 + 2], fmaf(g[1], T[i * 3 + 1], g[0] * T[i * 3]));
        }
        a.out[((size_t)b * 3 + i) * N + n0 + n] = v;
      }
    } else if (threadIdx.x < 192 + 9) {  // dT3 partial[i,j] = sum_n x[n,i] * g[n,j], ascending n (the others add zero)
      const int q = threadIdx.x - 192, i = q / 3, j = q % 3;
      float v = 0.f;
      for (int ci = 0; ci < D; ++ci) v = fmaf(sX[s_rowmap[ci] * 3 + i], sG[ci * 3 + j], v);
      a.dTpart[((size_t)b * ntiles + tile) * 9 + q] = v;
    }
  }
  V3_STAMP(7);
  V3_STAMP_FLUSH();
}

// ------------------------------------------------------------------------------------------------------------------
// rowmlp_bwd_k on the fp16 matrix cores (mode 1): the same compaction, the same lists, the same order of operations per
// output; every matrix product takes its operands as two fp16 pieces (three MFMAs of 32 cycles per 16 values of k instead of
// eight of 64).  In a 128-cloud stack the f32 form spent about half of its time in the matrix pipe of two of the four SIMDs
// (R = 1: only the waves of the first row block compute).  Differences in structure: the gradient values of the list carry
// their two pieces instead of the fp32 bits; the ReLU mask of the 64 -> 128 layer is applied when the gathered tile leaves
// the accumulators (masks fetched in the accumulator's layout), so the tile exists in LDS only as pieces and the separate
// mask pass with its barrier is gone; the 64 -> 3 layer reads its rows back as hi + 2^-11 lo.
// dA2[D rows, 32 columns of this wave] = S[D,M] @ W3r[list,:] as above, 16 list entries per MFMA step.
// Round 5: the selection matrix S (one non-zero per column: entry k routes its gradient to compact row i_k) is built on the
// PACKED gradient word (fp16 hi | fp16 lo << 16): one compare and ONE select per entry, lane and row block, the two piece vectors
// then assembled by byte permutes (v_perm_b32, two per pair of entries) -- it used to be a decode of both pieces and a select
// per piece (~100 vector instructions per 8 entries; the gather was most of the kernel's vector work, and the counters put the
// kernel at 50 % vector-instruction time: profiles/r05_loop_traffic.json).  The same fragments, bit for bit.  (Tried first: the
// fragments out of a per-wave LDS tile that 16 lanes scatter the entries into -- four dependent LDS round trips per 8 entries:
// 2.4x SLOWER end to end.)
template <bool TWO>
__device__ __forceinline__ void gather_rows16(const int2 *list, int M, const float *__restrict__ Wc, int r, int h,
                                              f32x16 (&acc)[2], f32x16 (&accl)[2]) {
  if (M <= 0) return;
  int2 enA[16], enB[16];
  float bvA[16], bvB[16];
  auto request = [&](int2 (&en)[16], float (&bv)[16], int q) {  // 32 entries: step s (0, 1), element j of half h = entry q + 16 s + 8 h + j
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      en[t] = list[q + 16 * (t >> 3) + 8 * h + (t & 7)];
      bv[t] = Wc[(size_t)(en[t].x & 0xffff) * 128];
    }
  };
  const uint32_t row0 = (uint32_t)r << 16, row1 = (uint32_t)(32 + r) << 16;
  auto multiply = [&](const int2 (&en)[16], const float (&bv)[16]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float bq[8];
      uint32_t w0[8], w1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int2 e = en[8 * s + j];
        bq[j] = bv[8 * s + j];
        const uint32_t key = (uint32_t)e.x & 0xffff0000u;  // the entry's compact row << 16
        w0[j] = key == row0 ? (uint32_t)e.y : 0u;          // (hi piece | lo piece << 16) of the gradient value, or nothing
        if (TWO) w1[j] = key == row1 ? (uint32_t)e.y : 0u;
      }
      uint4 h0, l0, h1, l1;
      h0.x = __builtin_amdgcn_perm(w0[1], w0[0], 0x05040100u), l0.x = __builtin_amdgcn_perm(w0[1], w0[0], 0x07060302u);
      h0.y = __builtin_amdgcn_perm(w0[3], w0[2], 0x05040100u), l0.y = __builtin_amdgcn_perm(w0[3], w0[2], 0x07060302u);
      h0.z = __builtin_amdgcn_perm(w0[5], w0[4], 0x05040100u), l0.z = __builtin_amdgcn_perm(w0[5], w0[4], 0x07060302u);
      h0.w = __builtin_amdgcn_perm(w0[7], w0[6], 0x05040100u), l0.w = __builtin_amdgcn_perm(w0[7], w0[6], 0x07060302u);
      if (TWO) {
        h1.x = __builtin_amdgcn_perm(w1[1], w1[0], 0x05040100u), l1.x = __builtin_amdgcn_perm(w1[1], w1[0], 0x07060302u);
        h1.y = __builtin_amdgcn_perm(w1[3], w1[2], 0x05040100u), l1.y = __builtin_amdgcn_perm(w1[3], w1[2], 0x07060302u);
        h1.z = __builtin_amdgcn_perm(w1[5], w1[4], 0x05040100u), l1.z = __builtin_amdgcn_perm(w1[5], w1[4], 0x07060302u);
        h1.w = __builtin_amdgcn_perm(w1[7], w1[6], 0x05040100u), l1.w = __builtin_amdgcn_perm(w1[7], w1[6], 0x07060302u);
      }
      const h8v ah0 = __builtin_bit_cast(h8v, h0), al0 = __builtin_bit_cast(h8v, l0);
      uint4 wh, wl;
      split8v(bq, wh, wl);
      const h8v bh = __builtin_bit_cast(h8v, wh), bl = __builtin_bit_cast(h8v, wl);
      accl[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh, accl[0], 0, 0, 0);
      accl[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl, accl[0], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh, acc[0], 0, 0, 0);
      if (TWO) {
        const h8v ah1 = __builtin_bit_cast(h8v, h1), al1 = __builtin_bit_cast(h8v, l1);
        accl[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh, accl[1], 0, 0, 0);
        accl[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl, accl[1], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh, acc[1], 0, 0, 0);
      }
    }
  };
  request(enA, bvA, 0);
  for (int q = 0; q < M; q += 64) {
    request(enB, bvB, q + 32);
    multiply(enA, bvA);
    request(enA, bvA, q + 64);
    multiply(enB, bvB);
  }
}

// NW = 64-point words per block tile.  The work of a block is set by the points that RECEIVE gradient (the max routes each
// channel's gradient to one point: ~10 of 64), not by the points it covers, and a block's time is a chain of global round
// trips and barriers (~13 us at two blocks per CU): at NW = 2 a 128-cloud stack is two rounds of 1024 blocks instead of four
// of 2048.  The compacted rows still have to fit the 64-row piece tiles: a tile with more than 64 winning points (rare) is
// processed word by word (``passes``), each pass exactly the NW = 1 algorithm on its 64 points, the tile's transform-gradient
// partial summed over the passes in word order.
// FIXUP: the same kernel launched a second time behind the first (NW > 1 only).  The first launch (FIXUP = false) runs the
// single pass and nothing else -- with the pass loop in it the register allocator needs 256 registers and 0.3-0.8 KB of scratch
// per lane instead of 211-226 and none -- and marks a tile it cannot take (more than 64 winning points) in a.overflow; the
// second launch returns at once for every other tile and takes the marked ones word by word.
template <int STAGE, int NW, bool FIXUP>
// (the second launch -- tiles with more than 64 winning points, both words in turn -- keeps the lists of both passes alive: at
// two workgroups per CU it spilled 0.9-1.6 KB per lane to scratch and ran 6x slower than its work: 388 us per stage-1 launch on
// surface-like clouds, where half the tiles take it (round 5, tools/r05 sphere runs); one workgroup per CU: AGPRs, no scratch)
__global__ __launch_bounds__(256, FIXUP ? 1 : 2) void rowmlp_bwd16_k(RowMlpBwd a) {
  constexpr int BT = 64 * NW;  // points per block tile
  if (FIXUP && a.overflow[(size_t)blockIdx.y * gridDim.x + blockIdx.x] == 0) return;  // block-uniform
  if (NW == 1 && !FIXUP && a.fix_parent != nullptr) {  // block-uniform: this word's tile was done by the first launch
    if (a.fix_parent[(size_t)blockIdx.y * (gridDim.x >> 1) + (blockIdx.x >> 1)] == 0) return;
    if ((int)blockIdx.x * BT >= a.N) {  // (the second word of a ragged last tile: no points, a zero partial)
      if (STAGE == 2) {
        float4 *o = reinterpret_cast<float4 *>(a.dTpart + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4096);
        for (int e = threadIdx.x; e < 1024; e += 256) o[e] = make_float4(0.f, 0.f, 0.f, 0.f);
      } else if (STAGE == 1 && threadIdx.x < 9) {
        a.dTpart[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 9 + threadIdx.x] = 0.f;
      }
      return;
    }
  }
  // piece tiles: tD [2][64 rows][128] (the gathered, masked gradient; later reused as tF [2][64][64]), tE [2][64][64]
  __shared__ __attribute__((aligned(16))) char tD[2][PM_TM * PM_LH128], tE[2][PM_TM * PM_LH64];
  __shared__ float sX[BT * 3], sG[PM_TM * 3];
  __shared__ int s_cnt[BW_CH][4];
  __shared__ unsigned long long s_present[NW];
  __shared__ int s_rowmap[PM_TM];
  char *tF0 = tD[0], *tF1 = tD[0] + PM_TM * PM_LH64;   // two 64-wide piece tiles inside tD[0] (dead by then)
  int2 *list = reinterpret_cast<int2 *>(tE[0]);         // [256 * BW_CH + 128] entries = 9.2 KB <= 2 x 9.2 KB of tE; dead before tE is written
  const int b = blockIdx.y, tile = blockIdx.x, ntiles = gridDim.x, n0 = tile * BT, N = a.N, Cout = a.Cout;
  const int rows = min(BT, N - n0);
  const size_t row0 = (size_t)b * N + n0;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int rb = wave & 1, cb = wave >> 1;
  V3_STAMP_DECL;
  V3_STAMP(0);
  unsigned long long incoming[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const int rw = min(64, max(0, rows - 64 * w));
    const unsigned long long rowmask = rw >= 64 ? ~0ull : ((1ull << rw) - 1ull);
    incoming[w] = (STAGE == 2 ? 0ull : (a.pres_in != nullptr ? a.pres_in[((size_t)b * ntiles + tile) * NW + w] : ~0ull)) & rowmask;
  }
  if (threadIdx.x < NW) s_present[threadIdx.x] = incoming[threadIdx.x];
  __syncthreads();
  int mn[BW_CH];
  float mg[BW_CH];
  {
    int64_t ti[BW_CH];
    float tg[BW_CH], tm[BW_CH];
#pragma unroll
    for (int ch = 0; ch < BW_CH; ++ch) {
      const size_t o = (size_t)b * Cout + min(ch * 256 + (int)threadIdx.x, Cout - 1);
      ti[ch] = a.idx[o];
      tg[ch] = a.dg[o];
      tm[ch] = (a.gmask != nullptr ? a.gmask : a.dg)[o];
    }
    const bool gated = a.gmask != nullptr;
#pragma unroll
    for (int ch = 0; ch < BW_CH; ++ch) {
      const bool in = ch * 256 + (int)threadIdx.x < Cout;
      mn[ch] = in ? (int)ti[ch] - n0 : -1;
      mg[ch] = (in && (!gated || tm[ch] > 0.f)) ? tg[ch] : 0.f;
    }
  }
  if (STAGE == 1) {
    for (int e = threadIdx.x; e < 3 * BT; e += 256) {
      const int c = e / BT, n = e - c * BT;
      sX[n * 3 + c] = n < rows ? a.x[((size_t)b * 3 + c) * N + n0 + n] : 0.f;
    }
  }
  unsigned rank[BW_CH];
#pragma unroll
  for (int ch = 0; ch < BW_CH; ++ch) {  // ranks of the single-pass case (every hit of the tile), ready behind the same barrier
    const int n = mn[ch];
    const bool hit = n >= 0 && n < rows && mg[ch] != 0.f;
    const unsigned long long m = __ballot(hit);
    if (lane == 0) s_cnt[ch][wave] = __popcll(m);
    if (hit) atomicOr(&s_present[n >> 6], 1ull << (n & 63));
    mn[ch] = hit ? n : -1;
    rank[ch] = __popcll(m & ((1ull << lane) - 1ull));
  }
  __syncthreads();
  V3_STAMP(1);
  unsigned long long present[NW];
  int Dall = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    present[w] = s_present[w];
    Dall += __popcll(present[w]);
  }
  if (threadIdx.x < NW && a.pres_out != nullptr) a.pres_out[((size_t)b * ntiles + tile) * NW + threadIdx.x] = s_present[threadIdx.x];
  if (Dall == 0) {  // block-uniform: nothing arrives in this tile
    if (NW > 1 && !FIXUP && threadIdx.x == 0) a.overflow[(size_t)b * ntiles + tile] = 0;
    if (STAGE == 2) {
      float4 *o = reinterpret_cast<float4 *>(a.dTpart + ((size_t)b * ntiles + tile) * 4096);
      for (int e = threadIdx.x; e < 1024; e += 256) o[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      if (STAGE == 1 && threadIdx.x < 9) a.dTpart[((size_t)b * ntiles + tile) * 9 + threadIdx.x] = 0.f;
      for (int e = threadIdx.x; e < 3 * BT; e += 256) {
        const int c = e / BT, n = e - c * BT;
        if (n < rows) {
          const size_t o = ((size_t)b * 3 + c) * N + n0 + n;
          a.out[o] = STAGE == 0 ? 0.f + a.dPin[o] : 0.f;
        }
      }
    }
    V3_STAMP(7);
    V3_STAMP_FLUSH();
    return;
  }
  // One pass over all the tile's points (the common case), or -- more than 64 winning points -- one pass per word.
  if (NW > 1 && !FIXUP) {
    if (threadIdx.x == 0) a.overflow[(size_t)b * ntiles + tile] = Dall > PM_TM;
    if (Dall > PM_TM) return;  // block-uniform: left to the second launch
  }
  const bool single = !FIXUP;  // (a tile reaches the second launch only with more than 64 winning points)
  const int passes = single ? 1 : NW;
#pragma unroll 1
  for (int pass = 0; pass < passes; ++pass) {
    constexpr int keep = 0;
    // the points this pass works on (all of the tile's, or one 64-point word of it) and writes outputs for
    unsigned long long act[NW];
    int D = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      act[w] = (single || w == pass) ? present[w] : 0ull;
      D += __popcll(act[w]);
    }
    auto in_pass = [&](int n) { return single || (n >> 6) == pass; };
    auto cidx = [&](int n) {  // compact index of point n among the pass's winning points (ascending)
      int c = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        if (w < (n >> 6)) c += __popcll(act[w]);
        if (w == (n >> 6)) c += __popcll(act[w] & ((1ull << (n & 63)) - 1ull));
      }
      return c;
    };
    auto has = [&](const unsigned long long (&set)[NW], int n) {
      unsigned long long v = 0ull;
#pragma unroll
      for (int w = 0; w < NW; ++w)
        if (w == (n >> 6)) v = set[w];
      return ((v >> (n & 63)) & 1ull) != 0ull;
    };
    const bool first = pass == 0;
    if (!single) {  // (rare) the lists of this word alone: ranks again, behind the barrier that also retires the previous pass
      __syncthreads();
#pragma unroll
      for (int ch = 0; ch < BW_CH; ++ch) {
        const bool hit = mn[ch] >= 0 && in_pass(mn[ch]);
        const unsigned long long m = __ballot(hit);
        if (lane == 0) s_cnt[ch][wave] = __popcll(m);
        rank[ch] = __popcll(m & ((1ull << lane) - 1ull));
      }
      __syncthreads();
    }
    const int R = (D + 31) >> 5;
    if (D == 0) {  // (only in a multi-pass tile) a word without winners: zeros for its points, nothing to add to the partials
      if (STAGE == 2) {
        if (first) {
          float4 *o = reinterpret_cast<float4 *>(a.dTpart + ((size_t)b * ntiles + tile) * 4096);
          for (int e = threadIdx.x; e < 1024; e += 256) o[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      } else {
        if (STAGE == 1 && first && threadIdx.x < 9) a.dTpart[((size_t)b * ntiles + tile) * 9 + threadIdx.x] = 0.f;
        for (int e = threadIdx.x; e < 3 * BT; e += 256) {
          const int c = e / BT, n = e - c * BT;
          if (n < rows && in_pass(n)) {
            const size_t o = ((size_t)b * 3 + c) * N + n0 + n;
            a.out[o] = STAGE == 0 ? 0.f + a.dPin[o] : 0.f;
          }
        }
      }
      continue;
    }
    if (threadIdx.x < BT && has(act, threadIdx.x)) s_rowmap[cidx(threadIdx.x)] = threadIdx.x;
    int M = 0;
#pragma unroll
    for (int ch = 0; ch < BW_CH; ++ch)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        if (w == wave && mn[ch] >= 0 && in_pass(mn[ch])) {
          _Float16 gh, gl;
          split_pair(mg[ch], gh, gl);
          const uint32_t g2 = (uint32_t)__builtin_bit_cast(uint16_t, gh) | ((uint32_t)__builtin_bit_cast(uint16_t, gl) << 16);
          list[M + rank[ch]] = make_int2((ch * 256 + (int)threadIdx.x) | (cidx(mn[ch]) << 16), (int)g2);
        }
        M += s_cnt[ch][w];
      }
    if (threadIdx.x < 128) list[M + threadIdx.x] = make_int2(0, 0);
    __syncthreads();
    V3_STAMP(2);
    // ---- gather on the fp16 matrix cores
    f32x16 gacc[2], gaccl[2];
    zero(gacc[0]);
    zero(gacc[1]);
    zero(gaccl[0]);
    zero(gaccl[1]);
    if (D > 32)
      gather_rows16<true>(list, M, (a.W3r + keep) + 32 * wave + r, r, h, gacc, gaccl);
    else
      gather_rows16<false>(list, M, (a.W3r + keep) + 32 * wave + r, r, h, gacc, gaccl);
    V3_STAMP(3);
    // ---- everything the chain will need from global memory
    const bool act_rows = rb < R;
    uint4 w2h[8], w2l[8], w1h[4], w1l[4];
    if (act_rows) load_w16<128, false>((a.W2r + keep), 64, 32 * cb, r, h, w2h, w2l);
    float m1v[16], mhv[16], dhv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int i = 32 * rb + acc_row(e, h);
      m1v[e] = STAGE != 2 ? (a.A1 + keep)[(row0 + s_rowmap[min(i, D - 1)]) * 64 + 32 * cb + r] : 0.f;
    }
    // ReLU mask of the 64 -> 128 layer in the gather's accumulator layout: row 32 q + acc_row(e, h), column 32 wave + r
    float a2m[2][16];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = 32 * q + acc_row(e, h);
        // rows past D carry row D-1's mask: they are exact zeros.  Packed pieces (mode 2): a ReLU output is positive iff its word
        // (hi | lo << 16) is not zero
        a2m[q][e] = (a.A2 + keep)[(row0 + s_rowmap[min(i, D - 1)]) * 128 + 32 * wave + r];
      }
    float4 h1t[4];
    if (STAGE == 2) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = threadIdx.x + 256 * u;
        const int i = e >> 4;
        h1t[u] = *reinterpret_cast<const float4 *>((a.H1 + keep) + (row0 + s_rowmap[min(i, D - 1)]) * 64 + 4 * (e & 15));
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = 32 * q + acc_row(e, h);
        if (q < R) {
          const float v = joined(gacc[q], gaccl[q], e);
          const bool pos = a.a2_packed ? __float_as_uint(a2m[q][e]) != 0u : a2m[q][e] > 0.f;
          put_pieces(tD[0], tD[1], PM_LH128, i, 32 * wave + r, pos ? v : 0.f);
        }
      }
    __syncthreads();
    V3_STAMP(4);
    if (act_rows) {  // through the 64->128 layer: [32R,128] @ W2r[128,64]
      if (STAGE == 1) load_w16<64, false>((a.W1r + keep), 64, 32 * cb, r, h, w1h, w1l);
      if (STAGE == 2) load_w16<64, true>((a.T + keep) + (size_t)b * 4096, 64, 32 * cb, r, h, w1h, w1l);
      if (STAGE == 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int i = 32 * rb + acc_row(e, h);
          const bool in = i < D;
          const int p = s_rowmap[min(i, D - 1)];
          const size_t o = (row0 + p) * 64 + 32 * cb + r;
          const float hv = (a.H1 + keep)[o], dv = (a.dH1in + keep)[o];
          mhv[e] = in ? hv : 0.f;
          dhv[e] = (in && has(incoming, p)) ? dv : 0.f;
        }
      }
      f32x16 acc[1], accl[1];
      zero(acc[0]);
      zero(accl[0]);
      mfma_apply16<128, 1>(tD[0], tD[1], PM_LH128, 32 * rb, w2h, w2l, acc, accl, r, h);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = joined(acc[0], accl[0], e);
        if (STAGE != 2) v = m1v[e] > 0.f ? v : 0.f;
        put_pieces(tE[0], tE[1], PM_LH64, 32 * rb + acc_row(e, h), 32 * cb + r, v);
      }
    }
    __syncthreads();
    V3_STAMP(5);
    if (STAGE == 2) {  // tD is dead: its first half takes the h1 rows (left operand of the transform gradient) as pieces
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = threadIdx.x + 256 * u;
        const int n = e / 16, c = 4 * (e % 16);
        put_pieces(tF0, tF1, PM_LH64, n, c, h1t[u].x);
        put_pieces(tF0, tF1, PM_LH64, n, c + 1, h1t[u].y);
        put_pieces(tF0, tF1, PM_LH64, n, c + 2, h1t[u].z);
        put_pieces(tF0, tF1, PM_LH64, n, c + 3, h1t[u].w);
      }
      __syncthreads();
      // (a) dT64 partial of this tile:  sum_n h1[n,i] * d[n,j]   (A = h1^T, B = d, K = the compacted points, ascending): both
      //     operands are read down the rows of their tiles, eight 2-byte reads per piece and step
      {
        f32x16 acc, accl;
        zero(acc);
        zero(accl);
        for (int s16 = 0; s16 < 2 * R; ++s16) {
          h8v ah, al, bh, bl;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int n = 16 * s16 + 8 * h + j;
            ah[j] = *reinterpret_cast<const _Float16 *>(tF0 + n * PM_LH64 + 2 * (32 * rb + r));
            al[j] = *reinterpret_cast<const _Float16 *>(tF1 + n * PM_LH64 + 2 * (32 * rb + r));
            bh[j] = *reinterpret_cast<const _Float16 *>(tE[0] + n * PM_LH64 + 2 * (32 * cb + r));
            bl[j] = *reinterpret_cast<const _Float16 *>(tE[1] + n * PM_LH64 + 2 * (32 * cb + r));
          }
          accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accl, 0, 0, 0);
          accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accl, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        }
        float *o = a.dTpart + ((size_t)b * ntiles + tile) * 4096;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float *d = o + (32 * rb + acc_row(e, h)) * 64 + 32 * cb + r;
          const float v = joined(acc, accl, e);
          *d = first ? v : *d + v;  // a later pass adds to what this thread wrote in the pass before (word order)
        }
      }
      // (b) dH1 = d @ T64^T, written for the compacted points only
      if (act_rows) {
        f32x16 acc[1], accl[1];
        zero(acc[0]);
        zero(accl[0]);
        mfma_apply16<64, 1>(tE[0], tE[1], PM_LH64, 32 * rb, w1h, w1l, acc, accl, r, h);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int i = 32 * rb + acc_row(e, h);
          if (i < D) a.out[(row0 + s_rowmap[i]) * 64 + 32 * cb + r] = joined(acc[0], accl[0], e);
        }
      }
      continue;
    }

    const char *inH = tE[0], *inL = tE[1];
    if (STAGE == 1) {  // through t1, add the gradient arriving at h1 from the encoder, through e1's ReLU
      if (act_rows) {
        f32x16 acc[1], accl[1];
        zero(acc[0]);
        zero(accl[0]);
        mfma_apply16<64, 1>(tE[0], tE[1], PM_LH64, 32 * rb, w1h, w1l, acc, accl, r, h);
#pragma unroll
        for (int e = 0; e < 16; ++e)
          put_pieces(tF0, tF1, PM_LH64, 32 * rb + acc_row(e, h), 32 * cb + r, mhv[e] > 0.f ? joined(acc[0], accl[0], e) + dhv[e] : 0.f);
      }
      __syncthreads();
      inH = tF0;
      inL = tF1;
    }
    V3_STAMP(6);
    if (wave < 3) {  // 64 -> 3 backwards on the VALU:  g[i,c] = sum_k d[i,k] * W0r[k,c]; the row comes back as hi + 2^-11 lo
      const int c = wave, i = lane;
      float v = 0.f;
      if (i < D) {
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {
          const h8v dh = __builtin_bit_cast(h8v, *reinterpret_cast<const uint4 *>(inH + i * PM_LH64 + 16 * k8));
          const h8v dl = __builtin_bit_cast(h8v, *reinterpret_cast<const uint4 *>(inL + i * PM_LH64 + 16 * k8));
#pragma unroll
          for (int j = 0; j < 8; ++j) v = fmaf(fmaf((float)dl[j], 1.f / PM_SC, (float)dh[j]), (a.W0r + keep)[(8 * k8 + j) * 3 + c], v);
        }
      }
      sG[i * 3 + c] = v;
    }
    __syncthreads();
    if (STAGE == 0) {
      if (wave < 3) {
        const int c = wave;
#pragma unroll
        for (int n = lane; n < BT; n += 64) {
          if (n < rows && in_pass(n)) {
            const size_t o = ((size_t)b * 3 + c) * N + n0 + n;
            const float v = has(act, n) ? sG[cidx(n) * 3 + c] : 0.f;
            a.out[o] = v + a.dPin[o];
          }
        }
      }
    } else {
      const float *T = (a.T + keep) + (size_t)b * 9;
      if (wave < 3) {
        const int i = wave;
#pragma unroll
        for (int n = lane; n < BT; n += 64) {
          if (n < rows && in_pass(n)) {
            float v = 0.f;
            if (has(act, n)) {
              const float *g = sG + cidx(n) * 3;
              v = fmaf(g[2], T[i * 3 + 2], fmaf(g[1], T[i * 3 + 1], g[0] * T[i * 3]));
            }
            a.out[((size_t)b * 3 + i) * N + n0 + n] = v;
          }
        }
      } else if (threadIdx.x < 192 + 9) {
        const int q = threadIdx.x - 192, i = q / 3, j = q % 3;
        float v = 0.f;
        for (int ci = 0; ci < D; ++ci) v = fmaf(sX[s_rowmap[ci] * 3 + i], sG[ci * 3 + j], v);
        float *d = a.dTpart + ((size_t)b * ntiles + tile) * 9 + q;
        *d = first ? v : *d + v;
      }
    }
  }
  V3_STAMP(7);
  V3_STAMP_FLUSH();
}

// out[b,m] = sum_t part[b,t,m] (+ extra[b,m]), ascending t.
__global__ __launch_bounds__(256) void sum_partials_k(const float *__restrict__ part, const float *__restrict__ extra,
                                                      int T, int M, float *__restrict__ out, long long total) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const long long b = e / M;
  const int m = (int)(e % M);
  const float *p = part + (size_t)b * T * M + m;
  const float x = (extra ? extra : part)[extra ? e : 0];  // requested with the partials, not before them
  float v = extra ? x : 0.f;
  for (int t0 = 0; t0 < T; t0 += 16) {  // sixteen partials in flight, added in ascending t
    float q[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) q[t] = p[(size_t)min(t0 + t, T - 1) * M];
#pragma unroll
    for (int t = 0; t < 16; ++t) v += t0 + t < T ? q[t] : 0.f;
  }
  out[e] = v;
}

// out[B,NOUT] = act(in'[B,K] @ Wt[K,NOUT] + bias) for a handful of rows (one per cloud): weight-read bound, so
// the work is spread as wide as the weights allow: block = one 32(clouds) x 32(columns) output tile x one
// 128-deep K chunk; its 4 waves take 32 K-values each (16 MFMA steps), the input chunk is staged through LDS with
// coalesced loads (gated by mask > 0 on the way in: the ReLU backward).  K chunks of a tile meet through global
// partials: every block publishes its partial, the LAST one to arrive (atomic ticket) adds all of them in chunk
// order, applies bias / ReLU and writes the tile -> one launch, deterministic.  The ticket resets itself (the next
// launch on the stream sees the reset: kernel boundaries order it).
constexpr int FC_CH = 128;
constexpr int FC_LD = FC_CH + 4;
constexpr int FC_TICKETS = 16384;  // fixed-size ticket area at the head of the scratch (one per output tile)

// PRE: the input is itself a small product that is evaluated on the way in instead of by a launch of its own,
//   in[b,k] = sum_j (sum_t pre[b,t,j]) * Wpre[j,k]      (J <= 64; t ascending; j ascending within a lane half),
// the first layer of a backward stack (its K is the 9 or 40 outputs of the forward stack's last layer) together with the
// sum over the tiles' partials that produced its input.  The block sums the partials of its 32 rows into LDS once; every
// wave then computes the 32 x 32 slice of the chunk that it consumes itself (J/2 MFMAs).
struct FcPre {
  const float *pre;   // [B,T,J]
  const float *Wpre;  // [J,K]
  int T, J;
};

template <int PRE>  // 0: plain; otherwise the number of MFMA steps of the product in front (8: J <= 16, 32: J <= 64)
__global__ __launch_bounds__(256) void fc_layer_k(const float *__restrict__ in, const float *__restrict__ mask,
                                                  const float *__restrict__ Wt, const float *__restrict__ bias, int B,
                                                  int K, int NOUT, int relu, int chunk, float *__restrict__ out,
                                                  float *part, int *ticket, FcPre pp) {
  __shared__ float4 sA4[32 * FC_LD / 4];
  __shared__ float red[4 * 1024];
  __shared__ int s_last;
  float *sA = reinterpret_cast<float *>(sA4);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int cbk = blockIdx.x, rbk = blockIdx.y, ks = blockIdx.z, KS = gridDim.z;
  const int tile = rbk * gridDim.x + cbk, ntile = gridDim.x * gridDim.y;
  const int col = cbk * 32 + r, row0 = rbk * 32;
  const int kb0 = ks * chunk, kb1 = min(K, kb0 + chunk);
  const bool cok = col < NOUT;
  f32x16 acc;
  zero(acc);
  // every global operand of a chunk is requested in one go (addresses clamped into range, out-of-range values dropped
  // where they are used): this wave's 32 rows of Wt and, PRE, its column of Wpre and the mask values of its output slots
  constexpr int PS = PRE > 0 ? PRE : 1;
  float wv[16], bq[PS], mk[16], av[PS];
  float4 si[4], sm[4];  // plain form, K % 4 == 0: the thread's four float4 of the input chunk and of its mask
  const bool vec = !PRE && (K & 3) == 0;
  const float *mp = mask != nullptr ? mask : in;  // a pointer select, so that the mask load is not under a condition
  auto issue = [&](int kc) {
    if (vec) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = threadIdx.x + 256 * u;
        const size_t o = (size_t)min(row0 + (e >> 5), B - 1) * K + min(kc + 4 * (e & 31), K - 4);
        si[u] = *reinterpret_cast<const float4 *>(in + o);
        sm[u] = *reinterpret_cast<const float4 *>(mp + o);
      }
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int k = kc + 32 * wave + kmap(t, h);
      wv[t] = (k < kb1 && cok) ? Wt[(size_t)k * NOUT + col] : 0.f;
    }
    if (PRE) {
      const int kk = min(kc + 32 * wave + r, K - 1);
#pragma unroll
      for (int t = 0; t < PS; ++t) bq[t] = pp.Wpre[(size_t)min(2 * t + h, pp.J - 1) * K + kk];
#pragma unroll
      for (int e = 0; e < 16; ++e)
        mk[e] = mask != nullptr ? mask[(size_t)min(row0 + acc_row(e, h), B - 1) * K + kk] : 1.f;
    }
  };
  issue(kb0);
  if (PRE) {  // P[32,J] = the summed partials of this block's rows, into the (still unused) reduction array
    if (pp.T == 1) {  // nothing to sum: a thread's (up to eight) values in one batch of loads
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = min((int)threadIdx.x + 256 * u, 32 * pp.J - 1);
        v[u] = pp.pre[(size_t)min(row0 + e / pp.J, B - 1) * pp.J + e % pp.J];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = threadIdx.x + 256 * u;
        if (e < 32 * pp.J) red[(e / pp.J) * 65 + e % pp.J] = row0 + e / pp.J < B ? v[u] : 0.f;
      }
    } else {
      for (int e = threadIdx.x; e < 32 * pp.J; e += 256) {
        const int rr = e / pp.J, j = e - rr * pp.J;
        const float *p = pp.pre + ((size_t)min(row0 + rr, B - 1) * pp.T) * pp.J + j;
        float v = 0.f;
        for (int t0 = 0; t0 < pp.T; t0 += 16) {  // sixteen partials in flight, added in ascending t
          float q[16];
#pragma unroll
          for (int t = 0; t < 16; ++t) q[t] = p[(size_t)min(t0 + t, pp.T - 1) * pp.J];
#pragma unroll
          for (int t = 0; t < 16; ++t) v += t0 + t < pp.T ? q[t] : 0.f;
        }
        red[rr * 65 + j] = row0 + rr < B ? v : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < PS; ++t) av[t] = 2 * t + h < pp.J ? red[r * 65 + min(2 * t + h, 63)] : 0.f;
    __syncthreads();  // every wave holds its operands: the array is free for the accumulators
  }
  for (int kc = kb0; kc < kb1; kc += FC_CH) {
    if (kc != kb0) {
      issue(kc);
      __syncthreads();
    }
    if (PRE) {
      f32x16 pa;
      zero(pa);
#pragma unroll
      for (int t = 0; t < PS; ++t)
        if (2 * t < pp.J) pa = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bq[t], pa, 0, 0, 0);  // wave-uniform guard
      const bool kok = kc + 32 * wave + r < kb1;
#pragma unroll
      for (int e = 0; e < 16; ++e)
        sA[acc_row(e, h) * FC_LD + 32 * wave + r] = (kok && mk[e] > 0.f) ? pa[e] : 0.f;  // rows past B are zeros of P
    } else if ((K & 3) == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = threadIdx.x + 256 * u, rr = e >> 5, k = kc + 4 * (e & 31);
        const bool ok = row0 + rr < B && k < kb1, gated = mask != nullptr;
        const float4 x = si[u], g = sm[u];
        float4 v;
        v.x = (ok && (!gated || g.x > 0.f)) ? x.x : 0.f;
        v.y = (ok && (!gated || g.y > 0.f)) ? x.y : 0.f;
        v.z = (ok && (!gated || g.z > 0.f)) ? x.z : 0.f;
        v.w = (ok && (!gated || g.w > 0.f)) ? x.w : 0.f;
        *reinterpret_cast<float4 *>(sA + rr * FC_LD + 4 * (e & 31)) = v;
      }
    } else {
      for (int e = threadIdx.x; e < 32 * FC_CH; e += 256) {
        const int rr = e >> 7, k = kc + (e & 127);
        float v = 0.f;
        if (row0 + rr < B && k < kb1) {
          const size_t o = (size_t)(row0 + rr) * K + k;
          v = in[o];
          if (mask != nullptr) v = mask[o] > 0.f ? v : 0.f;
        }
        sA[rr * FC_LD + (e & 127)] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 a = *reinterpret_cast<const float4 *>(sA + r * FC_LD + 32 * wave + 8 * j + 4 * h);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wv[4 * j], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wv[4 * j + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wv[4 * j + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wv[4 * j + 3], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) red[(wave * 16 + e) * 64 + lane] = acc[e];
  __syncthreads();
  float v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int o = threadIdx.x + 256 * u;
    v[u] = ((red[o] + red[1024 + o]) + red[2048 + o]) + red[3072 + o];
  }
  if (KS > 1) {
    // partials are written through (sc1) and read back with sc1 loads by the last block to arrive (common.hpp)
#pragma unroll
    for (int u = 0; u < 4; ++u)
      __hip_atomic_store(&part[((size_t)ks * ntile + tile) * 1024 + threadIdx.x + 256 * u], v[u], __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    if (!handoff_last_arriver(ticket, tile, KS, &s_last)) return;  // protocol: common.hpp
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int q0 = 0; q0 < KS; q0 += 8) {  // 32 loads in flight per thread, then added in chunk order
      float t[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          t[u][i] = q0 + i < KS ? __hip_atomic_load(&part[((size_t)(q0 + i) * ntile + tile) * 1024 + threadIdx.x + 256 * u],
                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[u] += t[u][i];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = a[u];
    if (threadIdx.x == 0) __hip_atomic_store(&ticket[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int o = threadIdx.x + 256 * u, e = o >> 6, l = o & 63;
    const int orow = row0 + acc_row(e, l >> 5), ocol = cbk * 32 + (l & 31);
    if (orow < B && ocol < NOUT) {
      float y = v[u] + (bias ? bias[ocol] : 0.f);
      if (relu) y = y > 0.f ? y : 0.f;
      out[(size_t)orow * NOUT + ocol] = y;
    }
  }
}

// The same layer for K = 64 NWV (used at K = 256 and 512) WITHOUT a split over blocks or a chunk loop: one block of NWV waves per 32 x 32 output tile, wave
// w takes the 64-deep slice k = 64 w .. 64 w + 63 -- its own staging area for its slice of the input rows (coalesced
// float4 loads, gated by the mask on the way in, no block barrier: only this wave reads it), 32 MFMAs -- and the NWV
// partial tiles meet in LDS, added in wave order.  No global partials, no ticket, no second pass by a last block: two
// dependent global round trips less than the split form; the price is that the tile's 64 NWV x 32 weights are read by
// one CU.
constexpr int FCW_LD = 68;  // floats per staged input row (64 + 4: conflict-free ds_read_b128)

// PRE > 0 (the number of MFMA steps of the product in front, as in fc_layer_k): the wave's 32 x 64 slice of the input is
// itself computed, P[32,J] @ Wpre[J, its 64 columns], from the block's summed partials P (behind the staging areas in LDS).
template <int NWV, int PRE>
__global__ __launch_bounds__(NWV * 64) void fc_wide_k(const float *__restrict__ in, const float *__restrict__ mask,
                                                      const float *__restrict__ Wt, const float *__restrict__ bias, int B,
                                                      int K, int NOUT, int relu, float *__restrict__ out, FcPre pp) {
  extern __shared__ __attribute__((aligned(16))) float fcw_sm[];  // NWV x 32 x FCW_LD staging, then NWV x 1024 partials
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int col = blockIdx.x * 32 + r, row0 = blockIdx.y * 32;
  const int cc = min(col, NOUT - 1);
  const int k0 = 64 * wave;
  // every operand is requested before anything is used; out-of-range columns read column NOUT - 1 and are never stored
  float wv[32];
#pragma unroll
  for (int t = 0; t < 32; ++t) wv[t] = Wt[(size_t)(k0 + kmap(t, h)) * NOUT + cc];
  const bool gated = mask != nullptr;
  float *sW = fcw_sm + (size_t)wave * 32 * FCW_LD;
  if (PRE == 0) {
    const float *mp = mask != nullptr ? mask : in;
    float4 x[8], g[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = lane + 64 * u;
      const size_t o = (size_t)min(row0 + (e >> 4), B - 1) * K + k0 + 4 * (e & 15);
      x[u] = *reinterpret_cast<const float4 *>(in + o);
      g[u] = *reinterpret_cast<const float4 *>(mp + o);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = lane + 64 * u;
      const bool ok = row0 + (e >> 4) < B;
      float4 v;
      v.x = (ok && (!gated || g[u].x > 0.f)) ? x[u].x : 0.f;
      v.y = (ok && (!gated || g[u].y > 0.f)) ? x[u].y : 0.f;
      v.z = (ok && (!gated || g[u].z > 0.f)) ? x[u].z : 0.f;
      v.w = (ok && (!gated || g[u].w > 0.f)) ? x[u].w : 0.f;
      *reinterpret_cast<float4 *>(sW + (e >> 4) * FCW_LD + 4 * (e & 15)) = v;
    }
  } else {
    constexpr int PS = PRE > 0 ? PRE : 1;
    float *sP = fcw_sm + (size_t)NWV * 32 * FCW_LD;  // [32][65]
    // this wave's columns of Wpre and the mask values of its output slots: requested with the weights
    float bq[2][PS], mk[2][16];
    const float *mp = mask != nullptr ? mask : Wt;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int kk = k0 + 32 * ct + r;
#pragma unroll
      for (int t = 0; t < PS; ++t) bq[ct][t] = pp.Wpre[(size_t)min(2 * t + h, pp.J - 1) * K + kk];
#pragma unroll
      for (int e = 0; e < 16; ++e) mk[ct][e] = mp[gated ? (size_t)min(row0 + acc_row(e, h), B - 1) * K + kk : 0];
    }
    // P[32,J]: the block sums its rows' partials once (ascending t), as fc_layer_k does
    if (pp.T == 1) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = min((int)threadIdx.x + NWV * 64 * u, 32 * pp.J - 1);
        v[u] = pp.pre[(size_t)min(row0 + e / pp.J, B - 1) * pp.J + e % pp.J];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = threadIdx.x + NWV * 64 * u;
        if (e < 32 * pp.J) sP[(e / pp.J) * 65 + e % pp.J] = row0 + e / pp.J < B ? v[u] : 0.f;
      }
    } else {
      for (int e = threadIdx.x; e < 32 * pp.J; e += NWV * 64) {
        const int rr = e / pp.J, j = e - rr * pp.J;
        const float *p = pp.pre + ((size_t)min(row0 + rr, B - 1) * pp.T) * pp.J + j;
        float v = 0.f;
        for (int t0 = 0; t0 < pp.T; t0 += 16) {
          float q[16];
#pragma unroll
          for (int t = 0; t < 16; ++t) q[t] = p[(size_t)min(t0 + t, pp.T - 1) * pp.J];
#pragma unroll
          for (int t = 0; t < 16; ++t) v += t0 + t < pp.T ? q[t] : 0.f;
        }
        sP[rr * 65 + j] = row0 + rr < B ? v : 0.f;
      }
    }
    __syncthreads();
    float av[PS];
#pragma unroll
    for (int t = 0; t < PS; ++t) av[t] = 2 * t + h < pp.J ? sP[r * 65 + min(2 * t + h, 63)] : 0.f;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      f32x16 pa;
      zero(pa);
#pragma unroll
      for (int t = 0; t < PS; ++t)
        if (2 * t < pp.J) pa = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bq[ct][t], pa, 0, 0, 0);  // wave-uniform guard
#pragma unroll
      for (int e = 0; e < 16; ++e)
        sW[acc_row(e, h) * FCW_LD + 32 * ct + r] = (!gated || mk[ct][e] > 0.f) ? pa[e] : 0.f;  // rows past B are zeros of P
    }
  }
  HITADV_WAVE_LDS_HANDOFF();  // sW was written by this wave's lanes and is read by them only: no block barrier (see the kernel's header)
  f32x16 acc;
  zero(acc);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float4 a = *reinterpret_cast<const float4 *>(sW + r * FCW_LD + 8 * j + 4 * h);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wv[4 * j], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wv[4 * j + 1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wv[4 * j + 2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wv[4 * j + 3], acc, 0, 0, 0);
  }
  __syncthreads();  // every wave is done with its staging area: the memory becomes the partial tiles
#pragma unroll
  for (int e = 0; e < 16; ++e) fcw_sm[(size_t)wave * 1024 + e * 64 + lane] = acc[e];
  __syncthreads();
  for (int o = threadIdx.x; o < 1024; o += NWV * 64) {
    float v = fcw_sm[o];
#pragma unroll
    for (int w = 1; w < NWV; ++w) v += fcw_sm[(size_t)w * 1024 + o];
    const int e = o >> 6, l = o & 63;
    const int orow = row0 + acc_row(e, l >> 5), ocol = blockIdx.x * 32 + (l & 31);
    if (orow < B && ocol < NOUT) {
      float y = v + (bias ? bias[ocol] : 0.f);
      if (relu) y = y > 0.f ? y : 0.f;
      out[(size_t)orow * NOUT + ocol] = y;
    }
  }
}

template <int NWV, int PRE>
static int launch_fc_wide(const float *in, const float *mask, const float *Wt, const float *bias, int B, int K, int NOUT,
                           int relu, float *out, FcPre pp, hipStream_t s) {
  constexpr int shm = NWV * 32 * FCW_LD * 4 + (PRE ? 32 * 65 * 4 : 0);  // staging (>= NWV * 4096 bytes of partials) + P
  HITADV_RAISE_LDS((&fc_wide_k<NWV, PRE>), shm);
  dim3 grid((NOUT + 31) / 32, (B + 31) / 32);
  fc_wide_k<NWV, PRE><<<grid, NWV * 64, shm, s>>>(in, mask, Wt, bias, B, K, NOUT, relu, out, pp);
  HITADV_LAUNCH_CHECK();
  return 0;
}

static void fc_split(int B, int K, int NOUT, int *chunk, int *KS, int *tiles) {
  // 128-deep chunks per block, measured at B = 32 (tools/fc_split_probe.py): K <= 256 runs without a split (no hand-off:
  // 6.5 instead of 7.0 us), 512 and 1024 one chunk per block (two chunks: +1.0 .. +1.6 us), 4096 two (10.8 instead of
  // 12.6 us: half the partials for the last block to add); never more than 32 splits per tile
  int n = K <= 2 * FC_CH ? 2 : (K >= 16 * FC_CH ? 2 : 1);
  while ((K + FC_CH * n - 1) / (FC_CH * n) > 32) ++n;
  static const int force = [] { const char *e = getenv("HITADV_FC_CHUNKS"); return e ? atoi(e) : 0; }();  // tuning only
  if (force > 0) n = force > n ? force : n;
  int c = FC_CH * n;
  *chunk = c;
  *KS = (K + c - 1) / c;
  *tiles = ((NOUT + 31) / 32) * ((B + 31) / 32);
}

}  // namespace hitadv

using namespace hitadv;

// which kernel serves the fp16 modes of hitadv_pointnet_rowmlp_fwd*: 0 = rowmlp_stream_k, 1 = rowmlp_fwd16_k (one 64-point
// tile per workgroup: the round-3 kernel, kept as the reference the streaming one is held to, bit for bit), 2 = the
// streaming kernel except for stage 0 with the deformation inside (its 192 exp / sqrt per point want more waves per SIMD than
// the streaming kernel's two: 81 vs 86 us per 256 clouds, tools/v2_probe.py)
static int g_rowmlp_form = 2;
extern "C" int hitadv_pointnet_rowmlp_form(int form) {
  const int before = g_rowmlp_form;
  if (form >= 0 && form <= 2) g_rowmlp_form = form;
  return before;
}

template <int STAGE>
static int launch_rowmlp_stream(const RowMlpFwd &a, int B, hipStream_t s) {
  const int tpc = (a.N + PM_TM - 1) / PM_TM;  // tiles per cloud
  const long long total = (long long)B * tpc;
  if (total > 0x7fffffffLL) return HITADV_E_ARG;
  // two workgroups per CU; a workgroup's run of tiles is a whole number of clouds or a whole fraction of one (the per-cloud
  // work -- STN3d's last layer, the 64 x 64 transform's pieces -- is then done once per run)
  int tpb = (int)((total + 511) / 512);
  if (tpb >= tpc) tpb = ((tpb + tpc - 1) / tpc) * tpc;
  else
    while (tpc % tpb) ++tpb;
  const int blocks = (int)((total + tpb - 1) / tpb);
  constexpr int shm = rowmlp_stream_lds<STAGE>();
  HITADV_RAISE_LDS((&rowmlp_stream_k<STAGE>), shm);
  rowmlp_stream_k<STAGE><<<blocks, 256, shm, s>>>(a, tpc, (int)total, tpb);
  return 0;
}
static int launch_rowmlp16(int stage, const RowMlpFwd &a, int B, hipStream_t s) {
  if (g_rowmlp_form == 0 || (g_rowmlp_form == 2 && !(stage == 0 && a.d_ori != nullptr))) {
    if (stage == 0) return launch_rowmlp_stream<0>(a, B, s);
    if (stage == 1) return launch_rowmlp_stream<1>(a, B, s);
    return launch_rowmlp_stream<2>(a, B, s);
  }
  dim3 grid((a.N + PM_TM - 1) / PM_TM, B);
  if (stage == 0) rowmlp_fwd16_k<0><<<grid, 256, 0, s>>>(a);
  else if (stage == 1) rowmlp_fwd16_k<1><<<grid, 256, 0, s>>>(a);
  else rowmlp_fwd16_k<2><<<grid, 256, 0, s>>>(a);
  return 0;
}

extern "C" int hitadv_pointnet_rowmlp_fwd(int stage, const float *x, const float *T, const float *hin, const float *W0,
                                          const float *b0, const float *W1, const float *b1, const float *W2,
                                          const float *b2, float *xp, float *o0, float *o1, float *o2, int B, int N,
                                          int mode, int32_t *range_flag, void *stream) {
  HITADV_ABLATE_RETURN("v2");
  if (stage < 0 || stage > 2 || B <= 0 || N <= 0 || !W2 || !b2 || !o2 || mode < 0 || mode > 2) return HITADV_E_ARG;
  if (stage < 2 && (!x || !W0 || !b0 || !o0)) return HITADV_E_ARG;
  if (stage == 1 && (!T || !W1 || !b1 || !o1)) return HITADV_E_ARG;
  if (stage == 2 && (!T || !hin)) return HITADV_E_ARG;
  RowMlpFwd a{x, T, hin, W0, b0, W1, b1, W2, b2, xp, o0, o1, o2, N, nullptr, nullptr, nullptr, nullptr,
              nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, mode == 2, range_flag};
  dim3 grid((N + PM_TM - 1) / PM_TM, B);
  hipStream_t s = (hipStream_t)stream;
  if (mode >= 1) {
    const int rc = launch_rowmlp16(stage, a, B, s);
    if (rc != 0) return rc;
  } else if (stage == 0) rowmlp_fwd_k<0><<<grid, 256, 0, s>>>(a);
  else if (stage == 1) rowmlp_fwd_k<1><<<grid, 256, 0, s>>>(a);
  else rowmlp_fwd_k<2><<<grid, 256, 0, s>>>(a);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_pointnet_rowmlp_fwd_stn(const float *x, const float *F5, const float *W6, const float *b6,
                                              float *Tout, const float *W0, const float *b0, const float *W1,
                                              const float *b1, const float *W2, const float *b2, float *xp, float *o0,
                                              float *o1, float *o2, int B, int N, int mode, int32_t *range_flag,
                                              void *stream) {
  HITADV_ABLATE_RETURN("v2");
  if (B <= 0 || N <= 0 || !x || !F5 || !W6 || !b6 || !Tout || !W0 || !b0 || !W1 || !b1 || !W2 || !b2 || !o0 || !o1 || !o2 ||
      mode < 0 || mode > 2)
    return HITADV_E_ARG;
  RowMlpFwd a{x, nullptr, nullptr, W0, b0, W1, b1, W2, b2, xp, o0, o1, o2, N, F5, W6, b6, Tout,
              nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, mode == 2, range_flag};
  dim3 grid((N + PM_TM - 1) / PM_TM, B);
  if (mode >= 1) {
    const int rc = launch_rowmlp16(1, a, B, (hipStream_t)stream);
    if (rc != 0) return rc;
  } else rowmlp_fwd_k<1><<<grid, 256, 0, (hipStream_t)stream>>>(a);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_pointnet_rowmlp_fwd_deform(const float *ori, const float *central, const float *perturb,
                                                 const float *sigma, int C, float *adv, float *inv_den, const float *W0,
                                                 const float *b0, const float *W2, const float *b2, float *o0, float *o2,
                                                 int B, int N, int mode, int32_t *range_flag, void *stream) {
  HITADV_ABLATE_RETURN("v2");
  if (B <= 0 || N <= 0 || C <= 0 || C > 256 || !ori || !central || !perturb || !sigma || !adv || !inv_den || !W0 || !b0 ||
      !W2 || !b2 || !o0 || !o2 || mode < 0 || mode > 2)
    return HITADV_E_ARG;
  RowMlpFwd a{adv, nullptr, nullptr, W0, b0, nullptr, nullptr, W2, b2, nullptr, o0, nullptr, o2, N, nullptr, nullptr, nullptr,
              nullptr, ori, central, perturb, sigma, adv, inv_den, C, mode == 2, range_flag};
  dim3 grid((N + PM_TM - 1) / PM_TM, B);
  if (mode >= 1) {
    const int rc = launch_rowmlp16(0, a, B, (hipStream_t)stream);
    if (rc != 0) return rc;
  } else rowmlp_fwd_k<0><<<grid, 256, 0, (hipStream_t)stream>>>(a);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t hitadv_pointnet_rowmlp_tiles(int N) { return N > 0 ? (N + PM_TM - 1) / PM_TM : 0; }

// 64-point words per block tile of the backward kernels: 2 in the fp16 modes, 1 in the f32 mode -- for EVERY launch size.
// (At B = 32 x 1024 points one word is exactly one round of blocks and the two-word form only adds its second launch: one
// attack alone runs at 329 instead of 312 us per iteration.  But the two forms sum a tile's rows in different groupings, and
// an attack inside a 128-cloud stack has to return the bits of the same attack run alone (B = 32): one form for all sizes.)
// HITADV_V3_WORDS=1|2 overrides (A/B timing).  Measured, cfg2 headline job (128-cloud launches): one word 43.7 clouds/s,
// two 48.2 (a second box: 47.7), four 47.0 (more tiles over 64 winning points, longer gathers).
extern "C" int hitadv_pointnet_rowmlp_bwd_words(int B, int N, int mode) {
  static const int forced = [] { const char *e = getenv("HITADV_V3_WORDS"); return e ? atoi(e) : 0; }();
  (void)B;
  (void)N;
  if (mode < 1) return 1;
  return forced == 1 ? 1 : 2;
}
extern "C" int64_t hitadv_pointnet_rowmlp_bwd_tiles(int N, int words) {
  const int bt = PM_TM * (words == 2 ? 2 : 1);
  return N > 0 ? (N + bt - 1) / bt : 0;
}

// how the two-word form handles a tile with more than 64 winning points: 1 (default) = two blocks of the one-word kernel,
// 0 = the two-pass instantiation of the two-word kernel (round 4)
static int g_v3_fix_form = [] { const char *e = getenv("HITADV_V3_FIX"); return e ? atoi(e) : 1; }();

// part[b,t,:] = fix[b,2t,:] + fix[b,2t+1,:] for the tiles the two-word launch left to the one-word launch (word order)
__global__ __launch_bounds__(256) void merge_fix_partials_k(float *__restrict__ part, const float *__restrict__ fix,
                                                            const int *__restrict__ over, int M) {
  const size_t bt = blockIdx.x;
  if (over[bt] == 0) return;
  for (int m = threadIdx.x; m < M; m += 256) part[bt * M + m] = fix[2 * bt * M + m] + fix[(2 * bt + 1) * M + m];
}

extern "C" int hitadv_pointnet_rowmlp_bwd(int stage, const float *dg, const float *gmask, const int64_t *idx,
                                          const float *W3r, int Cout, const float *A2, const float *W2r,
                                          const float *A1, const float *W1r, const float *H1, const float *dH1in,
                                          const float *W0r, const float *T, const float *x, const float *dPin,
                                          float *dTpart, float *out, const uint64_t *pres_in, uint64_t *pres_out,
                                          int32_t *overflow, int words, int B, int N, int mode, void *stream) {
  return hitadv_pointnet_rowmlp_bwd_fix(stage, dg, gmask, idx, W3r, Cout, A2, W2r, A1, W1r, H1, dH1in, W0r, T, x, dPin, dTpart, out,
                                        pres_in, pres_out, overflow, words, B, N, mode, nullptr, stream);
}

extern "C" int hitadv_pointnet_rowmlp_bwd_fix(int stage, const float *dg, const float *gmask, const int64_t *idx,
                                              const float *W3r, int Cout, const float *A2, const float *W2r,
                                              const float *A1, const float *W1r, const float *H1, const float *dH1in,
                                              const float *W0r, const float *T, const float *x, const float *dPin,
                                              float *dTpart, float *out, const uint64_t *pres_in, uint64_t *pres_out,
                                              int32_t *overflow, int words, int B, int N, int mode, float *dTfix,
                                              void *stream) {
  HITADV_ABLATE_RETURN("v3");
  if (stage < 0 || stage > 2 || B <= 0 || N <= 0 || N > 65535 || Cout <= 0 || Cout > 256 * BW_CH || !dg || !idx ||
      !W3r || !A2 || !W2r || !out || mode < 0 || mode > 2 || words < 1 || words > 2 || (mode == 0 && words != 1))
    return HITADV_E_ARG;
  if (stage == 0 && (!A1 || !W0r || !dPin)) return HITADV_E_ARG;
  if (stage == 1 && (!A1 || !W1r || !H1 || !dH1in || !W0r || !T || !x || !dTpart)) return HITADV_E_ARG;
  if (stage == 2 && (!H1 || !T || !dTpart)) return HITADV_E_ARG;
  RowMlpBwd a{dg, gmask, idx, W3r, A2, W2r, A1, W1r, H1, dH1in, W0r, T, x, dPin, dTpart, out,
              reinterpret_cast<const unsigned long long *>(pres_in), reinterpret_cast<unsigned long long *>(pres_out), N,
              Cout, mode == 2, overflow, nullptr};
  dim3 grid((unsigned)hitadv_pointnet_rowmlp_bwd_tiles(N, words), B);
  hipStream_t s = (hipStream_t)stream;
  if (mode >= 1 && words == 2 && (dTfix != nullptr || stage == 0) && overflow && g_v3_fix_form == 1) {
    // Round 5: the tiles with more than 64 winning points go to the ONE-WORD kernel (two blocks per tile, each a plain
    // single pass), not to the two-pass instantiation (which keeps both passes' lists alive and spilled: 388 us per stage-1
    // launch on surface-like clouds, where half the tiles take this path; the one-word kernel does ALL tiles in 60).  Their
    // per-tile partials are the two words' partials added in word order.
    RowMlpBwd f = a;
    f.fix_parent = overflow;
    f.overflow = nullptr;
    f.dTpart = dTfix;
    const dim3 grid1(2 * grid.x, B);
    if (stage == 0) {
      rowmlp_bwd16_k<0, 2, false><<<grid, 256, 0, s>>>(a);
      rowmlp_bwd16_k<0, 1, false><<<grid1, 256, 0, s>>>(f);
    } else if (stage == 1) {
      rowmlp_bwd16_k<1, 2, false><<<grid, 256, 0, s>>>(a);
      rowmlp_bwd16_k<1, 1, false><<<grid1, 256, 0, s>>>(f);
      merge_fix_partials_k<<<grid.x * B, 256, 0, s>>>(dTpart, dTfix, overflow, 9);
    } else {
      rowmlp_bwd16_k<2, 2, false><<<grid, 256, 0, s>>>(a);
      rowmlp_bwd16_k<2, 1, false><<<grid1, 256, 0, s>>>(f);
      merge_fix_partials_k<<<grid.x * B, 256, 0, s>>>(dTpart, dTfix, overflow, 4096);
    }
  } else if (mode >= 1 && words == 2) {
    if (!overflow) return HITADV_E_ARG;
    if (stage == 0) {
      rowmlp_bwd16_k<0, 2, false><<<grid, 256, 0, s>>>(a);
      rowmlp_bwd16_k<0, 2, true><<<grid, 256, 0, s>>>(a);
    } else if (stage == 1) {
      rowmlp_bwd16_k<1, 2, false><<<grid, 256, 0, s>>>(a);
      rowmlp_bwd16_k<1, 2, true><<<grid, 256, 0, s>>>(a);
    } else {
      rowmlp_bwd16_k<2, 2, false><<<grid, 256, 0, s>>>(a);
      rowmlp_bwd16_k<2, 2, true><<<grid, 256, 0, s>>>(a);
    }
  } else if (mode >= 1) {
    if (stage == 0) rowmlp_bwd16_k<0, 1, false><<<grid, 256, 0, s>>>(a);
    else if (stage == 1) rowmlp_bwd16_k<1, 1, false><<<grid, 256, 0, s>>>(a);
    else rowmlp_bwd16_k<2, 1, false><<<grid, 256, 0, s>>>(a);
  } else if (stage == 0) rowmlp_bwd_k<0><<<grid, 256, 0, s>>>(a);
  else if (stage == 1) rowmlp_bwd_k<1><<<grid, 256, 0, s>>>(a);
  else rowmlp_bwd_k<2><<<grid, 256, 0, s>>>(a);
  HITADV_LAUNCH_CHECK();
  return 0;
}

#ifdef HITADV_STAMPS
extern "C" int hitadv_debug_v3_stamps(unsigned long long *host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_v3_stamp), sizeof(unsigned long long) * (n < 3 * 1024 * 8 ? n : 3 * 1024 * 8));
}
#endif

extern "C" int hitadv_sum_partials(const float *part, const float *extra, int B, int T, int M, float *out,
                                   void *stream) {
  HITADV_ABLATE_RETURN("fc");
  if (!part || !out || B <= 0 || T <= 0 || M <= 0) return HITADV_E_ARG;
  const long long total = (long long)B * M;
  sum_partials_k<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(part, extra, T, M, out, total);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t hitadv_fc_layer_scratch_floats(int B, int K, int NOUT) {
  if (B <= 0 || K <= 0 || NOUT <= 0) return 0;
  int chunk, KS, tiles;
  fc_split(B, K, NOUT, &chunk, &KS, &tiles);
  return (int64_t)FC_TICKETS + (KS > 1 ? (int64_t)KS * tiles * 1024 : 0);
}

extern "C" int hitadv_fc_layer(const float *in, const float *mask, const float *Wt, const float *bias, int B, int K,
                               int NOUT, int relu, float *out, float *scratch, void *stream) {
  HITADV_ABLATE_RETURN("fc");
  if (!in || !Wt || !out || !scratch || B <= 0 || K <= 0 || NOUT <= 0) return HITADV_E_ARG;
  if ((K & 3) == 0 && (((uintptr_t)in | (uintptr_t)mask) & 15)) return HITADV_E_ARG;
  static const int wide = [] { const char *e = getenv("HITADV_FC_WIDE"); return e ? atoi(e) : 1; }();  // 0: tuning / A-B only
  // K = 256 and 512 (tools/fc_split_probe.py, B = 32: 4.3-4.6 instead of 5.7-5.9 us, 5.6 instead of 6.3-6.7 us); at K = 1024 a
  // block of sixteen waves is left with 64 registers per lane, its weight loads end up between the MFMAs, and the split
  // form wins (9.0 vs 7.4 us)
  if (wide && (K == 512 || K == 256)) {
    if (K == 512) return launch_fc_wide<8, 0>(in, mask, Wt, bias, B, K, NOUT, relu, out, FcPre{}, (hipStream_t)stream);
    return launch_fc_wide<4, 0>(in, mask, Wt, bias, B, K, NOUT, relu, out, FcPre{}, (hipStream_t)stream);
  }
  int chunk, KS, tiles;
  fc_split(B, K, NOUT, &chunk, &KS, &tiles);
  if (tiles > FC_TICKETS) return HITADV_E_ARG;
  dim3 grid((NOUT + 31) / 32, (B + 31) / 32, KS);
  fc_layer_k<0><<<grid, 256, 0, (hipStream_t)stream>>>(in, mask, Wt, bias, B, K, NOUT, relu, chunk, out,
                                                           scratch + FC_TICKETS, reinterpret_cast<int *>(scratch), FcPre{});
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_fc_layer_pre(const float *pre, int T, int J, const float *Wpre, const float *mask, const float *Wt,
                                   const float *bias, int B, int K, int NOUT, int relu, float *out, float *scratch,
                                   void *stream) {
  HITADV_ABLATE_RETURN("fc");
  if (!pre || !Wpre || !Wt || !out || !scratch || B <= 0 || K <= 0 || NOUT <= 0 || T <= 0 || J <= 0 || J > 64)
    return HITADV_E_ARG;
  int chunk, KS, tiles;
  fc_split(B, K, NOUT, &chunk, &KS, &tiles);
  if (tiles > FC_TICKETS) return HITADV_E_ARG;
  dim3 grid((NOUT + 31) / 32, (B + 31) / 32, KS);
  const FcPre pp{pre, Wpre, T, J};
  static const int wide = [] { const char *e = getenv("HITADV_FC_WIDE"); return e ? atoi(e) : 1; }();  // 0: tuning / A-B only
  if (wide && K == 256) {  // the stacks' case: four waves, one 64-deep slice each, no chunk loop
    if (J <= 16) return launch_fc_wide<4, 8>(nullptr, mask, Wt, bias, B, K, NOUT, relu, out, pp, (hipStream_t)stream);
    return launch_fc_wide<4, 32>(nullptr, mask, Wt, bias, B, K, NOUT, relu, out, pp, (hipStream_t)stream);
  }
  int *tk = reinterpret_cast<int *>(scratch);
  if (J <= 16)
    fc_layer_k<8><<<grid, 256, 0, (hipStream_t)stream>>>(nullptr, mask, Wt, bias, B, K, NOUT, relu, chunk, out,
                                                         scratch + FC_TICKETS, tk, pp);
  else
    fc_layer_k<32><<<grid, 256, 0, (hipStream_t)stream>>>(nullptr, mask, Wt, bias, B, K, NOUT, relu, chunk, out,
                                                          scratch + FC_TICKETS, tk, pp);
  HITADV_LAUNCH_CHECK();
  return 0;
}
