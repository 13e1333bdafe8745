// fp32-accurate GEMMs on the fp16 matrix cores for the victims' wide 1x1 convolutions, and DGCNN's embedding layer fused
// with its activation and poolings (model/dgcnn_cls.py:70-72,101-106: conv5 -> bn5 -> LeakyReLU(0.2) -> adaptive_max_pool1d
// | adaptive_avg_pool1d -> cat) -- forward and input gradient.
//
// Arithmetic: V1's fp16x2 scheme (csrc/victim_bf3.hip): every operand is two fp16 pieces, a = a1 + 2^-11 a2 (+ <= 2^-24 |a|),
// three exact fp16 x fp16 products per useful one accumulate in fp32 (v_mfma_f32_16x16x32_f16) into two accumulator sets
// (one per power of two), joined once by fmaf.  Error against float64: below the library's f32 GEMM (tools/split_gemm_probe.py:
// 3.6e-7 against 1.0e-6 of the scale at K = 512), at 1.6x its speed at K >= 256 (172 against 272 us for 32768 x 512 x 1024:
// hipBLASLt's f32 GEMM runs at the f32 MFMA rate, this kernel executes three times the flop at the rate hipBLASLt's own fp16
// kernel reaches at this K).  The kernel is bound by operand delivery, not by the three products (profiles/r03_g16_ablation.json).
// The same core carries the last shared layer + max over the neighbours of a sample-and-group block for the widths
// csrc/group_mlp.hip does not cover (GroupMaxEpi / GroupBwdA).
//
//   gemm_f16x2_k<AProd, Epi>   C-tile 256 x 128 per block of 8 waves (4 x 2, a wave owns 64 x 64 = 16 MFMA tiles x 2 accumulator
//                              sets = 128 VGPRs), K in steps of 32 (one MFMA slice) through a two-stage LDS ring: the A operand
//                              is PRODUCED by a functor (plain rows; rows gated by a ReLU mask; DGCNN's pooled gradient rebuilt
//                              from a sign bit mask and the pooled gradients) and split into pieces on the way into LDS, the B
//                              operand are weight pieces split once per weight ([2][N][K] fp16).  One barrier per K step.
//   LDS rows are dense (64 bytes = four 16-byte chunks) with the chunks of row r stored at position c ^ g[(r >> 2) & 3],
//   g = {0, 2, 3, 1}: a ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15, 20-27}, ... (MI355X_MICROARCH.md,
//   LDS) -- i.e. the rows 0-3 and 12-15 of a 16-row fragment at one chunk and the rows 4-11 at the next; with this swizzle each
//   group touches the sixteen 16-byte slots of the 256-byte bank row once.  Unswizzled 80-byte rows: 2-way conflicts.
//   Block order is XCD-aware: the column blocks that stream the same A rows are neighbours on one XCD (shared L2).
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"
#include "hitadv.h"

namespace hitadv {

typedef float f32x4m __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8m __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4m __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f16x8m as_f16x8m(uint4 u) { return __builtin_bit_cast(f16x8m, u); }

constexpr int G16_BM = 256, G16_BN = 128, G16_KS = 32, G16_RS = 64;
constexpr int G16_APIECE = G16_BM * G16_RS;
constexpr int G16_BPIECE = G16_BN * G16_RS;
constexpr int G16_STAGE = 2 * G16_APIECE + 2 * G16_BPIECE;  // 49,152 bytes; two stages
constexpr float G16_SCALE = 2048.f;
// byte offset of 16-byte chunk c of LDS row r (see the header)
__device__ __forceinline__ int g16_off(int r, int c) { return r * G16_RS + 16 * (c ^ ((0x78 >> (2 * ((r >> 2) & 3))) & 3)); }
#ifndef HITADV_G16_WR
#define HITADV_G16_WR 4
#endif

// eight values -> their fp16 pieces (csrc/pointnet.hip::split8v with this file's scale: hi pieces two per v_cvt_pk_f16_f32, each lo
// piece one v_fma_mix on the packed hi piece read in place; the bits of stash_as() below)
__device__ __forceinline__ void split8v_g16(const float (&v)[8], uint4 &hi, uint4 &lo) {
  uint32_t H[4], L[4];
  const float nsc = -G16_SCALE;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float s0 = v[2 * p] * G16_SCALE, s1 = v[2 * p + 1] * G16_SCALE;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(H[p]) : "v"(v[2 * p]), "v"(v[2 * p + 1]));
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(L[p]) : "v"(H[p]), "s"(nsc), "v"(s0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(L[p]) : "v"(H[p]), "s"(nsc), "v"(s1));
  }
  hi = make_uint4(H[0], H[1], H[2], H[3]);
  lo = make_uint4(L[0], L[1], L[2], L[3]);
}

struct G16Tile {
  int b;          // cloud (plain GEMM: 0)
  int p0;         // first point of the block within the cloud
  long long row0; // its global row
  int rows;       // valid rows in the block (<= 256)
  int rb;         // row-block index (b * chunks + chunk)
  int col0;       // first output column
};

// ---------------------------------------------------------------------------------------------- A producers
// fetch() only issues global loads (clamped, never conditional); values() turns the registers into the four floats of slot u.
// Slot u of thread t: row rr = (t >> 3) + 64 u of the block, k = k0 + 4 (t & 7) .. + 3.
template <bool GATED, int U>
struct PlainA {
  static constexpr int RSTEP = G16_BM / U;
  const float *X, *mask;
  int K;
  struct Regs {
    float4 v[U];
    float4 m[GATED ? U : 1];
  };
  __device__ __forceinline__ void fetch(Regs &r, const G16Tile &t, int k0, int tid) const {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rr = min((tid >> 3) + RSTEP * u, t.rows - 1);
      const size_t o = (size_t)(t.row0 + rr) * K + k0 + 4 * (tid & 7);
      r.v[u] = *reinterpret_cast<const float4 *>(X + o);
      if constexpr (GATED) r.m[u] = *reinterpret_cast<const float4 *>(mask + o);
    }
  }
  template <bool FULL>  // FULL: every row of the block exists (a type, not a test: see stash())
  __device__ __forceinline__ void values(const Regs &r, const G16Tile &t, int u, int tid, float (&o)[4]) const {
    o[0] = r.v[u].x, o[1] = r.v[u].y, o[2] = r.v[u].z, o[3] = r.v[u].w;
    if constexpr (GATED) {
      o[0] = r.m[u].x > 0.f ? o[0] : 0.f;
      o[1] = r.m[u].y > 0.f ? o[1] : 0.f;
      o[2] = r.m[u].z > 0.f ? o[2] : 0.f;
      o[3] = r.m[u].w > 0.f ? o[3] : 0.f;
    }
    if constexpr (!FULL) {
      const bool in = (tid >> 3) + RSTEP * u < t.rows;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = in ? o[j] : 0.f;
    }
  }
};

// d z[p,c] of  out = [max_p h | mean_p h],  h = lrelu(z):  s(z[p,c]) (gmean[b,c] / N + [arg[b,c] == p] gmax[b,c]),
// s = 1 where the forward pass saw z > 0 (its bit mask), the slope elsewhere.  K = C (the layer's output channels).
template <int U>
struct PoolBwdA {
  static constexpr int RSTEP = G16_BM / U;
  const uint32_t *bits;  // [rows][C / 32]
  const float *gout;     // [B][2 C]: d out, max half then mean half
  const int32_t *arg;    // [B][C]
  int C;
  float slope, inv_n;
  struct Regs {
    float4 gx, gm;
    int4 ar;
    uint32_t w[U];
  };
  __device__ __forceinline__ void fetch(Regs &r, const G16Tile &t, int k0, int tid) const {
    const int c = k0 + 4 * (tid & 7);
    r.gx = *reinterpret_cast<const float4 *>(gout + (size_t)t.b * 2 * C + c);
    r.gm = *reinterpret_cast<const float4 *>(gout + (size_t)t.b * 2 * C + C + c);
    r.ar = *reinterpret_cast<const int4 *>(arg + (size_t)t.b * C + c);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rr = min((tid >> 3) + RSTEP * u, t.rows - 1);
      r.w[u] = bits[(size_t)(t.row0 + rr) * (C / 32) + (k0 >> 5)];
    }
  }
  template <bool FULL>
  __device__ __forceinline__ void values(const Regs &r, const G16Tile &t, int u, int tid, float (&o)[4]) const {
    const int rr = (tid >> 3) + RSTEP * u;
    const bool in = FULL || rr < t.rows;
    const int p = t.p0 + rr;
    const uint32_t w = r.w[u] >> (4 * (tid & 7));
    const float gx[4] = {r.gx.x, r.gx.y, r.gx.z, r.gx.w}, gm[4] = {r.gm.x, r.gm.y, r.gm.z, r.gm.w};
    const int ar[4] = {r.ar.x, r.ar.y, r.ar.z, r.ar.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g = gm[j] * inv_n + (ar[j] == p ? gx[j] : 0.f);
      const float s = ((w >> j) & 1u) ? 1.f : slope;
      o[j] = in ? s * g : 0.f;
    }
  }
};

// d z[p,c] of  out[g,c] = relu(max_j z[g NS + j, c] + b[c]):  dm[g,c] where arg[g,c] == j (dm = d out gated by out > 0, [G,C]).
template <int U, int NS>
struct GroupBwdA {
  static constexpr int RSTEP = G16_BM / U;
  const float *dm;      // [G][C]
  const int32_t *arg;   // [G][C]
  int C;
  struct Regs {
    float4 d[U];
    int4 a[U];
  };
  __device__ __forceinline__ void fetch(Regs &r, const G16Tile &t, int k0, int tid) const {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long row = t.row0 + min((tid >> 3) + RSTEP * u, t.rows - 1);
      const size_t o = (size_t)(row / NS) * C + k0 + 4 * (tid & 7);
      r.d[u] = *reinterpret_cast<const float4 *>(dm + o);
      r.a[u] = *reinterpret_cast<const int4 *>(arg + o);
    }
  }
  template <bool FULL>
  __device__ __forceinline__ void values(const Regs &r, const G16Tile &t, int u, int tid, float (&o)[4]) const {
    const int j = (int)((t.row0 + (tid >> 3) + RSTEP * u) % NS);
    o[0] = r.a[u].x == j ? r.d[u].x : 0.f;
    o[1] = r.a[u].y == j ? r.d[u].y : 0.f;
    o[2] = r.a[u].z == j ? r.d[u].z : 0.f;
    o[3] = r.a[u].w == j ? r.d[u].w : 0.f;
  }
};

// ---------------------------------------------------------------------------------------------- epilogues
// A wave's accumulators: acc[rt][ct][i] = row 16 RT wr + 16 rt + 4 (lane / 16) + i, column 64 wc + 16 ct + lane % 16 of the block
// (RT row tiles per wave: 4 with eight waves per block, 8 with four).
struct PlainEpi {
  float *C;
  const float *bias;
  int N, relu;
  const float *gate = nullptr;  // [M,N] or NULL: the result is zeroed where gate <= 0 (the ReLU backward of the layer in front)
  template <int RT>
  __device__ __forceinline__ void operator()(const G16Tile &t, f32x4m (&acc)[RT][4], int wr, int wc, int lane, char *) const {
    const int l16 = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int col = t.col0 + 64 * wc + 16 * ct + l16;
      const float bv = bias != nullptr ? bias[col] : 0.f;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rr = 16 * RT * wr + 16 * rt + 4 * g4 + i;
          float v = acc[rt][ct][i] + bv;
          v = relu ? fmaxf(v, 0.f) : v;
          if (rr < t.rows) {
            const size_t o = (size_t)(t.row0 + rr) * N + col;
            if (gate != nullptr) v = gate[o] > 0.f ? v : 0.f;
            C[o] = v;
          }
        }
    }
  }
};

// out[g,c] = relu(max_j z[g NS + j, c] + bias[c]), arg[g,c] = the lowest such j: a wave's 64 rows hold 64 / NS whole groups, so
// the reduction stays inside the wave (rows of a column: 16 per lane, four lane groups).
template <int NS>
struct GroupMaxEpi {
  const float *bias;
  float *out;
  int32_t *arg;
  int C;
  template <int RT>
  __device__ __forceinline__ void operator()(const G16Tile &t, f32x4m (&acc)[RT][4], int wr, int wc, int lane, char *) const {
    static_assert(RT == 4 && (NS == 32 || NS == 64), "a wave tile of 64 rows, groups of 32 or 64 rows");
    constexpr int TPG = NS / 16;  // 16-row tiles per group
    const int l16 = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int col = t.col0 + 64 * wc + 16 * ct + l16;
      const float bv = bias[col];
#pragma unroll
      for (int h = 0; h < 64 / NS; ++h) {
        float best = -__builtin_inff();
        int bi = 0;
#pragma unroll
        for (int r = 0; r < TPG; ++r)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float v = acc[h * TPG + r][ct][i];
            const bool gt = v > best;  // a lane's rows ascend: the first maximum stays
            best = gt ? v : best;
            bi = gt ? 16 * r + 4 * g4 + i : bi;
          }
#pragma unroll
        for (int m = 16; m <= 32; m <<= 1) {
          const float ov = __shfl_xor(best, m, 64);
          const int oi = __shfl_xor(bi, m, 64);
          if (ov > best || (ov == best && oi < bi)) best = ov, bi = oi;
        }
        const int rr = 64 * wr + h * NS;
        if (g4 == 0 && rr < t.rows) {
          const size_t o = (size_t)((t.row0 + rr) / NS) * C + col;
          out[o] = fmaxf(best + bv, 0.f);
          arg[o] = bi;
        }
      }
    }
  }
};

// z = acc + bias;  bits: z > 0;  h = lrelu(z);  per column the block's max (lowest row on ties) and sum of h -> partials.
struct PoolEpi {
  const float *bias;
  uint16_t *bits;   // [rows][C / 16] (= uint32 [rows][C / 32], little endian)
  float *pmax, *psum;
  int32_t *parg;    // [row blocks][C]
  int C;
  float slope;
  template <int RT>
  __device__ __forceinline__ void operator()(const G16Tile &t, f32x4m (&acc)[RT][4], int wr, int wc, int lane, char *lds) const {
    constexpr int WR = 16 / RT;
    const int l16 = lane & 15, g4 = lane >> 4;
    float *rmax = reinterpret_cast<float *>(lds);            // [WR][128]
    float *rsum = rmax + WR * G16_BN;
    int *rarg = reinterpret_cast<int *>(rsum + WR * G16_BN);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int colb = 64 * wc + 16 * ct + l16;
      const float bv = bias[t.col0 + colb];
      float best = -__builtin_inff(), sum = 0.f;
      int bi = 0x7fffffff;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rr = 16 * RT * wr + 16 * rt + 4 * g4 + i;
          const float z = acc[rt][ct][i] + bv;
          const bool pos = z > 0.f;
          const unsigned long long bal = __builtin_amdgcn_ballot_w64(pos);
          if (l16 == 0 && rr < t.rows)
            bits[(size_t)(t.row0 + rr) * (C / 16) + ((t.col0 + 64 * wc + 16 * ct) >> 4)] = (uint16_t)(bal >> (16 * g4));
          const float h = pos ? z : slope * z;
          const bool in = rr < t.rows;
          sum += in ? h : 0.f;
          if (in && h > best) best = h, bi = t.p0 + rr;
        }
      // the four lane groups hold interleaved rows of the same column: (value, lowest row) and the sum in a fixed order
#pragma unroll
      for (int m = 16; m <= 32; m <<= 1) {
        const float ov = __shfl_xor(best, m, 64), os = __shfl_xor(sum, m, 64);
        const int oi = __shfl_xor(bi, m, 64);
        if (ov > best || (ov == best && oi < bi)) best = ov, bi = oi;
        sum = (lane & m) ? os + sum : sum + os;
      }
      if (g4 == 0) rmax[wr * G16_BN + colb] = best, rsum[wr * G16_BN + colb] = sum, rarg[wr * G16_BN + colb] = bi;
    }
    __syncthreads();
    if (wr == 0 && g4 == 0) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int colb = 64 * wc + 16 * ct + l16;
        float best = rmax[colb], sum = rsum[colb];
        int bi = rarg[colb];
#pragma unroll
        for (int w = 1; w < WR; ++w) {  // ascending rows: strict > keeps the lowest row
          const float ov = rmax[w * G16_BN + colb];
          if (ov > best) best = ov, bi = rarg[w * G16_BN + colb];
          sum += rsum[w * G16_BN + colb];
        }
        const size_t o = (size_t)t.rb * C + t.col0 + colb;
        pmax[o] = best, psum[o] = sum, parg[o] = bi == 0x7fffffff ? t.p0 : bi;  // (all-NaN column: a valid row, the range flag is up)
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------- the kernel
// ABL != 0: tuning builds only (tools/tune/g16_ablate.py: the kernel with one cost removed; results are garbage)
template <int WR, class AProd, class Epi, bool SYNC_EPI, int ABL = 0>
__global__ __launch_bounds__(WR * 128) void gemm_f16x2_k(AProd ap, const uint16_t *__restrict__ Wp, int npts, int chunks, int nrb,
                                                         int ncb, int N, int K, Epi epi, int *range_flag) {
  constexpr int NT = WR * 128;       // threads: WR x 2 waves
  constexpr int RT = 16 / WR;        // 16-row MFMA tiles per wave: 4 (64 x 64 per wave, two waves per SIMD) or 8 (128 x 64, one)
  constexpr int U = 2048 / NT;       // A slots (four consecutive k of one row) per thread and K step
  constexpr int V = 512 / NT;        // B chunks (eight consecutive k of one column, per piece) per thread and K step
  extern __shared__ __attribute__((aligned(16))) char sG16[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l16 = lane & 15, g4 = lane >> 4;
  G16Tile t;
  {
    int rb, cb;
    const int id = blockIdx.x;
    if ((nrb & 7) == 0) {
      const int xcd = id & 7, slot = id >> 3;
      cb = slot % ncb;
      rb = (slot / ncb) * 8 + xcd;
    } else {
      cb = id % ncb;
      rb = id / ncb;
    }
    t.rb = rb;
    t.b = rb / chunks;
    t.p0 = (rb % chunks) * G16_BM;
    t.row0 = (long long)t.b * npts + t.p0;
    t.rows = min(G16_BM, npts - t.p0);
    t.col0 = cb * G16_BN;
  }
  const int nk = K / G16_KS;

  // Operands travel global -> registers -> (split) -> LDS: the loads of K step s + 1 are issued at the top of step s and written
  // to the other LDS stage after the step's MFMAs.  (A second register set with the loads two steps ahead and the two waves of
  // a SIMD staggered, as in csrc/victim_bf3.hip, was measured: 177 -> 188 us at 32768 x 512 x 1024, it spills.)
  struct Set {
    typename AProd::Regs a;
    uint4 b[V][2];
  };
  Set s0;
  // B: 128 columns x 4 sixteen-byte chunks per piece; chunk e = tid + NT v -> column e >> 2, chunk e & 3
  const uint16_t *bsrc = Wp + (size_t)(t.col0 + (tid >> 2)) * K + 8 * (tid & 3);
  auto fetch = [&](Set &st, int ks) {
    if constexpr (ABL == 2) {
      if (ks > 0) return;  // no global loads after the first step
    }
    ap.fetch(st.a, t, ks * G16_KS, tid);
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const uint16_t *bp = bsrc + (size_t)(NT / 4) * v * K + ks * G16_KS;
      st.b[v][0] = *reinterpret_cast<const uint4 *>(bp);
      st.b[v][1] = *reinterpret_cast<const uint4 *>(bp + (size_t)N * K);
    }
  };
  PieceWatch big;
  // A block whose 256 rows all exist (every block but a cloud's last) takes the FULL form: the compiler turns a run-time bounds
  // test into a select per VALUE (csrc/victim_bf3.hip found the same), so the two forms are separate instantiations.
  auto stash_as = [&](const Set &st, int stage, auto full_c) {
    char *base = sG16 + (size_t)stage * G16_STAGE;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float a[4];
      ap.template values<decltype(full_c)::value>(st.a, t, u, tid, a);
      uint2 h1, h2;
      if constexpr (ABL == 1) {  // no conversions: raw bits
        h1 = make_uint2(__float_as_uint(a[0]), __float_as_uint(a[1]));
        h2 = make_uint2(__float_as_uint(a[2]), __float_as_uint(a[3]));
      } else {
        // hi pieces two per v_cvt_pk_f16_f32; each lo piece ONE fused multiply-add on the packed hi piece read in place,
        // hi (-2048) + 2048 a = 2048 (a - hi) exactly before its single rounding to fp16 (csrc/pointnet.hip::split8v: the same
        // bits as convert / convert back / subtract / scale / convert, at half the instructions -- and on this part vector
        // instructions are not hidden behind the matrix pipe).  The range watch looks at the hi pieces, two per instruction:
        // the split breaks down exactly when a hi piece is an infinity or a NaN (|a| >= 65520).
        const float nsc = -G16_SCALE;
        const float s0 = a[0] * G16_SCALE, s1 = a[1] * G16_SCALE, s2 = a[2] * G16_SCALE, s3 = a[3] * G16_SCALE;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h1.x) : "v"(a[0]), "v"(a[1]));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h1.y) : "v"(a[2]), "v"(a[3]));
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(h2.x) : "v"(h1.x), "s"(nsc), "v"(s0));
        asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(h2.x) : "v"(h1.x), "s"(nsc), "v"(s1));
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(h2.y) : "v"(h1.y), "s"(nsc), "v"(s2));
        asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(h2.y) : "v"(h1.y), "s"(nsc), "v"(s3));
        big.see_f16x2(h1.x);
        big.see_f16x2(h1.y);
      }
      char *dst = base + g16_off((tid >> 3) + (NT / 8) * u, (tid & 7) >> 1) + 8 * (tid & 1);
      *reinterpret_cast<uint2 *>(dst) = h1;
      *reinterpret_cast<uint2 *>(dst + G16_APIECE) = h2;
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
      char *bd = base + 2 * G16_APIECE + g16_off((tid >> 2) + (NT / 4) * v, tid & 3);
      *reinterpret_cast<uint4 *>(bd) = st.b[v][0];
      *reinterpret_cast<uint4 *>(bd + G16_BPIECE) = st.b[v][1];
    }
  };
  const bool full = t.rows == G16_BM;  // block-uniform
  auto stash = [&](const Set &st, int stage) {
    if (full) stash_as(st, stage, std::true_type{});
    else stash_as(st, stage, std::false_type{});
  };

  f32x4m acc[RT][4], accl[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = f32x4m{0.f, 0.f, 0.f, 0.f}, accl[rt][ct] = f32x4m{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int ks) {
    if constexpr (ABL == 3) return;  // no LDS reads, no MFMAs
    const char *base = sG16 + (size_t)(ks & 1) * G16_STAGE;
    // rows 16 x + l16: (row >> 2) & 3 = (l16 >> 2) & 3 for every 16-row tile, so one swizzled offset serves all of them
    const char *ab = base + 16 * RT * wr * G16_RS + g16_off(l16, g4);
    const char *bb = base + 2 * G16_APIECE + 64 * wc * G16_RS + g16_off(l16, g4);
    uint4 fb[4][2];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      fb[ct][0] = *reinterpret_cast<const uint4 *>(bb + 16 * ct * G16_RS);
      fb[ct][1] = *reinterpret_cast<const uint4 *>(bb + 16 * ct * G16_RS + G16_BPIECE);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const f16x8m ahi = as_f16x8m(*reinterpret_cast<const uint4 *>(ab + 16 * rt * G16_RS));
      const f16x8m alo = as_f16x8m(*reinterpret_cast<const uint4 *>(ab + 16 * rt * G16_RS + G16_APIECE));
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        if constexpr (ABL == 4) {  // one product instead of three
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, as_f16x8m(fb[ct][0]), acc[rt][ct], 0, 0, 0);
          accl[rt][ct][0] += (float)alo[0] + (float)as_f16x8m(fb[ct][1])[0];
          continue;
        }
        accl[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, as_f16x8m(fb[ct][0]), accl[rt][ct], 0, 0, 0);
        accl[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, as_f16x8m(fb[ct][1]), accl[rt][ct], 0, 0, 0);
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, as_f16x8m(fb[ct][0]), acc[rt][ct], 0, 0, 0);
      }
    }
  };
  fetch(s0, 0);
  stash(s0, 0);
  for (int ks = 0; ks < nk; ++ks) {
    __syncthreads();  // stage ks & 1 (written during step ks - 1) is complete; the other stage is no longer being read
    // requested AFTER the barrier: __syncthreads() waits for every load in flight, and these have the whole matrix phase to land
    if (ks + 1 < nk) fetch(s0, ks + 1);
    compute(ks);
    if (ks + 1 < nk) stash(s0, (ks + 1) & 1);
  }
  if (big.beyond_fp16() && range_flag != nullptr) *range_flag = 1;  // beyond fp16 (or NaN): the caller refuses the result
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[rt][ct][i] = fmaf(accl[rt][ct][i], 1.f / G16_SCALE, acc[rt][ct][i]);
  if constexpr (SYNC_EPI) __syncthreads();  // the epilogue reuses the LDS ring
  epi.template operator()<RT>(t, acc, wr, wc, lane, sG16);
}

// ---------------------------------------------------------------------------------------------- the ring kernel (round 5)
// gemm_f16x2_k above is bound by operand delivery, not by its products (docs/kernels/other_victims.md: with no LDS reads and no
// MFMAs it still takes 98 of 159 us): operands travel global -> registers -> split -> LDS one K step ahead, every wave's staging
// sits between every wave's products, and a second register set spills.  Here NOTHING is staged through registers:
//   * both operands go global -> LDS by global_load_lds_dwordx4 (16 bytes per lane, a wave fills 1 KB of LDS in lane order; the
//     swizzle is applied on the GLOBAL side: lane p fetches the chunk that belongs at LDS position p), into a THREE-stage ring,
//     so the loads of K step s + 2 are in flight while step s multiplies; one s_barrier and one s_waitcnt vmcnt per step
//   * a plain fp32 A operand lands in LDS as fp32 rows (128 bytes per row and step) and is split into its two fp16 pieces by the
//     wave that reads the fragment (split8v: the same bits as the staging split; the two waves of a row block both split their
//     rows: 28 vector instructions per 16 x 32 fragment against 48 MFMAs of 16 cycles)
//   * the products, their order and the accumulators are gemm_f16x2_k's: the results are the same bits.
// LDS-DMA and its waits are inline asm: the compiler's counter tracking would put s_waitcnt vmcnt(0) in front of every LDS read
// that may alias a DMA write, which is exactly the wait the ring exists to avoid.
constexpr int GR_AROW = 128;                        // bytes per fp32 A row and K step
constexpr int GR_A = G16_BM * GR_AROW;              // 32 KB
constexpr int GR_STAGE = GR_A + 2 * G16_BPIECE;     // 48 KB
constexpr int GR_STAGES = 3;
// 16-byte chunk c of fp32 row r sits at position c ^ s(r), s = bit 1 of r | bit 2 of r << 2: the sixteen lanes a ds_read_b128 serves
// together (rows 0-3, 12-15 at chunk 2 g, rows 4-11 at chunk 2 g + 2) then cover the sixteen 16-byte slots of a 256-byte bank
// row once (rows alternate between its halves, s takes {0, 1, 4, 5} on each parity of either row set)
__device__ __forceinline__ int gr_swz(int r) { return ((r >> 1) & 1) | (((r >> 2) & 1) << 2); }

// (m0 cannot usefully be declared clobbered: clang lists it among the RESERVED registers -- the clobber is accepted with
// "-Winline-asm: clobber list contains reserved registers" and has no effect (tried in round 6 for ADVICE r05: identical ISA, one warning per
// instantiation).  Nothing else in this kernel uses m0: gfx9 LDS instructions do not read it, and the kernel has no relative-indexed
// register move, LDS-direct load or s_sendmsg; tests/test_isa_guards.py checks that the ring kernel's only writes of m0 are these.)
__device__ __forceinline__ void gr_dma16(uint32_t lds_at, uint32_t voff, const void *sbase) {
  const unsigned long long a = (unsigned long long)(uintptr_t)sbase;
  const unsigned long long u = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(__builtin_amdgcn_readfirstlane((int)lds_at)), "v"(voff),
               "s"(u)
               : "memory");
}

// ABL != 0: tuning builds only (tools/tune/g16_ablate.py; results are garbage): 11 no MFMAs, 12 no split (raw bits), 13 no LDS reads,
// split or MFMAs (the DMA ring and its barriers alone), 14 no DMA after the prologue, 15 one product instead of three
template <class Epi, bool SYNC_EPI, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_f16x2_ring_k(const float *__restrict__ X, const uint16_t *__restrict__ Wp, int npts,
                                                         int chunks, int nrb, int ncb, int N, int K, Epi epi, int *range_flag) {
  constexpr int RT = 4;
  extern __shared__ __attribute__((aligned(16))) char sG16[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l16 = lane & 15, g4 = lane >> 4;
  G16Tile t;
  {
    int rb, cb;
    const int id = blockIdx.x;
    if ((nrb & 7) == 0) {
      const int xcd = id & 7, slot = id >> 3;
      cb = slot % ncb;
      rb = (slot / ncb) * 8 + xcd;
    } else {
      cb = id % ncb;
      rb = id / ncb;
    }
    t.rb = rb;
    t.b = rb / chunks;
    t.p0 = (rb % chunks) * G16_BM;
    t.row0 = (long long)t.b * npts + t.p0;
    t.rows = min(G16_BM, npts - t.p0);
    t.col0 = cb * G16_BN;
  }
  const int nk = K / G16_KS;
  typedef __attribute__((address_space(3))) char lds_char;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char *)sG16;
  // A: wave-load i of this wave fills rows 8 (4 wave + i) .. + 7 (1 KB): lane p -> row + (p >> 3), position p & 7
  uint32_t va[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 8 * (4 * wave + i) + (lane >> 3);
    va[i] = (uint32_t)min(r, t.rows - 1) * (uint32_t)K * 4u + 16u * (uint32_t)((lane & 7) ^ gr_swz(r));
  }
  // B: wave-load j fills piece j, columns 16 wave .. + 15 (1 KB): lane p -> column + (p >> 2), position p & 3 (g16_off's swizzle)
  const int bcol = 16 * wave + (lane >> 2);
  const uint32_t vb = (uint32_t)bcol * (uint32_t)K * 2u + 16u * (uint32_t)((lane & 3) ^ ((0x78 >> (2 * ((bcol >> 2) & 3))) & 3));
  const char *abase = reinterpret_cast<const char *>(X + (size_t)t.row0 * K);
  const char *bbase0 = reinterpret_cast<const char *>(Wp + (size_t)t.col0 * K);
  const char *bbase1 = bbase0 + (size_t)N * K * 2;
  auto issue = [&](int ks) {
    const uint32_t st = lds0 + (uint32_t)(ks % GR_STAGES) * GR_STAGE;
    const char *ak = abase + (size_t)ks * (G16_KS * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) gr_dma16(st + 1024u * (uint32_t)(4 * wave + i), va[i], ak);
    gr_dma16(st + GR_A + 1024u * (uint32_t)wave, vb, bbase0 + (size_t)ks * (G16_KS * 2));
    gr_dma16(st + GR_A + G16_BPIECE + 1024u * (uint32_t)wave, vb, bbase1 + (size_t)ks * (G16_KS * 2));
  };

  f32x4m acc[RT][4], accl[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = f32x4m{0.f, 0.f, 0.f, 0.f}, accl[rt][ct] = f32x4m{0.f, 0.f, 0.f, 0.f};
  PieceWatch big;
  // fragment rows 16 x + l16: bits 1, 2 (A) and (row >> 2) & 3 (B) of the row are l16's, one swizzled offset serves every tile
  const int aoff = l16 * GR_AROW + 16 * ((2 * g4) ^ gr_swz(l16)), aoff1 = l16 * GR_AROW + 16 * ((2 * g4 + 1) ^ gr_swz(l16));
  const int boff = g16_off(l16, g4);
  // The fragments of step s + 1 are read WHILE step s multiplies: a barrier aligns the waves, and with the reads at the top of a
  // step every wave waited out the LDS at the same time, then every wave split, then every wave multiplied (measured: delivery
  // alone 63 us, products 59, split 20, and 148 in all -- nothing overlapped).  So a step is: split the A rows read during the
  // previous step (raw fp32 -> pieces, in registers), wait + barrier for stage s + 1, request stage s + 3 into the buffer stage
  // s has just left, request the A rows of step s + 1 from LDS, then the products column tile by column tile, each tile's B
  // fragments re-read for step s + 1 as soon as its twelve products are issued.
  uint4 fb[4][2];
  float4 fa[RT][2];
  uint4 ah[RT], al[RT];
  auto frag_base = [&](int ks) { return sG16 + (size_t)(ks % GR_STAGES) * GR_STAGE; };
  auto read_a = [&](int ks) {
    if constexpr (ABL == 13) return;
    const char *ab = frag_base(ks) + 64 * wr * GR_AROW + aoff, *ab1 = frag_base(ks) + 64 * wr * GR_AROW + aoff1;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      fa[rt][0] = *reinterpret_cast<const float4 *>(ab + 16 * rt * GR_AROW);
      fa[rt][1] = *reinterpret_cast<const float4 *>(ab1 + 16 * rt * GR_AROW);
    }
  };
  auto read_b = [&](int ks, int ct) {
    if constexpr (ABL == 13) return;
    const char *bb = frag_base(ks) + GR_A + 64 * wc * G16_RS + boff;
    fb[ct][0] = *reinterpret_cast<const uint4 *>(bb + 16 * ct * G16_RS);
    fb[ct][1] = *reinterpret_cast<const uint4 *>(bb + 16 * ct * G16_RS + G16_BPIECE);
  };
  auto split_a = [&]() {
    if constexpr (ABL == 13) return;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const float4 v0 = fa[rt][0], v1 = fa[rt][1];
      const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      if constexpr (ABL == 12) {
        ah[rt] = __builtin_bit_cast(uint4, v0), al[rt] = __builtin_bit_cast(uint4, v1);
      } else {
        split8v_g16(v, ah[rt], al[rt]);
        big.see_f16x2(ah[rt].x), big.see_f16x2(ah[rt].y), big.see_f16x2(ah[rt].z), big.see_f16x2(ah[rt].w);
      }
    }
  };
  auto products = [&](int ct) {
    if constexpr (ABL == 13) return;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const f16x8m ahi = as_f16x8m(ah[rt]), alo = as_f16x8m(al[rt]);
      if constexpr (ABL == 11) {
        accl[rt][ct][0] += (float)alo[0] + (float)as_f16x8m(fb[ct][1])[0] + (float)ahi[1] + (float)as_f16x8m(fb[ct][0])[2];
        continue;
      }
      if constexpr (ABL == 15) {
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, as_f16x8m(fb[ct][0]), acc[rt][ct], 0, 0, 0);
        accl[rt][ct][0] += (float)alo[0] + (float)as_f16x8m(fb[ct][1])[0];
        continue;
      }
    }
    if constexpr (ABL == 11 || ABL == 15) return;
    // the two products into accl[rt][ct] are four MFMAs apart, not back to back (the second reads the first's result)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) accl[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8m(al[rt]), as_f16x8m(fb[ct][0]), accl[rt][ct], 0, 0, 0);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8m(ah[rt]), as_f16x8m(fb[ct][0]), acc[rt][ct], 0, 0, 0);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) accl[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8m(ah[rt]), as_f16x8m(fb[ct][1]), accl[rt][ct], 0, 0, 0);
  };
  issue(0);
  if (nk > 1) issue(1);
  if (nk > 2) issue(2);
  // stage 0: this wave's loads (the twelve newer ones may fly), then every wave's
  if (nk > 2)
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (nk > 1)
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_a(0);
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) read_b(0, ct);
  for (int ks = 0; ks < nk; ++ks) {
    split_a();  // (waits for the A rows of this step, requested during the previous one)
    __builtin_amdgcn_sched_barrier(0);
    const bool more = ks + 1 < nk;
    if (more) {
      // stage ks + 1: this wave's loads have landed (six newer ones, stage ks + 2's, may still fly) ...
      // (lgkmcnt(0): the B fragments of THIS step, read from stage ks during the previous one, are in registers before any wave may
      // request stage ks + 3 into that buffer -- split_a() above only waited for the A rows, which were requested first)
      if (ks + 2 < nk && ABL != 14)
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      // ... and after the barrier every wave's have; every wave has also taken stage ks out of LDS, whose buffer stage ks + 3 refills
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (ks + 3 < nk && ABL != 14) issue(ks + 3);
      read_a(ks + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      products(ct);
      __builtin_amdgcn_sched_barrier(0);
      if (more) read_b(ks + 1, ct);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (big.beyond_fp16() && range_flag != nullptr) *range_flag = 1;  // beyond fp16 (or NaN): the caller refuses the result
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[rt][ct][i] = fmaf(accl[rt][ct][i], 1.f / G16_SCALE, acc[rt][ct][i]);
  if constexpr (SYNC_EPI) __syncthreads();  // the epilogue reuses the LDS ring
  epi.template operator()<RT>(t, acc, wr, wc, lane, sG16);
}

// out[b] = [max over the cloud's row blocks | sum / n], arg = the row of the max (lowest on ties: blocks ascend)
__global__ void pool_merge_k(const float *__restrict__ pmax, const float *__restrict__ psum, const int32_t *__restrict__ parg,
                             int B, int chunks, int C, float inv_n, float *__restrict__ out, int32_t *__restrict__ arg) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * C) return;
  const int b = e / C, c = e % C;
  float best = -__builtin_inff(), sum = 0.f;
  int bi = 0;
  for (int k = 0; k < chunks; ++k) {
    const size_t o = (size_t)(b * chunks + k) * C + c;
    const float v = pmax[o];
    if (v > best) best = v, bi = parg[o];
    sum += psum[o];
  }
  out[(size_t)b * 2 * C + c] = best;
  out[(size_t)b * 2 * C + C + c] = sum * inv_n;
  arg[(size_t)b * C + c] = bi;
}

// W [N][K] fp32 -> pieces [2][N][K] fp16
__global__ void split_rows_f16x2_k(const float *__restrict__ W, uint16_t *__restrict__ Wp, long long total, int *range_flag) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const float a = W[e];
  const _Float16 h1 = (_Float16)a;
  const _Float16 h2 = (_Float16)((a - (float)h1) * G16_SCALE);
  Wp[e] = __builtin_bit_cast(uint16_t, h1);
  Wp[total + e] = __builtin_bit_cast(uint16_t, h2);
  if (!(fabsf(a) < 65504.f) && range_flag != nullptr) *range_flag = 1;
}

constexpr int G16_WR = HITADV_G16_WR;  // wave rows per block: 4 = eight waves of 64 x 64, 2 = four waves of 128 x 64
constexpr int G16_U = 2048 / (G16_WR * 128);

template <class AProd, class Epi, bool SYNC_EPI>
static int launch_gemm16(const AProd &ap, const uint16_t *Wp, int B, int npts, int N, int K, const Epi &epi, int32_t *range_flag,
                         hipStream_t s) {
  const int chunks = (npts + G16_BM - 1) / G16_BM, nrb = B * chunks, ncb = N / G16_BN;
  HITADV_RAISE_LDS((&gemm_f16x2_k<G16_WR, AProd, Epi, SYNC_EPI>), 2 * G16_STAGE);
  gemm_f16x2_k<G16_WR, AProd, Epi, SYNC_EPI><<<dim3((unsigned)(nrb * ncb)), G16_WR * 128, 2 * G16_STAGE, s>>>(
      ap, Wp, npts, chunks, nrb, ncb, N, K, epi, range_flag);
  HITADV_LAUNCH_CHECK();
  return 0;
}

static int g_g16_ring = [] { const char *e = getenv("HITADV_G16_RING"); return e && e[0] >= '0' && e[0] <= '2' ? e[0] - '0' : 1; }();  // plain fp32 A operands: 1 = gemm_f16x2_ring_k where it is faster (K >= 256), 0 = gemm_f16x2_k, 2 = the ring for every K >= 64 (tests)

template <class Epi, bool SYNC_EPI>
static int launch_gemm16_plain(const float *X, const uint16_t *Wp, int B, int npts, int N, int K, const Epi &epi, int32_t *range_flag,
                               hipStream_t s) {
  if (!g_g16_ring || K / G16_KS < 2 || (g_g16_ring == 1 && K < 256))
    return launch_gemm16<PlainA<false, G16_U>, Epi, SYNC_EPI>(PlainA<false, G16_U>{X, nullptr, K}, Wp, B, npts, N, K, epi, range_flag, s);
  const int chunks = (npts + G16_BM - 1) / G16_BM, nrb = B * chunks, ncb = N / G16_BN;
  HITADV_RAISE_LDS((&gemm_f16x2_ring_k<Epi, SYNC_EPI>), GR_STAGES * GR_STAGE);
  gemm_f16x2_ring_k<Epi, SYNC_EPI><<<dim3((unsigned)(nrb * ncb)), 512, GR_STAGES * GR_STAGE, s>>>(X, Wp, npts, chunks, nrb, ncb, N, K, epi,
                                                                                               range_flag);
  HITADV_LAUNCH_CHECK();
  return 0;
}

static bool g16_shape_ok(long long M, int N, int K) { return M > 0 && M < (1ll << 31) && N > 0 && K > 0 && (N % G16_BN) == 0 && (K % G16_KS) == 0; }

}  // namespace hitadv

using namespace hitadv;

extern "C" int hitadv_gemm_f16x2_supported(int N, int K) { return (N > 0 && K > 0 && (N % G16_BN) == 0 && (K % G16_KS) == 0) ? 1 : 0; }

extern "C" int hitadv_split_rows_f16x2(const float *W, int N, int K, uint16_t *Wp, int32_t *range_flag, void *stream) {
  if (!W || !Wp || N <= 0 || K <= 0) return HITADV_E_ARG;
  const long long total = (long long)N * K;
  split_rows_f16x2_k<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(W, Wp, total, range_flag);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_gemm_f16x2(const float *X, const float *mask, const uint16_t *Wp, const float *bias, int64_t M, int N, int K,
                                 int relu, float *C, int32_t *range_flag, void *stream) {
  if (!X || !Wp || !C || !g16_shape_ok(M, N, K) || ((uintptr_t)X & 15) || ((uintptr_t)Wp & 15) || (mask && ((uintptr_t)mask & 15)))
    return HITADV_E_ARG;
  const PlainEpi epi{C, bias, N, relu};
  if (mask != nullptr)
    return launch_gemm16<PlainA<true, G16_U>, PlainEpi, false>(PlainA<true, G16_U>{X, mask, K}, Wp, 1, (int)M, N, K, epi, range_flag, (hipStream_t)stream);
  return launch_gemm16_plain<PlainEpi, false>(X, Wp, 1, (int)M, N, K, epi, range_flag, (hipStream_t)stream);
}

extern "C" int hitadv_debug_g16_ring(int on) {
  const int old = g_g16_ring;
  if (on >= 0 && on <= 2) g_g16_ring = on;
  return old;
}

extern "C" int64_t hitadv_linear_lrelu_pool_scratch(int B, int npts, int C) {
  if (B <= 0 || npts <= 0 || C <= 0) return 0;
  return (int64_t)B * ((npts + G16_BM - 1) / G16_BM) * C;
}

extern "C" int hitadv_linear_lrelu_pool_fwd(const float *X, const uint16_t *Wp, const float *bias, int B, int npts, int Cin, int C,
                                            float slope, float *pmax, float *psum, int32_t *parg, uint32_t *bits, float *out,
                                            int32_t *arg, int32_t *range_flag, void *stream) {
  if (!X || !Wp || !bias || !pmax || !psum || !parg || !bits || !out || !arg || B <= 0 || npts <= 0 ||
      !g16_shape_ok((long long)B * npts, C, Cin) || ((uintptr_t)X & 15) || ((uintptr_t)Wp & 15))
    return HITADV_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const PoolEpi epi{bias, reinterpret_cast<uint16_t *>(bits), pmax, psum, parg, C, slope};
  const int rc = launch_gemm16_plain<PoolEpi, true>(X, Wp, B, npts, C, Cin, epi, range_flag, s);
  if (rc) return rc;
  const int chunks = (npts + G16_BM - 1) / G16_BM;
  pool_merge_k<<<(unsigned)((B * C + 255) / 256), 256, 0, s>>>(pmax, psum, parg, B, chunks, C, 1.f / (float)npts, out, arg);
  HITADV_LAUNCH_CHECK();
  return 0;
}

extern "C" int hitadv_linear_lrelu_pool_bwd(const float *gout, const int32_t *arg, const uint32_t *bits, const uint16_t *Wtp, int B,
                                            int npts, int Cin, int C, float slope, float *dX, int32_t *range_flag, void *stream) {
  if (!gout || !arg || !bits || !Wtp || !dX || B <= 0 || npts <= 0 || !g16_shape_ok((long long)B * npts, Cin, C) ||
      ((uintptr_t)gout & 15) || ((uintptr_t)arg & 15) || ((uintptr_t)Wtp & 15))
    return HITADV_E_ARG;
  const PoolBwdA<G16_U> ap{bits, gout, arg, C, slope, 1.f / (float)npts};
  const PlainEpi epi{dX, nullptr, Cin, 0};
  return launch_gemm16<PoolBwdA<G16_U>, PlainEpi, false>(ap, Wtp, B, npts, Cin, C, epi, range_flag, (hipStream_t)stream);
}

extern "C" int hitadv_group_linear_max_g16_supported(int Cin, int Cout, int ns) {
  return ((ns == 32 || ns == 64) && hitadv_gemm_f16x2_supported(Cout, Cin) && hitadv_gemm_f16x2_supported(Cin, Cout)) ? 1 : 0;
}

extern "C" int hitadv_group_linear_max_g16_fwd(const float *X, const uint16_t *Wp, const float *bias, int64_t G, int ns, int Cin,
                                               int Cout, float *out, int32_t *arg, int32_t *range_flag, void *stream) {
  if (!X || !Wp || !bias || !out || !arg || G <= 0 || !hitadv_group_linear_max_g16_supported(Cin, Cout, ns) ||
      !g16_shape_ok(G * ns, Cout, Cin) || ((uintptr_t)X & 15) || ((uintptr_t)Wp & 15))
    return HITADV_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (ns == 32) return launch_gemm16_plain<GroupMaxEpi<32>, false>(X, Wp, 1, (int)(G * ns), Cout, Cin, GroupMaxEpi<32>{bias, out, arg, Cout}, range_flag, s);
  return launch_gemm16_plain<GroupMaxEpi<64>, false>(X, Wp, 1, (int)(G * ns), Cout, Cin, GroupMaxEpi<64>{bias, out, arg, Cout}, range_flag, s);
}

extern "C" int hitadv_group_linear_max_g16_bwd(const float *dm, const int32_t *arg, const uint16_t *Wtp, int64_t G, int ns, int Cin,
                                               int Cout, const float *xmask, float *dX, int32_t *range_flag, void *stream) {
  if (!dm || !arg || !Wtp || !dX || G <= 0 || !hitadv_group_linear_max_g16_supported(Cin, Cout, ns) ||
      !g16_shape_ok(G * ns, Cin, Cout) || ((uintptr_t)dm & 15) || ((uintptr_t)arg & 15) || ((uintptr_t)Wtp & 15))
    return HITADV_E_ARG;
  PlainEpi epi{dX, nullptr, Cin, 0};
  epi.gate = xmask;
  hipStream_t s = (hipStream_t)stream;
  if (ns == 32)
    return launch_gemm16<GroupBwdA<G16_U, 32>, PlainEpi, false>(GroupBwdA<G16_U, 32>{dm, arg, Cout}, Wtp, 1, (int)(G * ns), Cin, Cout, epi, range_flag, s);
  return launch_gemm16<GroupBwdA<G16_U, 64>, PlainEpi, false>(GroupBwdA<G16_U, 64>{dm, arg, Cout}, Wtp, 1, (int)(G * ns), Cin, Cout, epi, range_flag, s);
}

#ifdef HITADV_G16_TUNE
template <int ABL>
static int g16_ablate_launch(const float *X, const uint16_t *Wp, long long M, int N, int K, float *C, hipStream_t s) {
  using AP = PlainA<false, G16_U>;
  const AP ap{X, nullptr, K};
  const PlainEpi epi{C, nullptr, N, 0};
  const int chunks = (int)((M + G16_BM - 1) / G16_BM), nrb = chunks, ncb = N / G16_BN;
  HITADV_RAISE_LDS((&gemm_f16x2_k<G16_WR, AP, PlainEpi, false, ABL>), 2 * G16_STAGE);
  gemm_f16x2_k<G16_WR, AP, PlainEpi, false, ABL><<<dim3((unsigned)(nrb * ncb)), G16_WR * 128, 2 * G16_STAGE, s>>>(
      ap, Wp, (int)M, chunks, nrb, ncb, N, K, epi, nullptr);
  return (int)hipGetLastError();
}
extern "C" int hitadv_gemm_f16x2_ablate(int abl, const float *X, const uint16_t *Wp, long long M, int N, int K, float *C, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (abl >= 10) {
    const PlainEpi epi{C, nullptr, N, 0};
    const int chunks = (int)((M + G16_BM - 1) / G16_BM), nrb = chunks, ncb = N / G16_BN;
#define HITADV_RING_ABL(A_)                                                                                                    \
  case A_: {                                                                                                                   \
    HITADV_RAISE_LDS((&gemm_f16x2_ring_k<PlainEpi, false, A_ == 10 ? 0 : A_>), GR_STAGES * GR_STAGE);                          \
    gemm_f16x2_ring_k<PlainEpi, false, A_ == 10 ? 0 : A_><<<dim3((unsigned)(nrb * ncb)), 512, GR_STAGES * GR_STAGE, s>>>(       \
        X, Wp, (int)M, chunks, nrb, ncb, N, K, epi, nullptr);                                                                  \
    return (int)hipGetLastError();                                                                                             \
  }
    switch (abl) {
      HITADV_RING_ABL(10)
      HITADV_RING_ABL(11)
      HITADV_RING_ABL(12)
      HITADV_RING_ABL(13)
      HITADV_RING_ABL(14)
      HITADV_RING_ABL(15)
    }
#undef HITADV_RING_ABL
    return -1;
  }
  switch (abl) {
    case 0: return g16_ablate_launch<0>(X, Wp, M, N, K, C, s);
    case 1: return g16_ablate_launch<1>(X, Wp, M, N, K, C, s);
    case 2: return g16_ablate_launch<2>(X, Wp, M, N, K, C, s);
    case 3: return g16_ablate_launch<3>(X, Wp, M, N, K, C, s);
    case 4: return g16_ablate_launch<4>(X, Wp, M, N, K, C, s);
  }
  return -1;
}
#endif
