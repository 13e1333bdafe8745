// Device-side body of the deformation's forward pass (csrc/deform.hip), shared with the PointNet engine's first kernel.
#pragma once
#include "common.hpp"

namespace hitadv {

constexpr int DF_PTS = 64;    // points per block (forward)
constexpr float LOG2E = 1.4426950408889634f;

// One block = DF_PTS points of cloud b (tile = which).  CMAX = centres staged per LDS pass.  xs != nullptr: the deformed
// points of the tile are also left in LDS, xs[3 * lane + d] (zeros past N), for a caller that goes on with them
// (csrc/pointnet.hip: the PointNet engine's first kernel).
// LDS the body needs, in float4: CMAX (centre + exponent scale) + CMAX (translation) + 4 * DF_PTS (the waves' partial sums)
template <int CMAX>
__host__ __device__ constexpr int deform_fwd_lds_float4() { return 2 * CMAX + 4 * DF_PTS; }

template <int CMAX>
__device__ __forceinline__ void deform_fwd_body_in(float4 *lds, const float *__restrict__ ori, const float *__restrict__ central,
                                                   const float *__restrict__ perturb, const float *__restrict__ sigma, int N,
                                                   int C, float *__restrict__ adv, float *__restrict__ inv_den, const int b,
                                                   const int tile, float *xs) {
  float4 *const sc = lds;                // cx cy cz a
  float4 *const sp = lds + CMAX;         // px py pz -
  float4(*const part)[DF_PTS] = reinterpret_cast<float4(*)[DF_PTS]>(lds + 2 * CMAX);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = tile * DF_PTS + lane;
  const int nn = n < N ? n : N - 1;
  const float *op = ori + (size_t)b * 3 * N;
  const float x = op[nn], y = op[N + nn], z = op[2 * N + nn];
  float sx = 0.f, sy = 0.f, sz = 0.f, den = 0.f;
  for (int c0 = 0; c0 < C; c0 += CMAX) {
    const int cnt = min(CMAX, C - c0);
    __syncthreads();
    for (int j = threadIdx.x; j < cnt; j += 256) {
      const float *cp = central + (size_t)b * 3 * C + c0 + j;
      const float *pp = perturb + ((size_t)b * C + c0 + j) * 3;
      const float s = sigma[(size_t)b * C + c0 + j];
      sc[j] = make_float4(cp[0], cp[C], cp[2 * C], -LOG2E / (2.0f * s * s));
      sp[j] = make_float4(pp[0], pp[1], pp[2], 0.f);
    }
    __syncthreads();
    const int per = (cnt + 3) >> 2;
    const int lo = wave * per, hi = min(lo + per, cnt);
#pragma unroll 4
    for (int j = lo; j < hi; ++j) {
      const float4 c = sc[j];
      const float4 p = sp[j];
      const float r = sqrt_rn_ranged(sqdist3(x, y, z, c.x, c.y, c.z));  // (k itself keeps exp2f: the sum of the k is divided by)
      const float k = exp2f(r * c.w);
      sx = fmaf(k, p.x, sx);
      sy = fmaf(k, p.y, sy);
      sz = fmaf(k, p.z, sz);
      den = den + k;
    }
  }
  part[wave][lane] = make_float4(sx, sy, sz, den);
  __syncthreads();
  if (wave == 0 && n < N) {
    float4 a = part[0][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 o = part[w][lane];
      a.x += o.x;
      a.y += o.y;
      a.z += o.z;
      a.w += o.w;
    }
    const float inv = 1.0f / a.w;
    float *ap = adv + (size_t)b * 3 * N;
    ap[n] = x + a.x * inv;
    ap[N + n] = y + a.y * inv;
    ap[2 * N + n] = z + a.z * inv;
    inv_den[(size_t)b * N + n] = inv;
    if (xs != nullptr) {
      xs[3 * lane] = x + a.x * inv;
      xs[3 * lane + 1] = y + a.y * inv;
      xs[3 * lane + 2] = z + a.z * inv;
    }
  } else if (wave == 0 && xs != nullptr) {
    xs[3 * lane] = xs[3 * lane + 1] = xs[3 * lane + 2] = 0.f;
  }
}

// ... with LDS of its own (csrc/deform.hip, rowmlp_fwd_k / rowmlp_fwd16_k)
template <int CMAX>
__device__ __forceinline__ void deform_fwd_body(const float *__restrict__ ori, const float *__restrict__ central,
                                                const float *__restrict__ perturb, const float *__restrict__ sigma, int N,
                                                int C, float *__restrict__ adv, float *__restrict__ inv_den, const int b,
                                                const int tile, float *xs) {
  __shared__ float4 lds[deform_fwd_lds_float4<CMAX>()];
  deform_fwd_body_in<CMAX>(lds, ori, central, perturb, sigma, N, C, adv, inv_den, b, tile, xs);
}


}  // namespace hitadv
