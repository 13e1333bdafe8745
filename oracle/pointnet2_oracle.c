/*
 * CPU ORACLE -- test infrastructure, NOT product code.
 *
 * Serial C restatement of the reference's pointnet2_ops CUDA extension
 * (pointnet2_ops_lib/pointnet2_ops/_ext-src/src/ .cu files) plus the project's
 * canonical direct-difference kNN / nearest-neighbour / FPS-from-start rules.
 * The original kernels need nvcc + ATen + an NVIDIA GPU, none of which exist in
 * the build image, so they are "unbuildable here" and this restatement is
 * PARITY-UNPINNED against them: the reference ships no tests or golden vectors
 * for these kernels; tests/test_oracle_natives.py pins it with hand-derived
 * known-answer cases instead.
 *
 * Arithmetic rule used everywhere: fp32, one rounding per operation, no FMA
 * contraction (build with -ffp-contract=off), squared distance evaluated as
 * ((dx*dx + dy*dy) + dz*dz).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float sqdist3(const float *a, const float *b) {
  float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return (dx * dx + dy * dy) + dz * dz;
}

/*
 * The reference's OWN fp32 arithmetic for a squared distance, as executed by torch on an x86 CPU (the GEMM behind
 * torch.bmm / torch.matmul accumulates a K = 3 dot product as one FMA chain in k order; checked bit for bit against
 * torch in tests/test_oracle_gram.py):
 *   form 1  util/set_distance.py:15-32   P = (rx_i + ry_j) - 2 zz_ij, rx / ry = diagonal of bmm(x, x^T) (an FMA chain)
 *   form 2  util/dist_utils.py:148-150   dist = (xx_j + (-2 zz_ij)) + xx_i, xx = sum(pc ** 2, dim=1) (plain adds)
 *   form 3  model/pointnet2_utils.py:19-41 (= model/pct_utils.py:40-58, ShapeAttack/HiT_ADV.py:447-468) square_distance(src = q, dst = p):
 *           dist = -2 zz; dist += sum(src ** 2, -1); dist += sum(dst ** 2, -1)  =>  ((-2 zz_ij) + r_i) + r_j, r = (x*x + y*y) + z*z
 *   form 4  util/other_utils.py:237-251 get_dists(points1 = ONE point per cloud [B,1,3], points2) as PCT's sampler calls it:
 *           sqrt(where(d < 0, 1e-7, d)), d = (r_q + r_p) - 2 zz.  The product of a ONE-row matrix does not go through the GEMM
 *           kernel: MKL evaluates it as fma(q1, p1, q0 p0) + q2 p2 (checked bit for bit, tests/test_oracle_gram.py).
 * form 0 is the project's direct form.  q = the row / query point (i), p = the column / reference point (j).
 */
static inline float dot3_fma(const float *a, const float *b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }
static inline float dot3_row(const float *a, const float *b) { return fmaf(a[1], b[1], a[0] * b[0]) + a[2] * b[2]; }
static inline float sumsq3(const float *a) { return (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]; }

static inline float pair_value(int form, const float *q, const float *p) {
  if (form == 1) {
    float rq = dot3_fma(q, q), rp = dot3_fma(p, p), zz = dot3_fma(q, p);
    return (rq + rp) - 2.0f * zz;
  }
  if (form == 2) {
    float xq = (q[0] * q[0] + q[1] * q[1]) + q[2] * q[2], xp = (p[0] * p[0] + p[1] * p[1]) + p[2] * p[2];
    float inner = -2.0f * dot3_fma(q, p);
    return (xp + inner) + xq;
  }
  if (form == 3) return ((-2.0f * dot3_fma(q, p)) + sumsq3(q)) + sumsq3(p);
  if (form == 4) {
    float d = (sumsq3(q) + sumsq3(p)) - 2.0f * dot3_row(q, p);
    if (d < 0) d = 1e-7f;
    return sqrtf(d);
  }
  return sqdist3(q, p);
}

/* P[b,i,j] in the given form (x [b,n,3] rows, y [b,m,3] columns). */
void oracle_pairwise_form(int b, int n, int m, int form, const float *x, const float *y, float *P) {
  for (int bi = 0; bi < b; ++bi)
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < m; ++j)
        P[((size_t)bi * n + i) * m + j] = pair_value(form, x + ((size_t)bi * n + i) * 3, y + ((size_t)bi * m + j) * 3);
}

/* include/cuda_utils.h:15-18: clamp(2^floor(log2 n), 1, 512) */
static int opt_n_threads(int work) {
  int p = (int)(log((double)work) / log(2.0));
  int t = 1 << p;
  if (t > 512) t = 512;
  if (t < 1) t = 1;
  return t;
}

/*
 * furthest_point_sampling_kernel, src/sampling_gpu.cu:69-173.
 * Emulates the thread-block structure because the tie rule depends on it:
 * thread t scans k = t, t+bs, ... keeping the first strict maximum; the shared
 * memory tree then keeps the LOWER slot on ties (__update :59-65).
 * dataset (b,n,3) -> idxs (b,m) int32; temp is the caller's (b,n) scratch,
 * initialised to 1e10 by the host wrapper (src/sampling.cpp:70-76).
 */
void oracle_fps_ext(int b, int n, int m, const float *dataset, float *temp, int32_t *idxs) {
  if (m <= 0) return;
  const int bs = opt_n_threads(n);
  float *dists = (float *)malloc(sizeof(float) * bs);
  int *dists_i = (int *)malloc(sizeof(int) * bs);
  for (int bi = 0; bi < b; ++bi) {
    const float *pts = dataset + (size_t)bi * n * 3;
    float *tmp = temp + (size_t)bi * n;
    int32_t *out = idxs + (size_t)bi * m;
    int old = 0;
    out[0] = old;
    for (int j = 1; j < m; ++j) {
      const float x1 = pts[old * 3 + 0], y1 = pts[old * 3 + 1], z1 = pts[old * 3 + 2];
      for (int tid = 0; tid < bs; ++tid) {
        int besti = 0;
        float best = -1;
        for (int k = tid; k < n; k += bs) {
          float x2 = pts[k * 3 + 0], y2 = pts[k * 3 + 1], z2 = pts[k * 3 + 2];
          float mag = (x2 * x2 + y2 * y2) + z2 * z2;
          if (mag <= 1e-3) continue; /* zero-point skip, :100-101 */
          float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
          float d = (dx * dx + dy * dy) + dz * dz;
          float d2 = d < tmp[k] ? d : tmp[k];
          tmp[k] = d2;
          besti = d2 > best ? k : besti;
          best = d2 > best ? d2 : best;
        }
        dists[tid] = best;
        dists_i[tid] = besti;
      }
      for (int s = bs / 2; s >= 1; s >>= 1) {
        for (int tid = 0; tid < s; ++tid) {
          float v1 = dists[tid], v2 = dists[tid + s];
          int i1 = dists_i[tid], i2 = dists_i[tid + s];
          dists[tid] = v1 > v2 ? v1 : v2;
          dists_i[tid] = v2 > v1 ? i2 : i1;
        }
      }
      old = dists_i[0];
      out[j] = old;
    }
  }
  free(dists);
  free(dists_i);
}

/* gather_points_kernel, src/sampling_gpu.cu:8-20: out[b,c,j] = points[b,c,idx[b,j]] */
void oracle_gather_points(int b, int c, int n, int npoints, const float *points,
                          const int32_t *idx, float *out) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        out[((size_t)i * c + l) * npoints + j] =
            points[((size_t)i * c + l) * n + idx[(size_t)i * npoints + j]];
}

/* gather_points_grad_kernel, src/sampling_gpu.cu:34-47 (atomicAdd scatter; serial order here) */
void oracle_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                               const int32_t *idx, float *grad_points) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        grad_points[((size_t)i * c + l) * n + idx[(size_t)i * npoints + j]] +=
            grad_out[((size_t)i * c + l) * npoints + j];
}

/*
 * query_ball_point_kernel, src/ball_query_gpu.cu:9-44.  First nsample indices
 * (ascending) with d2 < r2 (strict); the first hit pre-fills the whole row; a
 * row with no hit keeps the caller's zeros (ball_query.cpp:19-21).
 */
void oracle_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                       const float *xyz, int32_t *idx) {
  const float r2 = radius * radius;
  for (int bi = 0; bi < b; ++bi) {
    const float *P = xyz + (size_t)bi * n * 3;
    const float *Q = new_xyz + (size_t)bi * m * 3;
    int32_t *out = idx + (size_t)bi * m * nsample;
    for (int j = 0; j < m; ++j) {
      const float qx = Q[j * 3 + 0], qy = Q[j * 3 + 1], qz = Q[j * 3 + 2];
      for (int k = 0, cnt = 0; k < n && cnt < nsample; ++k) {
        float dx = qx - P[k * 3 + 0], dy = qy - P[k * 3 + 1], dz = qz - P[k * 3 + 2];
        float d2 = (dx * dx + dy * dy) + dz * dz;
        if (d2 < r2) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) out[j * nsample + l] = k;
          out[j * nsample + cnt] = k;
          ++cnt;
        }
      }
    }
  }
}

/* group_points_kernel, src/group_points_gpu.cu:8-28 */
void oracle_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                         const int32_t *idx, float *out) {
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k) {
          int ii = idx[((size_t)bi * npoints + j) * nsample + k];
          out[(((size_t)bi * c + l) * npoints + j) * nsample + k] =
              points[((size_t)bi * c + l) * n + ii];
        }
}

/* group_points_grad_kernel, src/group_points_gpu.cu:43-64 */
void oracle_group_points_grad(int b, int c, int n, int npoints, int nsample,
                              const float *grad_out, const int32_t *idx, float *grad_points) {
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k) {
          int ii = idx[((size_t)bi * npoints + j) * nsample + k];
          grad_points[((size_t)bi * c + l) * n + ii] +=
              grad_out[(((size_t)bi * c + l) * npoints + j) * nsample + k];
        }
}

/* three_nn_kernel, src/interpolate_gpu.cu:9-59: double-typed bests, strict '<' cascade */
void oracle_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                     int32_t *idx) {
  for (int bi = 0; bi < b; ++bi) {
    const float *U = unknown + (size_t)bi * n * 3;
    const float *K = known + (size_t)bi * m * 3;
    for (int j = 0; j < n; ++j) {
      double best1 = 1e40, best2 = 1e40, best3 = 1e40;
      int i1 = 0, i2 = 0, i3 = 0;
      for (int k = 0; k < m; ++k) {
        float dx = U[j * 3 + 0] - K[k * 3 + 0], dy = U[j * 3 + 1] - K[k * 3 + 1],
              dz = U[j * 3 + 2] - K[k * 3 + 2];
        float d = (dx * dx + dy * dy) + dz * dz;
        if (d < best1) {
          best3 = best2; i3 = i2;
          best2 = best1; i2 = i1;
          best1 = d; i1 = k;
        } else if (d < best2) {
          best3 = best2; i3 = i2;
          best2 = d; i2 = k;
        } else if (d < best3) {
          best3 = d; i3 = k;
        }
      }
      size_t o = ((size_t)bi * n + j) * 3;
      dist2[o + 0] = (float)best1; dist2[o + 1] = (float)best2; dist2[o + 2] = (float)best3;
      idx[o + 0] = i1; idx[o + 1] = i2; idx[o + 2] = i3;
    }
  }
}

/* three_interpolate_kernel, src/interpolate_gpu.cu:72-101 */
void oracle_three_interpolate(int b, int c, int m, int n, const float *points, const int32_t *idx,
                              const float *weight, float *out) {
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        const size_t o = ((size_t)bi * n + j) * 3;
        const float *p = points + ((size_t)bi * c + l) * m;
        out[((size_t)bi * c + l) * n + j] =
            (p[idx[o]] * weight[o] + p[idx[o + 1]] * weight[o + 1]) + p[idx[o + 2]] * weight[o + 2];
      }
}

/* three_interpolate_grad_kernel, src/interpolate_gpu.cu:116-143 */
void oracle_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                   const int32_t *idx, const float *weight, float *grad_points) {
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        const size_t o = ((size_t)bi * n + j) * 3;
        float *g = grad_points + ((size_t)bi * c + l) * m;
        const float go = grad_out[((size_t)bi * c + l) * n + j];
        g[idx[o]] += go * weight[o];
        g[idx[o + 1]] += go * weight[o + 1];
        g[idx[o + 2]] += go * weight[o + 2];
      }
}

/*
 * Canonical K nearest neighbours (stand-in rule for pytorch3d.ops.knn_points as
 * called at ShapeAttack/HiT_ADV.py:78,320,329): K smallest direct-difference
 * squared distances, ascending, ties -> lower index.  Insertion list per query.
 */
void oracle_knn_points_form(int b, int n, int m, int K, int form, const float *q, const float *p, float *dists,
                            int64_t *idx) {
  for (int bi = 0; bi < b; ++bi)
    for (int i = 0; i < n; ++i) {
      float *bd = dists + ((size_t)bi * n + i) * K;
      int64_t *bix = idx + ((size_t)bi * n + i) * K;
      int cnt = 0;
      for (int j = 0; j < m; ++j) {
        float d = pair_value(form, q + ((size_t)bi * n + i) * 3, p + ((size_t)bi * m + j) * 3);
        if (cnt == K && !(d < bd[K - 1])) continue;
        int pos = cnt < K ? cnt : K - 1;
        while (pos > 0 && d < bd[pos - 1]) {
          bd[pos] = bd[pos - 1];
          bix[pos] = bix[pos - 1];
          --pos;
        }
        bd[pos] = d;
        bix[pos] = j;
        if (cnt < K) ++cnt;
      }
    }
}

void oracle_knn_points(int b, int n, int m, int K, const float *q, const float *p, float *dists,
                       int64_t *idx) {
  oracle_knn_points_form(b, n, m, K, 0, q, p, dists, idx);
}

/* Nearest neighbour of every x_i in y (lowest index on ties), distances in the given form. */
void oracle_nn_min_form(int b, int n, int m, int form, const float *x, const float *y, float *mind,
                        int32_t *argm) {
  for (int bi = 0; bi < b; ++bi)
    for (int i = 0; i < n; ++i) {
      float best = INFINITY;
      int bj = 0;
      for (int j = 0; j < m; ++j) {
        float d = pair_value(form, x + ((size_t)bi * n + i) * 3, y + ((size_t)bi * m + j) * 3);
        if (d < best) { best = d; bj = j; }
      }
      mind[(size_t)bi * n + i] = best;
      argm[(size_t)bi * n + i] = bj;
    }
}

void oracle_nn_min(int b, int n, int m, const float *x, const float *y, float *mind, int32_t *argm) {
  oracle_nn_min_form(b, n, m, 0, x, y, mind, argm);
}

/*
 * FPS with a given first index (ShapeAttack/HiT_ADV.py:489-510): running
 * distance 1e10, strict '<' update, arg-max with lowest index on ties.
 */
void oracle_fps_from_start(int b, int n, int m, const float *xyz, const int64_t *start,
                           int64_t *idxs) {
  float *run = (float *)malloc(sizeof(float) * n);
  for (int bi = 0; bi < b; ++bi) {
    const float *P = xyz + (size_t)bi * n * 3;
    for (int k = 0; k < n; ++k) run[k] = 1e10f;
    int64_t far = start[bi];
    for (int j = 0; j < m; ++j) {
      idxs[(size_t)bi * m + j] = far;
      float best = -INFINITY;
      int64_t bk = 0;
      for (int k = 0; k < n; ++k) {
        float d = sqdist3(P + k * 3, P + far * 3);
        if (d < run[k]) run[k] = d;
        if (run[k] > best) { best = run[k]; bk = k; }
      }
      far = bk;
    }
  }
  free(run);
}

/*
 * PCT's sampler, util/other_utils.py:254-272: running distance 1e5, distances by get_dists (form 4: sqrt of the clamped
 * Gram form, the current point as a one-row matrix), update where cur < running, arg-max of the running distances with
 * the lowest index on ties (torch.max over a CPU tensor).
 */
void oracle_fps_pct(int b, int n, int m, const float *xyz, const int64_t *start, int64_t *idxs) {
  float *run = (float *)malloc(sizeof(float) * n);
  for (int bi = 0; bi < b; ++bi) {
    const float *P = xyz + (size_t)bi * n * 3;
    for (int k = 0; k < n; ++k) run[k] = 1e5f;
    int64_t far = start[bi];
    for (int j = 0; j < m; ++j) {
      idxs[(size_t)bi * m + j] = far;
      float best = -INFINITY;
      int64_t bk = 0;
      for (int k = 0; k < n; ++k) {
        float d = pair_value(4, P + far * 3, P + k * 3);
        if (d < run[k]) run[k] = d;
        if (run[k] > best) { best = run[k]; bk = k; }
      }
      far = bk;
    }
  }
  free(run);
}

/*
 * query_ball_point of the victims, model/pointnet2_utils.py:87-107 (= model/pct_utils.py:77-96,
 * ShapeAttack/HiT_ADV.py:512-532): square_distance in the given form, entries with d > r2 dropped (r2 = the fp32 value
 * of the Python double radius ** 2: a float tensor compared with a Python scalar is compared in fp32), the first nsample
 * survivors in index order, padded with the first survivor; an empty ball is nsample times n.
 */
void oracle_query_ball_form(int b, int n, int m, float r2, int nsample, int form, const float *new_xyz, const float *xyz,
                            int64_t *idx) {
  for (int bi = 0; bi < b; ++bi)
    for (int j = 0; j < m; ++j) {
      const float *q = new_xyz + ((size_t)bi * m + j) * 3;
      int64_t *out = idx + ((size_t)bi * m + j) * nsample;
      int cnt = 0;
      for (int k = 0; k < n && cnt < nsample; ++k) {
        float d = pair_value(form, q, xyz + ((size_t)bi * n + k) * 3);
        if (!(d > r2)) out[cnt++] = k;
      }
      int64_t first = cnt ? out[0] : n;
      for (int l = cnt; l < nsample; ++l) out[l] = first;
    }
}
