// HOST build of hit_adv_amd/csrc/arith.hpp -- the very source text the gfx950 kernels compile (through common.hpp) -- behind a C ABI for
// tests/test_arith_host.py.  Test infrastructure; built into a temporary directory by the test (g++ -O2 -ffp-contract=off, like the
// library's own -ffp-contract=off), never shipped.
#include "arith.hpp"

using namespace hitadv;

template <int FORM>
static void pairwise_form(long n, long m, const float *x, const float *y, float *P) {
  for (long i = 0; i < n; ++i) {
    const float *q = x + 3 * i;
    const float rq = sq_norm<FORM>(q[0], q[1], q[2]);  // as the kernels do: the norms in the form's own arithmetic (csrc/pairwise.hip, knn.hip)
    for (long j = 0; j < m; ++j) {
      const float *p = y + 3 * j;
      const float rp = sq_norm<FORM>(p[0], p[1], p[2]);
      P[i * m + j] = pair_dist<FORM>(q[0], q[1], q[2], rq, p[0], p[1], p[2], rp);
    }
  }
}

extern "C" {

// P[n, m] = the kernels' pair value of (row point x_i, column point y_j) in form 0..4 (include/hitadv.h HITADV_FORM_*; 4 = PCT's get_dists)
int arith_pairwise(int form, long n, long m, const float *x, const float *y, float *P) {
  switch (form) {
    case 0: pairwise_form<0>(n, m, x, y, P); return 0;
    case 1: pairwise_form<1>(n, m, x, y, P); return 0;
    case 2: pairwise_form<2>(n, m, x, y, P); return 0;
    case 3: pairwise_form<3>(n, m, x, y, P); return 0;
    case 4:
      for (long i = 0; i < n; ++i)
        for (long j = 0; j < m; ++j) {
          const float *q = x + 3 * i, *p = y + 3 * j;
          P[i * m + j] = pct_dist(q[0], q[1], q[2], sq_norm<0>(q[0], q[1], q[2]), p[0], p[1], p[2], sq_norm<0>(p[0], p[1], p[2]));
        }
      return 0;
  }
  return -1;
}

void arith_split3(long n, const float *a, uint32_t *hi, uint32_t *mid, uint32_t *lo, uint32_t *packed_hi_pairs) {
  for (long i = 0; i < n; ++i) split3(a[i], hi[i], mid[i], lo[i]);
  for (long i = 0; i + 1 < n; i += 2) packed_hi_pairs[i / 2] = pack_hi(hi[i], hi[i + 1]);
}

void arith_sqrt_preimage_floor(long n, const float *s, float *out) {
  for (long i = 0; i < n; ++i) out[i] = sqrt_preimage_floor(s[i]);
}

// one torch.optim.Adam step (t = 1-based step number) on n coordinates, in place, by the kernels' own adam_coef / adam_update
void arith_adam_step(long n, int t, double lr, float *p, const float *g, float *m, float *v) {
  const AdamCoef k = adam_coef(t, lr);
  for (long i = 0; i < n; ++i) p[i] = adam_update(p[i], g[i], m[i], v[i], k);
}

void arith_fbits(long n, const float *v, uint32_t *out) {
  for (long i = 0; i < n; ++i) out[i] = fbits(v[i]);
}
}
