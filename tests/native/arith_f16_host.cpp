// HOST build (ROCm's clang++ for x86-64: _Float16) of the two-piece fp16 split of hit_adv_amd/csrc/arith.hpp, for tests/test_arith_host.py.
// `spelled` is the split as csrc/victim_bf3.hip's staging (MODE 1) and split_weights_k<1> write it out: convert back, subtract, scale, convert.
#include "arith.hpp"

using namespace hitadv;

extern "C" void arith_split_pair(long n, const float *a, uint16_t *hi, uint16_t *lo, uint16_t *lo_spelled) {
  for (long i = 0; i < n; ++i) {
    _Float16 h, l;
    split_pair(a[i], h, l);
    const _Float16 h1 = (_Float16)a[i];
    const _Float16 l2 = (_Float16)((a[i] - (float)h1) * F16X2_PIECE_SCALE);
    hi[i] = __builtin_bit_cast(uint16_t, h);
    lo[i] = __builtin_bit_cast(uint16_t, l);
    lo_spelled[i] = __builtin_bit_cast(uint16_t, l2);
  }
}
