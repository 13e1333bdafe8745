/* Entry points of libhitadv_experimental.so (`make -C hit_adv_amd/csrc experimental`): kernels that were built, tested and measured
 * but that NO product path calls.  Not part of libhitadv_hip.so, not loaded by the package; tools/experimental/ holds the Python
 * wrappers and the tests skip when the library has not been built.  Today: the filtered form of PointNet's 128 -> 1024 layer + max
 * over the points (model/feature_models.py:126-127 of the reference; docs/kernels/round5.md section 3: parity with the unfiltered
 * kernel at 257 us, so it stayed out of the loop). */
#ifndef HITADV_EXPERIMENTAL_H
#define HITADV_EXPERIMENTAL_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* The same layer + max over the points WITHOUT two thirds of its matrix work (csrc/victim_filter.hip, round 5): the first of the
 * three fp16 products alone is evaluated for every (point, channel); only the points that can still be the maximum -- those whose
 * approximate value comes within a rigorous error bound of the exact value at last iteration's winner -- are evaluated exactly,
 * and the maximum / first arg-max is taken over them.  Same contract as hitadv_linear_max_fwd_f16x2_packed (Xp: packed pieces,
 * W2: hitadv_split_weights_f16x2), fp32-accurate values, ties to the lower point; the result does not depend on `seed`.
 *   wnorm [Cout]      >= |W[c,:]|_2 (fp32; an upper bound is fine)
 *   seed  [B,Cout]    in: any point per channel (last call's idx: the closer to the winner, the shorter the lists; out-of-range
 *                     values are read as 0); out: this call's idx
 *   scratch           hitadv_linear_max_filter_scratch_words(B, Cout) 32-bit words
 *   range_flag        raised (never cleared) when a candidate list does not fit (HITADV_V1F_CAP entries per cloud and 32
 *                     channels: pathological input); out / idx are then invalid and the caller uses the unfiltered form
 * Supported (hitadv_linear_max_filter_supported): Cin = 128, Cout a multiple of 256, N a multiple of 128, and at least as
 * many (cloud, 256-channel group) pairs as workgroups (`blocks`, 0 = 256): whole clouds stream through a workgroup. */
#define HITADV_V1F_CAP 2048
int hitadv_linear_max_filter_supported(int B, int N, int Cin, int Cout, int blocks);
int64_t hitadv_linear_max_filter_scratch_words(int B, int Cout);
int hitadv_linear_max_fwd_f16x2_filtered(const uint32_t *Xp, const uint16_t *W2, const float *wnorm, const float *bias, int B,
                                         int N, int Cin, int Cout, int relu, int blocks, int64_t *seed, uint32_t *scratch,
                                         float *out, int64_t *idx, int32_t *range_flag, void *stream);

#ifdef __cplusplus
}
#endif
#endif
