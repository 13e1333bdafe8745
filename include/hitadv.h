/*
 * hitadv.h -- C ABI of libhitadv_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the HiT-ADV data-parallel hot path.  The reference
 * (TRLou/HiT-ADV) has no FFI of its own for most of this path -- it is Python
 * calling torch ops -- so each entry point below names the reference code it
 * replaces (file:line relative to the reference root).  The nine pointnet2_ops
 * entry points mirror the reference's own `*_kernel_wrapper` prototypes
 * (pointnet2_ops_lib/pointnet2_ops/_ext-src/src/{sampling,ball_query,
 * group_points,interpolate}.cpp) one for one.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HBM), fp32 / int32 / int64 as typed,
 *     dense row-major ("contiguous") in the shapes given;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *     every call only enqueues work -- no allocation, no synchronisation, so
 *     calls may be captured into a hipGraph;
 *   - return value: 0 on success, a hipError_t code if the launch failed,
 *     HITADV_E_ARG (-1) for an invalid argument.  Nothing calls exit() (the
 *     reference's CUDA_CHECK_ERRORS does, include/cuda_utils.h:30-39);
 *   - inputs are never written; outputs are fully overwritten unless stated.
 *
 * Squared distances in the "direct" form are evaluated in fp32 as
 * ((dx*dx + dy*dy) + dz*dz), one rounding per operation, no FMA contraction,
 * which is what makes index outputs bit-reproducible against the CPU oracle.
 */
#ifndef HITADV_H
#define HITADV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HITADV_E_ARG (-1)

#define HITADV_FORM_DIRECT 0 /* ((dx*dx+dy*dy)+dz*dz)                           */
#define HITADV_FORM_GRAM 1   /* (|x|^2+|y|^2) - 2 x.y, dot products as the FMA chain fma(a2,b2,fma(a1,b1,a0*b0)):
                              * _Distance.batch_pairwise_dist (util/set_distance.py:15-32) as torch.bmm evaluates it on
                              * an x86 CPU -- the reference's values bit for bit (tests/test_oracle_gram.py)          */
#define HITADV_FORM_GRAM_KNN 2 /* (|y_j|^2 + (-2 x_i.y_j)) + |x_i|^2, dot product an FMA chain, |.|^2 = (a*a+b*b)+c*c:
                                * the distance matrix of KNNDist (util/dist_utils.py:148-150), bit for bit             */
#define HITADV_FORM_SQUARE_DISTANCE 3 /* ((-2 q_i.p_j) + |q_i|^2) + |p_j|^2, dot product an FMA chain, |.|^2 plain:
                                       * square_distance(src = q, dst = p) of the victims (model/pointnet2_utils.py:19-41,
                                       * model/pct_utils.py:40-58; also ShapeAttack/HiT_ADV.py:447-468), bit for bit   */

/* Library / build identification (static string). */
const char *hitadv_version(void);

/* ------------------------------------------------------------------ set distances */

/* P[b,i,j] = |x_i - y_j|^2.  x[B,N,D], y[B,M,D] -> P[B,N,M].
 * Replaces _Distance.batch_pairwise_dist (util/set_distance.py:15-32) and its inline
 * variants (util/dist_utils.py:148-150, model/dgcnn_cls.py:8-10).
 * D == 3 is the tuned path (HBM-store-bound); other D use a generic kernel that always
 * evaluates the direct form. */
int hitadv_pairwise_sqdist(const float *x, const float *y, float *P, int B, int N, int M, int D,
                           int form, void *stream);

/* Fused nearest-neighbour reduction in both directions, no matrix materialised (D == 3):
 *   min_x[b,i] = min_j |x_i - y_j|^2, arg_x[b,i] = lowest such j   (and symmetrically for y).
 * Replaces the two torch.min passes of ChamferDistance/HausdorffDistance.forward
 * (util/set_distance.py:40-50, 58-70).  `form` = HITADV_FORM_DIRECT, or HITADV_FORM_GRAM (D == 3 only): the minima of
 * the reference's own Gram-form matrix, bit for bit.  For D != 3, `scratch` must hold B*N*M floats.
 * Any of the four outputs may be NULL only in pairs (min_y/arg_y together). */
int hitadv_nn_min(const float *x, const float *y, int B, int N, int M, int D, int form, float *min_x,
                  int32_t *arg_x, float *min_y, int32_t *arg_y, float *scratch, void *stream);

/* Backward of hitadv_nn_min through the saved arg-mins (what autograd does through
 * torch.min at util/set_distance.py:46-49):
 *   grad_x[b,i,:] = 2 g_min_x[b,i] (x_i - y_arg_x[i]) + sum_{j: arg_y[j]==i} 2 g_min_y[b,j] (x_i - y_j)
 * and symmetrically grad_y (skipped when grad_y == NULL).  Deterministic (no atomics). */
int hitadv_nn_min_bwd(const float *x, const float *y, const int32_t *arg_x, const int32_t *arg_y,
                      const float *g_min_x, const float *g_min_y, int B, int N, int M, int D,
                      float *grad_x, float *grad_y, void *stream);

/* K nearest neighbours, ascending, ties -> lower index.  q[B,N,3], p[B,M,3] ->
 * dists[B,N,K], idx[B,N,K] (int64 when idx_is_i64 != 0, else int32).  1 <= K <= min(M, 64).
 * `form` = HITADV_FORM_DIRECT (pytorch3d's rule), HITADV_FORM_GRAM_KNN (KNNDist's own distance matrix) or
 * HITADV_FORM_SQUARE_DISTANCE (the k smallest entries of the victims' square_distance(q, p): PCT's knn_point,
 * model/pct_utils.py:98-109).
 * Replaces pytorch3d.ops.knn_points as called at ShapeAttack/HiT_ADV.py:78,320,329 and
 * util/dist_utils.py:482, and the Gram+topk of KNNDist (util/dist_utils.py:148-158). */
int hitadv_knn_points(const float *q, const float *p, int B, int N, int M, int K, int form, float *dists,
                      void *idx, int idx_is_i64, void *stream);

/* Backward of hitadv_knn_points w.r.t. both point sets, given g_dists[B,N,K]:
 *   grad_q[b,i,:]  =  sum_t 2 g[b,i,t] (q_i - p_idx[b,i,t])
 *   grad_p[b,j,:]  = -sum_{(i,t): idx[b,i,t]==j} 2 g[b,i,t] (q_i - p_j)      (deterministic)
 * grad_q or grad_p may be NULL. */
int hitadv_knn_points_bwd(const float *q, const float *p, const void *idx, int idx_is_i64,
                          const float *g_dists, int B, int N, int M, int K, float *grad_q,
                          float *grad_p, void *stream);

/* K largest (largest != 0) or smallest entries of every row of P[rows, M], sorted, ties -> lower column.
 * vals[rows,K], idx[rows,K] int64.  The selection step of DGCNN's feature-space kNN
 * (model/dgcnn_cls.py:12 `pairwise_distance.topk(k)`) and of model/pct_utils.py:98-109. */
int hitadv_topk_rows(const float *P, int64_t rows, int M, int K, int largest, float *vals, int64_t *idx,
                     void *stream);

/* ------------------------------------------------------------------ HiT-ADV deformation */

/* Kernel-weighted deformation, replaces HiT_ADV.kernel_density + the C-step accumulation loop
 * (ShapeAttack/HiT_ADV.py:160-175, 298-304):
 *   k[n,j] = exp(-|x_n - c_j|_2 / (2 sigma_j^2)),   adv_n = x_n + (sum_j k p_j) / (sum_j k)
 * ori[B,3,N], central[B,3,C], perturb[B,C,3], sigma[B,C] -> adv[B,3,N], inv_den[B,N] (=1/sum_j k,
 * saved for the backward).  */
int hitadv_deform_fwd(const float *ori, const float *central, const float *perturb,
                      const float *sigma, int B, int N, int C, float *adv, float *inv_den,
                      void *stream);

/* Gradient of the above w.r.t. perturb and sigma for upstream g_adv[B,3,N].
 * `partials` is caller scratch of hitadv_deform_bwd_scratch_floats(B,N,C) floats.
 * Accumulation order is fixed -> bitwise reproducible. */
int hitadv_deform_bwd(const float *ori, const float *central, const float *perturb,
                      const float *sigma, const float *adv, const float *inv_den,
                      const float *g_adv, int B, int N, int C, float *partials,
                      float *grad_perturb, float *grad_sigma, void *stream);
int64_t hitadv_deform_bwd_scratch_floats(int B, int N, int C);

/* ------------------------------------------------------------------ attack-state kernels */

/* On-device replacement of the per-iteration host bookkeeping (ShapeAttack/HiT_ADV.py:186-217):
 * pred = argmax(logits[b,:]) (lowest index on ties), dist_val[b] = (|P_b|_F + |1-sigma_b|_2)/C
 * (transformation_loss batch_avg=False, :313-316), then, only when pred != label,
 * strict-'<' updates of (bestdist,bestscore) and (o_bestdist,o_bestscore,o_bestattack[b,:,:]=adv[b]).
 * Also writes pred_out[B] (int64) and dist_val_out[B], and adds 1 to *iter_counter if non-NULL. */
int hitadv_best_update(const float *logits, const int64_t *label, const float *perturb,
                       const float *sigma, const float *adv, int B, int num_class, int N, int C,
                       float *bestdist, int64_t *bestscore, float *o_bestdist, int64_t *o_bestscore,
                       float *o_bestattack, int64_t *pred_out, float *dist_val_out,
                       int32_t *iter_counter, void *stream);

/* Adam step for the two parameter groups of ShapeAttack/HiT_ADV.py:142-145 in one launch
 * (torch.optim.Adam defaults: betas (0.9,0.999), eps 1e-8, no weight decay, no amsgrad).
 * *step is incremented by the kernel (device-side counter -> graph-capturable). */
int hitadv_adam_step(float *perturb, const float *g_perturb, float *m_perturb, float *v_perturb,
                     int64_t n_perturb, float lr_perturb, float *sigma, const float *g_sigma,
                     float *m_sigma, float *v_sigma, int64_t n_sigma, float lr_sigma,
                     int32_t *step, void *stream);
/* dst[0, nbytes) = src[0, nbytes) by a KERNEL (16-byte words, grid-stride; the two ranges must not overlap).  The attack
 * loops keep their state at fixed addresses and move it with this instead of hipMemcpyAsync (what a contiguous
 * tensor.copy_ / clone is, as in ShapeAttack/HiT_ADV.py:186-217's host copies): inside a captured iteration a memcpy NODE
 * makes graph launches of three streams serialise and holds the host (tools/stream_overlap_probe.py: three streams of
 * captured PCT passes take 1.92x one stream's time without such nodes, 2.47x with seven per pass). */
int hitadv_copy(void *dst, const void *src, int64_t nbytes, void *stream);

/* dst [B,N,C] = src [B,C,N] transposed (to_points_major != 0) or dst [B,C,N] = src [B,N,C] (0), C <= 16: a cloud between the
 * victims' channel-major interface (model/pointnet2_cls_ssg.py:27 `xyz [B,3,N]`) and the points-major layout its samplers work on
 * (`xyz.permute(0, 2, 1)`, model/pointnet2_utils.py:176), as one coalesced kernel. */
int hitadv_transpose_small(const float *src, float *dst, int B, int C, int N, int to_points_major, void *stream);

/* The same step with the gradient given as g + g2 (g2 may be NULL: the deformation's and the regulariser's terms
 * need no separate add), followed by the projection the reference applies at the top of the next iteration
 * (ShapeAttack/HiT_ADV.py:157-158; skipped for a group when lo > hi).  *step is the 1-based step number and is
 * NOT modified (pass the counter that hitadv_best_update bumps once per iteration). */
int hitadv_adam_step_sum(float *perturb, const float *g_perturb, const float *g_perturb2, float *m_perturb,
                         float *v_perturb, int64_t n_perturb, float lr_perturb, float lo_perturb, float hi_perturb,
                         float *sigma, const float *g_sigma, const float *g_sigma2, float *m_sigma, float *v_sigma,
                         int64_t n_sigma, float lr_sigma, float lo_sigma, float hi_sigma, const int32_t *step,
                         void *stream);

/* Adversarial loss on logits[B,K] and its gradient in one launch (util/adv_utils.py):
 *   kind 0 UntargetedLogitsAdvLoss (:38-67), kind 1 LogitsAdvLoss (:6-35), kind 2 CrossEntropyAdvLoss (:70-85);
 * *loss = mean over the batch, dlogits[B,K] = d loss / d logits.  kappa is ignored for kind 2.  B <= 8192. */
int hitadv_adv_loss(int kind, const float *logits, const int64_t *target, int B, int K, float kappa, float *loss,
                    float *dlogits, void *stream);

/* The three distance regularisers of the HiT-ADV loss, fused with the per-sample weighting
 * (ShapeAttack/HiT_ADV.py:229-245):  dist = cd_w*mean_b Q1_b + ker_w*(|P|_F+|1-sigma|_F)/C + hide_w*mean_b cos_b,
 * where Q1_b is ChamferDist('adv2ori') evaluated on the [3,N] tensors exactly as the reference calls it
 * (:230, quirk Q1) and cos_b the cosine similarity of :341-346 against hide_ref[B,C] (the min-max normalised
 * centre curvature-std, constant over an attack).  Writes *dist_loss = dist and
 * *scaled_loss = mean(scale_const) * dist  (== mean_b(scale_const_b * dist), :243-245).
 * scratch: hitadv_regulariser_scratch_floats(B) floats, kept for the backward. */
int hitadv_regulariser_fwd(const float *perturb, const float *sigma, const float *adv, const float *ori,
                           const float *hide_ref, const float *scale_const, int B, int N, int C,
                           float cd_w, float ker_w, float hide_w, float min_sigm, float max_sigm,
                           float *scratch, float *dist_loss, float *scaled_loss, void *stream);
/* Gradients of scaled_loss (times *grad_out) w.r.t. perturb[B,C,3], sigma[B,C] and adv[B,3,N]. */
int hitadv_regulariser_bwd(const float *perturb, const float *sigma, const float *adv, const float *ori,
                           const float *hide_ref, const float *scratch, const float *grad_out, int B, int N,
                           int C, float cd_w, float ker_w, float hide_w, float min_sigm, float max_sigm,
                           float *grad_perturb, float *grad_sigma, float *grad_adv, void *stream);
/* Same, with grad_out optional (NULL = 1) and grad_adv = (regulariser's term) + add_adv[B,3,N] when add_adv != NULL:
 * the victim's input gradient joins here instead of in a separate add. */
int hitadv_regulariser_bwd_add(const float *perturb, const float *sigma, const float *adv, const float *ori,
                               const float *hide_ref, const float *scratch, const float *grad_out,
                               const float *add_adv, int B, int N, int C, float cd_w, float ker_w, float hide_w,
                               float min_sigm, float max_sigm, float *grad_perturb, float *grad_sigma,
                               float *grad_adv, void *stream);
int64_t hitadv_regulariser_scratch_floats(int B);

/* ------------------------------------------------------------------ farthest point sampling */

/* FPS with a given first index per cloud; running distance 1e10, strict-'<' update, arg-max with
 * lowest index on ties.  xyz[B,N,3], start[B] -> idx[B,m] (int64).
 * Replaces HiT_ADV.farthest_point_sample (ShapeAttack/HiT_ADV.py:489-510; the random start of
 * :501 is drawn by the caller) and model/pointnet2_utils.py:63-84. */
int hitadv_fps_from_start(const float *xyz, const int64_t *start, int B, int N, int m,
                          int64_t *idx, void *stream);

/* PCT's sampler (util/other_utils.py:254-272) with a given first index per cloud: running distance 1e5, distances by
 * get_dists (:237-251) = sqrt(d < 0 ? 1e-7 : d), d = (|c|^2 + |p|^2) - 2 c.p in torch's own fp32 arithmetic for a
 * one-row matrix product (c.p = fma(c1, p1, c0 p0) + c2 p2), update where smaller, arg-max with the lowest index on
 * ties: idx[B,m] (int64) equals the reference's table. */
int hitadv_fps_pct(const float *xyz, const int64_t *start, int B, int N, int m, int64_t *idx, void *stream);

/* ------------------------------------------------------------------ pointnet2_ops natives
 * One for one with the reference's kernel wrappers; same argument order.               */

/* src/sampling.cpp:11-13 furthest_point_sampling_kernel_wrapper.  temp[B,N] is scratch that this
 * call initialises itself (the reference's host code fills it with 1e10, sampling.cpp:70-76).
 * Starts at index 0, skips points with |p|^2 <= 1e-3, reference tie rule (thread-slot order). */
int hitadv_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp,
                                   int32_t *idxs, void *stream);
/* src/sampling.cpp:4-6 */
int hitadv_gather_points(int b, int c, int n, int npoints, const float *points, const int32_t *idx,
                         float *out, void *stream);
/* src/sampling.cpp:7-9; grad_points[b,c,n] is fully overwritten (deterministic, no atomics). */
int hitadv_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                              const int32_t *idx, float *grad_points, void *stream);
/* src/ball_query.cpp:4-6; idx[b,m,nsample] is fully written (zeros for an empty ball). */
int hitadv_query_ball_point(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                            const float *xyz, int32_t *idx, void *stream);
/* The pure-torch ball query of the PointNet++ victim and of HiT_ADV.query_ball_point
 * (model/pointnet2_utils.py:87-107, ShapeAttack/HiT_ADV.py:512-532): first nsample indices (ascending) that are NOT
 * `d^2 > radius2`, short rows padded with the first hit, an empty ball filled with n (the reference's out-of-range
 * marker), int64 indices.  `radius2` is the threshold itself: the reference compares an fp32 tensor with the Python
 * double `radius ** 2`, which torch rounds to fp32 -- NOT the fp32 product radius * radius (0.2f * 0.2f is one ulp
 * above (float)0.04).  `form` = HITADV_FORM_SQUARE_DISTANCE (the reference's Gram-form square_distance(new_xyz, xyz),
 * bit for bit: the table equals the reference's) or HITADV_FORM_DIRECT. */
int hitadv_query_ball_point_victim(int b, int n, int m, float radius2, int nsample, int form, const float *new_xyz,
                                   const float *xyz, int64_t *idx, void *stream);
/* src/group_points.cpp:4-6 */
int hitadv_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                        const int32_t *idx, float *out, void *stream);
/* src/group_points.cpp:8-10; grad_points fully overwritten (deterministic). */
int hitadv_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                             const int32_t *idx, float *grad_points, void *stream);
/* src/interpolate.cpp:4-5 */
int hitadv_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                    int32_t *idx, void *stream);
/* src/interpolate.cpp:6-8 */
int hitadv_three_interpolate(int b, int c, int m, int n, const float *points, const int32_t *idx,
                             const float *weight, float *out, void *stream);
/* src/interpolate.cpp:9-12; grad_points fully overwritten (deterministic). */
int hitadv_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                  const int32_t *idx, const float *weight, float *grad_points,
                                  void *stream);

/* ------------------------------------------------------------------ victim-side helper
 * Backward of "shared linear layer -> max over points" (model/feature_models.py:126-127 conv3+bn3 then
 * torch.max(x, 2), and the same pattern in STN3d/STNkd :164-165, :209-210):
 *   dX[b,n,:] = sum_{j : idx[b,j]==n} dg[b,j] * W[j,:]
 * dg[B,Cout], W[Cout,Cin] (BatchNorm already folded), idx[B,Cout] int64 arg-max over points,
 * act_out[B,Cout] (optional, NULL = none): the ReLU'd forward output; gradient passes only where it is > 0.
 * dX[B*N,Cin] fully overwritten.  Cin <= 512.  Deterministic (fixed order, no atomics). */
int hitadv_linear_max_bwd(const float *dg, const float *W, const int64_t *idx, const float *act_out, int B, int N,
                          int Cout, int Cin, float *dX, void *stream);

/* g[b,c] = act(max_n y[b,n,c] + bias[c]) and the arg-max (lowest n on ties) for points-major activations
 * y[B,N,C], C % 4 == 0, y 16-byte aligned; bias may be NULL, act = ReLU when relu != 0 (max, the bias add and
 * ReLU commute).  Replaces conv bias + bn + (relu) + torch.max(x, 2) of model/feature_models.py:126-127,164-165,
 * 209-210 in the attack-time victim view.  part_val / part_idx: scratch of hitadv_max_over_points_scratch(B,C). */
int hitadv_max_over_points(const float *y, int B, int N, int C, const float *bias, int relu, float *part_val,
                           int32_t *part_idx, float *out, int64_t *idx, void *stream);
int64_t hitadv_max_over_points_scratch(int B, int C);

/* Fused shared linear layer + max over points on the f32 matrix cores (MFMA 32x32x2 f32, exact f32):
 *   out[b,c] = act(max_n (X[b,n,:] . Wt[:,c]) + bias[c]),  idx[b,c] = first arg-max point,
 * i.e. `conv3 -> bn3 -> (relu) -> torch.max(x, 2)` of model/feature_models.py:173-175 (STN3d), :215-217 (STNkd)
 * and :139-140 (PointNetEncoder) with the BatchNorm folded into (Wt, bias).  The [B*N,Cout] activation is
 * never materialised.  X [B*N,Cin] points-major, Wt [Cin,Cout]; Cin in {64,128}, Cout % 64 == 0.
 * part_val / part_idx: scratch of hitadv_linear_max_fwd_scratch(B,N,Cout) entries each.
 * tickets: B * ceil(Cout/256) int32 that the caller ZEROES ONCE (every call leaves them at zero; calls that may run
 * concurrently need their own): the point splits of a cloud are then merged by the last block to finish, inside the
 * same launch.  NULL = merge with a second, tiny launch. */
int hitadv_linear_max_fwd(const float *X, const float *Wt, const float *bias, int B, int N, int Cin, int Cout,
                          int relu, float *part_val, int32_t *part_idx, float *out, int64_t *idx, int32_t *tickets,
                          void *stream);
int64_t hitadv_linear_max_fwd_scratch(int B, int N, int Cout);

/* The same operator on the bf16 matrix cores at fp32 accuracy (csrc/victim_bf3.hip): every fp32 operand is carried as
 * three bf16 pieces that sum to it EXACTLY (a = a1 + a2 + a3, 8 significant bits each) and the product is evaluated as
 * the six leading cross terms in an fp32 accumulator -- the dropped terms are below 2^-24 of the product, i.e. below the
 * rounding of an fp32 GEMM; nothing is rounded to bf16 precision.  2.67x less matrix time than the f32 MFMA form.
 * W3 = the three bf16 pieces of the layer's weight W [Cout,Cin] (row-major, one row per output channel; Cout % 32 == 0,
 * Cin % 32 == 0), 3*Cout*Cin uint16 in the kernel's fragment order [piece][c/16][k/32][(k%32)/8][c%16][k%8], made once
 * per attack by hitadv_split_weights_bf16x3; 16-byte aligned.  Scratch / tickets as for hitadv_linear_max_fwd
 * (tickets must not be NULL). */
int hitadv_split_weights_bf16x3(const float *W, int Cout, int Cin, uint16_t *W3, void *stream);
/* `blocks`: how many workgroups the kernel spreads over (0 = default: one per CU, or HITADV_V1_CUS; else 8..256): with
 * four attacks in flight, 128 -- half the chip, for twice as long -- leaves the other half to the other streams' short
 * kernels (28.1 instead of 27.0 clouds/s).  Results do not depend on it; the scratch size does: ask with the same value. */
int hitadv_linear_max_fwd_bf16x3(const float *X, const uint16_t *W3, const float *bias, int B, int N, int Cin, int Cout,
                                 int relu, int blocks, float *part_val, int32_t *part_idx, float *out, int64_t *idx,
                                 int32_t *tickets, void *stream);
int64_t hitadv_linear_max_fwd_bf16x3_scratch(int B, int N, int Cout, int blocks);

/* The same operator on the fp16 matrix cores with TWO pieces per operand and THREE products per useful one (half the matrix
 * time of the bf16 form): a = a1 + 2^-11 a2 + ra with a1 = fp16(a), a2 = fp16(2^11 (a - a1)) (round to nearest; a - a1 is exact
 * in fp32) and |ra| <= 2^-24 |a|;  a.b = a1.b1 + 2^-11 (a1.b2 + a2.b1) + terms <= 2^-24 |a.b| each -- fp32's own unit roundoff --
 * with every fp16 x fp16 product exact in the fp32 accumulators (one accumulator set per power of two, joined per tile).
 * Error against float64 measured next to the f32-MFMA kernel's: tests/test_gpu_kernels.py::test_linear_max_fwd_f16x2_*.
 * W2 = hitadv_split_weights_f16x2(Wr): 2*Cout*Cin uint16 (fp16 bits) in the fragment order of the bf16 form.
 * fp16 ends at 65504: `range_flag` (device int32, may be NULL; zero it before the first call) is set to 1 by either entry
 * point when an operand lies beyond that (or is NaN); results are then meaningless and the caller must fall back to the
 * bf16x3 / f32 form.  Scratch: hitadv_linear_max_fwd_bf16x3_scratch with the same `blocks`. */
int hitadv_split_weights_f16x2(const float *W, int Cout, int Cin, uint16_t *W2, int32_t *range_flag, void *stream);
/* ... and with X already in packed pieces (Xp [B*N, Cin] uint32, see hitadv_pointnet_rowmlp_fwd mode 2). */
int hitadv_linear_max_fwd_f16x2_packed(const uint32_t *Xp, const uint16_t *W2, const float *bias, int B, int N, int Cin, int Cout,
                                       int relu, int blocks, float *part_val, int32_t *part_idx, float *out, int64_t *idx,
                                       int32_t *tickets, void *stream);
int hitadv_linear_max_fwd_f16x2(const float *X, const uint16_t *W2, const float *bias, int B, int N, int Cin, int Cout,
                                int relu, int blocks, float *part_val, int32_t *part_idx, float *out, int64_t *idx,
                                int32_t *tickets, int32_t *range_flag, void *stream);

/* ------------------------------------------------------------------ PointNet victim, attack-time view
 * The eval.py victim (model/feature_models.py:71-230: PointNetFeatureModel = PointNetEncoder + STN3d + STNkd) in
 * eval mode with every BatchNorm folded into the layer in front of it, as four building blocks that together
 * with hitadv_linear_max_fwd / hitadv_linear_max_bwd give logits = f(x) and d logits / d x without any
 * rocBLAS / MIOpen call.  Activations are points-major [B*N,C]; x and its gradient are [B,3,N] as the attack
 * holds them.  Wt = [Cin,Cout] (forward operand), Wr = [Cout,Cin] (backward operand).  Matrix products: see `mode` below.
 *
 * hitadv_pointnet_rowmlp_fwd: the shared per-point layers in front of a 128->1024 layer, 64 points per block.
 *   stage 0 (STN3d, :168-171)       x -> o0 = relu(x W0 + b0) [.,64] -> o2 = relu(o0 W2 + b2) [.,128]
 *   stage 1 (encoder + STNkd, :119-128, :210-213)
 *                                   xp = x @ T[b] (T [B,3,3]) -> o0 = relu(xp W0 + b0) -> o1 = relu(o0 W1 + b1) [.,64]
 *                                   -> o2 = relu(o1 W2 + b2)
 *   stage 2 (encoder, :129-137)     o0 = hin @ T[b] (T [B,64,64]) -> o2 = relu(o0 W2 + b2)
 * Unused pointers of a stage may be NULL; so may xp (stage 1) and o0 (stage 2) when the caller does not need them.
 * `mode` (all three forward entry points and the backward one): 0 = the products on the f32 matrix cores (an exact fp32 FMA
 * chain per output), 1 = on the fp16 matrix cores with every operand as two fp16 pieces and three exact products per useful
 * one (the scheme of hitadv_linear_max_fwd_f16x2: 5.3x less matrix time, errors at fp32's unit roundoff; the 3 -> 64 and
 * 64 -> 3 layers and the transforms' 3x3 products stay exact f32 on the VALU); 2 = mode 1 with the 128-wide activation o2 / A2
 * held as PACKED PIECES, one 32-bit word per value (fp16 hi | fp16 lo << 16, lo = the residual scaled by 2^11): what
 * hitadv_linear_max_fwd_f16x2_packed consumes without splitting anything (the split is otherwise redone by each of its four
 * column-group blocks) and what the backward chain reads as that layer's ReLU mask (word != 0).  range_flag (forward entry
 * points, may be NULL) is raised by a value beyond fp16's range in mode 2. */
int hitadv_pointnet_rowmlp_fwd(int stage, const float *x, const float *T, const float *hin, const float *W0,
                               const float *b0, const float *W1, const float *b1, const float *W2, const float *b2,
                               float *xp, float *o0, float *o1, float *o2, int B, int N, int mode, int32_t *range_flag,
                               void *stream);
/* Number of 64-point tiles per cloud = leading dimension of the dTpart scratch below. */
/* Stage 0 whose input IS HiT-ADV's deformation (hitadv_deform_fwd, ShapeAttack/HiT_ADV.py:160-175) of `ori`: every block
 * deforms its 64 points itself (C <= 256 centres), writes them to adv [B,3,N] and 1 / sum_j k to inv_den [B,N] -- what
 * hitadv_deform_fwd writes, bit for bit -- and goes on with them as hitadv_pointnet_rowmlp_fwd(stage = 0, x = adv). */
int hitadv_pointnet_rowmlp_fwd_deform(const float *ori, const float *central, const float *perturb, const float *sigma,
                                      int C, float *adv, float *inv_den, const float *W0, const float *b0, const float *W2,
                                      const float *b2, float *o0, float *o2, int B, int N, int mode, int32_t *range_flag,
                                      void *stream);
/* Stage 1 with the input transform evaluated inside: T3[b] = F5[b,:256] @ W6[256,9] + b6 (STN3d's last layer, :186-190,
 * the identity folded into b6) is computed by every block of the cloud (2304 multiply-adds, fixed order) and written to
 * Tout [B,9] for the backward pass; everything else as hitadv_pointnet_rowmlp_fwd(stage = 1, T = Tout). */
int hitadv_pointnet_rowmlp_fwd_stn(const float *x, const float *F5, const float *W6, const float *b6, float *Tout,
                                   const float *W0, const float *b0, const float *W1, const float *b1, const float *W2,
                                   const float *b2, float *xp, float *o0, float *o1, float *o2, int B, int N, int mode,
                                   int32_t *range_flag, void *stream);
int64_t hitadv_pointnet_rowmlp_tiles(int N);
/* Which kernel serves modes 1 / 2 of the three forward entry points above: 0 = the STREAMING form (round 5: a
 * workgroup takes a run of tiles, weights split once per workgroup, transposed products so that pieces and results leave a
 * lane four channels at a time, every result row-wise through an LDS tile); 1 = one 64-point tile per workgroup (the round-3
 * kernel); 2 (default) = streaming, except stage 0 with the deformation inside, which keeps the tile kernel (faster there).
 * All write the same bits; the switch exists so that a test can say so and a profile can compare them.  Process-wide
 * (environment: HITADV_ROWMLP_FORM, read when the library is loaded by hit_adv_amd._lib); returns the previous value; any
 * other argument only reads. */
int hitadv_pointnet_rowmlp_form(int form);
/* Input-gradient chain of the same stages, starting at the max-pooled output of the stage's 128->Cout layer:
 * dg [B,Cout] is the gradient there, idx [B,Cout] the arg-max point of every channel (hitadv_linear_max_fwd),
 * gmask [B,Cout] the ReLU'd forward output (NULL when that layer has no ReLU), W3r [Cout,128] the layer's weights.
 * Each 64-point block gathers  dA2[n,:] = sum_{j: idx[b,j]==n} dg[b,j] * W3r[j,:]  for its own points in LDS (fixed
 * order, no atomics; what hitadv_linear_max_bwd writes to HBM), then runs the chain; A2 / A1 / H1 are the activations
 * saved by the forward (ReLU masks).  Cout <= 1024, N <= 65535.
 *   stage 2: out = dH1 [B*N,64] = (dA2 . [A2>0]) W2r @ T^T;  dTpart [B,tiles,64,64] = per-tile  h1^T @ (.)
 *   stage 1: through t2, t1, + dH1in, e1's ReLU, e1 -> g [.,3];  out = dPts [B,3,N] = g @ T^T;
 *            dTpart [B,tiles,9] = per-tile x^T @ g
 *   stage 0: through s2, s1 -> [.,3], + dPin;  out = dX [B,3,N]
 * The per-tile partials are summed in tile order by hitadv_sum_partials (deterministic).
 * Row sparsity: a max-pool sends each channel's gradient to one point, and the chain is row-wise, so only the points
 * that receive something are processed.  pres_out [B,tiles] (or NULL) receives, per tile, the bit set of those points;
 * pres_in [B,tiles] is the previous stage's pres_out and says which rows of dH1in (stage 1) / dPin (stage 0) are
 * non-zero -- stage 2 writes ONLY the rows of dH1 in its pres_out, and stage 1 reads dH1in only at pres_in rows.
 * pres_in == NULL: every row of the incoming gradient may be non-zero (dense behaviour; dH1in fully defined).
 * "tiles" here = hitadv_pointnet_rowmlp_bwd_tiles(N, words): a backward block covers `words` 64-point words (1 or 2; mode 0:
 * 1).  The work of a block is set by the ~10 of 64 points that receive gradient, so the larger tile halves the number of
 * latency-bound blocks; hitadv_pointnet_rowmlp_bwd_words(B, N, mode) recommends 2 where one word would be more than one round
 * of blocks on the chip.  pres_in / pres_out are [B,tiles,words] (one bit set per word: the same 64-bit words in both forms).
 * overflow: int32 [B,tiles] scratch, required when words = 2 (may be NULL otherwise): the compacted rows of a tile live in
 * 64-row LDS tiles, so a two-word tile with more than 64 winning points is marked there by the first of the entry point's
 * two launches and taken word by word by the second (which returns at once for every other tile). */
int64_t hitadv_pointnet_rowmlp_bwd_tiles(int N, int words);
int hitadv_pointnet_rowmlp_bwd_words(int B, int N, int mode);
int hitadv_pointnet_rowmlp_bwd(int stage, const float *dg, const float *gmask, const int64_t *idx, const float *W3r,
                               int Cout, const float *A2, const float *W2r, const float *A1, const float *W1r,
                               const float *H1, const float *dH1in, const float *W0r, const float *T, const float *x,
                               const float *dPin, float *dTpart, float *out, const uint64_t *pres_in,
                               uint64_t *pres_out, int32_t *overflow, int words, int B, int N, int mode, void *stream);
/* The same with scratch for the second launch of the two-word form (words = 2): dTfix [B, 2 * tiles, 9] (stage 1) or
 * [B, 2 * tiles, 4096] (stage 2), unused (may be NULL) in stage 0.  With it the tiles the first launch leaves (more than 64
 * winning points in 128) are done by the one-word kernel, two blocks per tile, and their partials added in word order --
 * instead of by a two-pass instantiation that spills (round 5: 6x faster on surface-like clouds, where half the tiles take
 * this path; nothing changes where none does).  dTfix == NULL: hitadv_pointnet_rowmlp_bwd. */
int hitadv_pointnet_rowmlp_bwd_fix(int stage, const float *dg, const float *gmask, const int64_t *idx, const float *W3r,
                                   int Cout, const float *A2, const float *W2r, const float *A1, const float *W1r,
                                   const float *H1, const float *dH1in, const float *W0r, const float *T, const float *x,
                                   const float *dPin, float *dTpart, float *out, const uint64_t *pres_in,
                                   uint64_t *pres_out, int32_t *overflow, int words, int B, int N, int mode, float *dTfix,
                                   void *stream);
/* out[b,m] = (extra ? extra[b,m] : 0) + sum_t part[b,t,m], ascending t. */
int hitadv_sum_partials(const float *part, const float *extra, int B, int T, int M, float *out, void *stream);
/* out[B,NOUT] = act(in'[B,K] @ Wt[K,NOUT] + bias), in' = in gated by (mask > 0) when mask != NULL (the backward of a
 * ReLU'd layer: in = dOut, mask = the saved output, Wt = that layer's Wr).  bias may be NULL.  The fc1/fc2/fc3
 * stacks of :176-186 (STN3d), :218-228 (STNkd) and :88-91 (classifier head).
 * scratch: hitadv_fc_layer_scratch_floats(B,K,NOUT) floats that the caller ZEROES ONCE (its first 16384 words
 * are the split-K tickets, which every call leaves at zero again); calls that may run concurrently need their own. */
int hitadv_fc_layer(const float *in, const float *mask, const float *Wt, const float *bias, int B, int K, int NOUT,
                    int relu, float *out, float *scratch, void *stream);
int64_t hitadv_fc_layer_scratch_floats(int B, int K, int NOUT);
/* The same layer with its input produced on the way in:  in[b,k] = sum_j (sum_t pre[b,t,j]) * Wpre[j,k]  (pre [B,T,J]
 * partials summed in ascending t, Wpre [J,K], J <= 64), then gated by mask as above.  One launch for what would be three
 * in the backward stacks: the sum over the tiles' dT partials, the stack's first layer (K = 9 or 40: fc3 of :186 / :91
 * backwards) and its second.  Scratch as for hitadv_fc_layer(B,K,NOUT). */
int hitadv_fc_layer_pre(const float *pre, int T, int J, const float *Wpre, const float *mask, const float *Wt,
                        const float *bias, int B, int K, int NOUT, int relu, float *out, float *scratch, void *stream);

/* ------------------------------------------------------------------ DGCNN victim: EdgeConv neighbour reduction
 * get_graph_feature + Conv2d(1x1) + BatchNorm2d + LeakyReLU + max over the k neighbours (model/dgcnn_cls.py:16-43,
 * 93-112) with the convolution split algebraically into two per-point products U = X (s Wa)^T, V = X (s (Wb - Wa))^T + t
 * (W = [Wa | Wb] acting on [x_j - x_i ; x_i], BatchNorm scale s / shift t folded):
 *   out[b,i,c] = lrelu(V[b,i,c] + max_{j in idx[b,i,:]} U[b,j,c]),  arg[b,i,c] = the winning j.
 * out, arg [B,N,C] points-major, C % 4 == 0; U and V are [B,N,C] views with a row pitch of ld floats (ld = C for two
 * separate tensors; ld = 2C with V = U + C when one product X [Wu | Wv] produced both); idx [B,N,k] int64; the
 * [B,2C,N,k] edge tensor is never built. */
int hitadv_edge_max_fwd(const float *U, const float *V, int ld, const int64_t *idx, int B, int N, int C, int k,
                        float slope, float *out, int32_t *arg, void *stream);
/* dV = dout * lrelu'(out);  dU[b,j,c] = sum_{i: arg[b,i,c]==j} dV[b,i,c], the terms added in ascending i: the neighbour
 * table is reversed on the device first (integer atomics only) and every (j,c) gathers its terms, so the result is the
 * same bit pattern on every run (torch's index_select backward, which the reference goes through, is not).
 * idx: the table the forward was given; dU, dV: [B,N,C] views with a row pitch of ldg floats (as U, V in the forward);
 * scratch: hitadv_edge_max_bwd_scratch_ints(B,N,k) int32, contents irrelevant. */
int hitadv_edge_max_bwd(const float *dout, const float *out, const int32_t *arg, const int64_t *idx, int B, int N, int C,
                        int k, float slope, float *dU, float *dV, int ldg, int32_t *scratch, void *stream);
int64_t hitadv_edge_max_bwd_scratch_ints(int B, int N, int k);

/* DGCNN's pooling after the embedding layer (model/dgcnn_cls.py:101-106: LeakyReLU, adaptive_max_pool1d and
 * adaptive_avg_pool1d over the points, concatenated), from the layer's PRE-activation Z [B,N,C] points-major, C % 64 == 0:
 *   out[b,c] = max_i lrelu(Z[b,i,c]),  out[b,C+c] = mean_i lrelu(Z[b,i,c]),  arg[b,c] = first i attaining the max.
 * One pass over Z, fixed summation order.  Backward: dZ[b,i,c] = lrelu'(Z) * (g[b,C+c]/N + (i == arg[b,c]) g[b,c]). */
int hitadv_lrelu_pool_fwd(const float *Z, int B, int N, int C, float slope, float *out, int32_t *arg, void *stream);
int hitadv_lrelu_pool_bwd(const float *Z, const float *g, const int32_t *arg, int B, int N, int C, float slope, float *dZ,
                          void *stream);

/* First shared layer of a sample-and-group block (PointNet++ set abstraction, model/pointnet2_utils.py:161-205; PCT
 * Local_op, model/pct_utils.py:98-140) with the 1x1 convolution split over its two inputs, W [x_j - c_i ; c_i] + t =
 * Wa x_j + ((Wb - Wa) c_i + t):  H[b,i,s,:] = relu(U[b, idx[b,i,s], :] + V[b,i,:]).
 * U [B,N,C] (one product over the N points), V [B,S,C] (one over the S centres), idx [B,S,ns] int64 (entries outside
 * [0,N) give a zero row), H [B,S,ns,C]; C % 4 == 0.  The [B,S,ns,Cin] gather / subtract / concat never exists. */
int hitadv_group_add_relu_fwd(const float *U, const float *V, const int64_t *idx, int B, int N, int S, int ns, int C,
                              float *H, void *stream);
/* dV[b,i,:] = sum_s dH[b,i,s,:] * mask,  dU[b,j,:] = sum_{(i,s): idx[b,i,s]==j} dH[b,i,s,:] * mask in ascending (i,s),
 * mask = [U + V > 0] recomputed (H is not needed); no float atomics.  ns <= 64.
 * scratch: hitadv_group_add_relu_bwd_scratch_ints(B,N,S,ns) int32, 8-byte aligned, contents irrelevant. */
int hitadv_group_add_relu_bwd(const float *dH, const float *U, const float *V, const int64_t *idx, int B, int N, int S,
                              int ns, int C, float *dU, float *dV, int32_t *scratch, void *stream);
int64_t hitadv_group_add_relu_bwd_scratch_ints(int B, int N, int S, int ns);

/* ------------------------------------------------------------------ merged launches of one HiT-ADV iteration
 * (csrc/iteration.hip, csrc/regulariser.hip, csrc/deform.hip).  Results are those of the entry points they merge, bit for
 * bit; they exist because at one attack in flight an iteration is a chain of ~40 dependent launches of 4-7 us each.
 *
 * hitadv_iteration_head = hitadv_best_update + hitadv_adv_loss (both read only the logits, ShapeAttack/HiT_ADV.py:
 *   186-217 and util/adv_utils.py:18-85).  scratch: hitadv_iteration_head_scratch_floats(B) floats, ZEROED ONCE by the
 *   caller (it holds a ticket that every call leaves at zero).
 * hitadv_regulariser_fwd_fused = hitadv_regulariser_fwd in one launch; the LAST float of its scratch is a ticket and
 *   must be zero on first use.
 * hitadv_deform_bwd_partials = the deformation backward without its reduce launch: partials [B,nslab,4,C],
 *   nslab = hitadv_deform_bwd_slabs(N);  hitadv_adam_step_partials = hitadv_adam_step_sum that sums those partials itself
 *   (ascending slab = the reduce launch's order) before the Adam update and the projection. */
int hitadv_iteration_head(const float *logits, const int64_t *label, const float *perturb, const float *sigma,
                          const float *adv, int B, int num_class, int N, int C, float *bestdist, int64_t *bestscore,
                          float *o_bestdist, int64_t *o_bestscore, float *o_bestattack, int64_t *pred_out,
                          float *dist_val_out, int32_t *iter_counter, int kind, float kappa, float *loss, float *dlogits,
                          float *scratch, void *stream);
/* hitadv_iteration_head and hitadv_regulariser_fwd_fused in ONE launch (blocks 0..B-1 / B..2B-1): the regularisers' forward
 * pass needs perturb, sigma and the deformed cloud only, nothing the victim produces.  Arguments: those of the two entry
 * points (scratch = the head's, reg_scratch = the regularisers'; both zeroed once by the caller); same results, bit for bit.
 * feat != NULL: `logits` is an OUTPUT -- the classifier's last layer (feature_models.py:91) is evaluated by the head blocks,
 * logits[b,:] = feat[b,:feat_dim] @ Wlog[feat_dim,num_class] + blog (feat_dim <= 256, num_class <= 64; fixed summation
 * order: 64 features per wave in ascending order, the four waves in order, the bias last), and written there. */
int hitadv_iteration_head_reg(const float *logits, const int64_t *label, const float *perturb, const float *sigma,
                              const float *adv, int B, int num_class, int N, int C, float *bestdist, int64_t *bestscore,
                              float *o_bestdist, int64_t *o_bestscore, float *o_bestattack, int64_t *pred_out,
                              float *dist_val_out, int32_t *iter_counter, int kind, float kappa, float *loss,
                              float *dlogits, float *scratch, const float *ori, const float *hide_ref,
                              const float *scale_const, float cd_w, float ker_w, float hide_w, float min_sigm,
                              float max_sigm, float *reg_scratch, float *dist_loss, float *scaled_loss, const float *feat,
                              const float *Wlog, const float *blog, int feat_dim, void *stream);
int64_t hitadv_iteration_head_scratch_floats(int B);
int hitadv_regulariser_fwd_fused(const float *perturb, const float *sigma, const float *adv, const float *ori,
                                 const float *hide_ref, const float *scale_const, int B, int N, int C, float cd_w,
                                 float ker_w, float hide_w, float min_sigm, float max_sigm, float *scratch,
                                 float *dist_loss, float *scaled_loss, void *stream);
int hitadv_deform_bwd_partials(const float *ori, const float *central, const float *perturb, const float *sigma,
                               const float *adv, const float *inv_den, const float *g_adv, int B, int N, int C,
                               float *partials, void *stream);
int64_t hitadv_deform_bwd_slabs(int N);
int hitadv_adam_step_partials(float *perturb, float *sigma, const float *partials, int nslab, const float *g_perturb2,
                              const float *g_sigma2, float *m_perturb, float *v_perturb, float *m_sigma, float *v_sigma,
                              int B, int C, float lr_perturb, float lo_perturb, float hi_perturb, float lr_sigma,
                              float lo_sigma, float hi_sigma, const int32_t *step, void *stream);
/* The same two launches WITHOUT hitadv_regulariser_bwd_add between the victim and them: the regularisers' backward terms are
 * closed-form in what these kernels read anyway, so they evaluate them on the fly (same expressions, same order, same
 * bits).  _reg forms: g_victim = the victim's gradient at the deformed cloud (the regularisers' term is added inside);
 * reg_scratch = the scratch of the forward pass (hitadv_regulariser_fwd_fused / hitadv_iteration_head_reg). */
int hitadv_deform_bwd_partials_reg(const float *ori, const float *central, const float *perturb, const float *sigma,
                                   const float *adv, const float *inv_den, const float *g_victim,
                                   const float *reg_scratch, float cd_w, int B, int N, int C, float *partials, void *stream);
/* Both of them in ONE launch: the blocks of a cloud draw a ticket after publishing their partials and the last one to
 * arrive runs the Adam step for the cloud's centres (tickets: int32 [B], zeroed once by the caller, left at zero). */
int hitadv_deform_bwd_adam_reg(const float *ori, const float *central, float *perturb, float *sigma, const float *adv,
                               const float *inv_den, const float *g_victim, const float *hide_ref,
                               const float *reg_scratch, float cd_w, float ker_w, float hide_w, float min_sigm,
                               float max_sigm, float *m_perturb, float *v_perturb, float *m_sigma, float *v_sigma, int B,
                               int N, int C, float lr_perturb, float lo_perturb, float hi_perturb, float lr_sigma,
                               float lo_sigma, float hi_sigma, const int32_t *step, float *partials, int32_t *tickets,
                               void *stream);
int hitadv_adam_step_partials_reg(float *perturb, float *sigma, const float *partials, int nslab, const float *hide_ref,
                                  const float *reg_scratch, float cd_w, float ker_w, float hide_w, float min_sigm,
                                  float max_sigm, float *m_perturb, float *v_perturb, float *m_sigma, float *v_sigma, int B,
                                  int C, float lr_perturb, float lo_perturb, float hi_perturb, float lr_sigma,
                                  float lo_sigma, float hi_sigma, const int32_t *step, void *stream);

/* ------------------------------------------------------------------ last shared layer of a sample-and-group block + max
 * PointNet++'s set abstraction (model/pointnet2_utils.py:197-201: conv -> bn -> relu -> torch.max(new_points, 2)) and PCT's
 * Local_op (model/pct_cls.py:14-24) end in a shared layer over [G = B*npoint groups, ns neighbours] rows followed by a max
 * over the neighbours.  Fused, for ns in {32, 64} and (Cin, Cout) in {(64,128), (128,128), (128,256)}
 * (hitadv_group_linear_max_supported):
 *   fwd: out[g,c] = relu(max_j x[g*ns + j,:] . W[c,:] + bias[c]), arg[g,c] = the lowest such j (int32); X [G*ns, Cin] fp32,
 *        W2 = hitadv_split_weights_f16x2(Wr [Cout,Cin]) -- V1's fp16x2 scheme (two pieces per operand, three exact products,
 *        fp32 accumulators), so the [G*ns, Cout] activation, its ReLU pass and its max pass never exist;
 *   bwd: dX[g*ns + j,:] = sum_{c: arg[g,c] == j, out[g,c] > 0} dOut[g,c] W[c,:]; Wb2 = hitadv_split_weights_f16x2(Wt [Cin,Cout])
 *        (the same layout with the roles of the two dimensions swapped); every row of dX is written.
 * range_flag as for hitadv_linear_max_fwd_f16x2 (may be NULL). */
int hitadv_group_linear_max_supported(int Cin, int Cout, int ns);

/* ---- rows_linear: Y[rows,Cout] = act(X[rows,Cin] W^T + bias) for very many rows of a narrow layer --------------------
 * The middle shared layer of a sample-and-group block applied to the grouped rows (model/pointnet2_utils.py:197-201:
 * `new_points = F.relu(bn(conv(new_points)))` on [B, C, nsample, npoint]; 0.5-1 M rows of 64 / 128 channels at cfg4) and, with the
 * pieces of the transposed weights, no bias and relu = 0, its input gradient dX = dY W.  fp16x2 arithmetic (two fp16 pieces per
 * operand, three exact products per useful one, fp32 accumulation: fp32-accurate), bound by the read of X and the write of Y.
 *   X [rows,Cin] fp32, 16-byte aligned;  W2 = hitadv_split_weights_f16x2(W [Cout,Cin]);  bias [Cout] or NULL;  relu: 0 / 1;
 *   Y [rows,Cout] fp32, 16-byte aligned;  range_flag: int32[1] on the device or NULL, raised (never cleared) when an entry of X is
 *   not finite or too large for the split (|x| >= 65520).
 * Supported: Cin, Cout in {64, 128} (hitadv_rows_linear_supported); anything else: HITADV_E_ARG. */
int hitadv_rows_linear_supported(int Cin, int Cout);
int hitadv_rows_linear(const float *X, const uint16_t *W2, const float *bias, int64_t rows, int Cin, int Cout, int relu, float *Y,
                       int32_t *range_flag, void *stream);

/* The two together: Y[(b,i,s), :] = act(relu(U[b, idx[b,i,s], :] + V[b,i,:]) W^T + bias) -- hitadv_group_add_relu_fwd followed by
 * hitadv_rows_linear without the [B,S,ns,C] activation between them (same bits as the two calls).  U [B,N,C], V [B,S,C], idx
 * [B,S,ns] int64, W2 = hitadv_split_weights_f16x2(W [Cout,C]), Y [B*S*ns, Cout].  Supported (hitadv_group_add_relu_linear_supported):
 * C, Cout in {64, 128}, ns in {16, 32, 64}, S * ns a multiple of 64; anything else: HITADV_E_ARG. */
int hitadv_group_add_relu_linear_supported(int C, int Cout, int S, int ns);
int hitadv_group_add_relu_linear(const float *U, const float *V, const int64_t *idx, int B, int N, int S, int ns, int C,
                                 const uint16_t *W2, const float *bias, int Cout, int relu, float *Y, int32_t *range_flag, void *stream);
int hitadv_group_linear_max_fwd(const float *X, const uint16_t *W2, const float *bias, int64_t G, int ns, int Cin, int Cout,
                                float *out, int32_t *arg, int32_t *range_flag, void *stream);
int hitadv_group_linear_max_bwd(const float *dOut, const float *out, const int32_t *arg, const uint16_t *Wb2, int64_t G, int ns,
                                int Cin, int Cout, float *dX, int32_t *range_flag, void *stream);
/* ... with dX also gated by (xmask > 0), xmask [G*ns, Cin] = the layer's input when that is the ReLU output of the shared layer in
 * front (model/pointnet2_utils.py:197-200: the MLP is conv -> bn -> relu per layer): that layer's ReLU backward pass happens on
 * the way out of this kernel. */
int hitadv_group_linear_max_bwd_masked(const float *dOut, const float *out, const int32_t *arg, const uint16_t *Wb2, int64_t G,
                                       int ns, int Cin, int Cout, const float *xmask, float *dX, int32_t *range_flag,
                                       void *stream);

/* ------------------------------------------------------------------ fp32-accurate GEMMs on the fp16 matrix cores
 * The victims' wide 1x1 convolutions are GEMMs over [B*N] rows (PyTorch-ROCm: hipBLASLt's f32 GEMM, which runs at the f32
 * MFMA rate).  These entry points run them in V1's fp16x2 scheme (hitadv_linear_max_fwd_f16x2: two fp16 pieces per operand,
 * three exact products per useful one, fp32 accumulators; error against float64 below the f32 GEMM's) at 2-3x the speed.
 *   hitadv_split_rows_f16x2     W [N,K] fp32 (N output columns) -> Wp [2][N][K] fp16 pieces, once per weight
 *   hitadv_gemm_f16x2           C [M,N] = act((X . [mask > 0]) Wp^T + bias): X [M,K]; mask [M,K] or NULL (the input gradient
 *                               of a ReLU'd layer: X = dOut, mask = the saved output, Wp = pieces of Wt); bias [N] or NULL;
 *                               relu 0/1.  N % 128 == 0, K % 32 == 0 (hitadv_gemm_f16x2_supported).
 *   hitadv_linear_lrelu_pool_fwd   DGCNN's embedding layer with its activation and both poolings (model/dgcnn_cls.py:70-72,
 *                               101-104: conv5 -> bn5 -> LeakyReLU -> adaptive_max_pool1d | adaptive_avg_pool1d -> cat):
 *                               z = X Wp^T + bias per point (X [B*npts,Cin], BatchNorm folded into Wp / bias), out [B,2C] =
 *                               [max_p lrelu(z) | mean_p lrelu(z)], arg [B,C] int32 = the lowest point attaining the max,
 *                               bits [B*npts, C/32] = (z > 0): the [B*npts,C] activation never exists.  pmax / psum / parg:
 *                               scratch of hitadv_linear_lrelu_pool_scratch(B,npts,C) elements each.
 *   hitadv_linear_lrelu_pool_bwd   dX [B*npts,Cin] = G Wtp^T with G[p,c] = s(z[p,c]) (gout[b,C+c] / npts + [arg[b,c] == p]
 *                               gout[b,c]) rebuilt on the fly from bits / arg (s = 1 or the slope); Wtp = pieces of Wt [Cin,C].
 * range_flag as for hitadv_linear_max_fwd_f16x2 (may be NULL). */
/*   hitadv_group_linear_max_g16_*   hitadv_group_linear_max_fwd / _bwd_masked (below) for the widths its register-resident form
 *                               does not cover (PCT's second Local_op: 256 -> 256 over 32 neighbours, model/pct_cls.py:14-24), on
 *                               this GEMM core: the forward epilogue reduces max / arg-max over each group of ns rows inside the
 *                               wave that holds them; the backward operand is rebuilt from arg and dm = dOut gated by out > 0
 *                               ([G,Cout], the caller's one small element-wise pass); xmask [G*ns,Cin] or NULL gates dX as in
 *                               hitadv_group_linear_max_bwd_masked.  Wp / Wtp: hitadv_split_rows_f16x2 of Wr [Cout,Cin] / Wt. */
int hitadv_group_linear_max_g16_supported(int Cin, int Cout, int ns);
int hitadv_group_linear_max_g16_fwd(const float *X, const uint16_t *Wp, const float *bias, int64_t G, int ns, int Cin, int Cout,
                                    float *out, int32_t *arg, int32_t *range_flag, void *stream);
int hitadv_group_linear_max_g16_bwd(const float *dm, const int32_t *arg, const uint16_t *Wtp, int64_t G, int ns, int Cin, int Cout,
                                    const float *xmask, float *dX, int32_t *range_flag, void *stream);
int hitadv_gemm_f16x2_supported(int N, int K);
int hitadv_split_rows_f16x2(const float *W, int N, int K, uint16_t *Wp, int32_t *range_flag, void *stream);
int hitadv_gemm_f16x2(const float *X, const float *mask, const uint16_t *Wp, const float *bias, int64_t M, int N, int K, int relu,
                      float *C, int32_t *range_flag, void *stream);
int64_t hitadv_linear_lrelu_pool_scratch(int B, int npts, int C);
int hitadv_linear_lrelu_pool_fwd(const float *X, const uint16_t *Wp, const float *bias, int B, int npts, int Cin, int C, float slope,
                                 float *pmax, float *psum, int32_t *parg, uint32_t *bits, float *out, int32_t *arg,
                                 int32_t *range_flag, void *stream);
int hitadv_linear_lrelu_pool_bwd(const float *gout, const int32_t *arg, const uint32_t *bits, const uint16_t *Wtp, int B, int npts,
                                 int Cin, int C, float slope, float *dX, int32_t *range_flag, void *stream);

/* Small batched fp32 products for PCT's offset attention (model/pct_cls.py:111-139 and its backward: energy = q k, x_r = v attention):
 * C[b] [M,N] = op(A[b]) op(B[b]) for `batches` contiguous matrices; trans_a: A is stored [K,M] (else [M,K]); trans_b: B is stored
 * [N,K] (else [K,N]).  Exact fp32 FMA chains (k ascending) on the f32 matrix cores, 64 x 64 tiles: the library's batched GEMM
 * runs such a call on `batches` workgroups.  M, N multiples of 64, K a multiple of 32 (hitadv_bmm_f32_supported). */
int hitadv_bmm_f32_supported(int M, int N, int K);
int hitadv_bmm_f32(const float *A, const float *B, float *C, int batches, int M, int N, int K, int trans_a, int trans_b,
                   void *stream);

/* PCT's offset attention between its two batched products (model/pct_cls.py:127-131):
 *   A = softmax(E, dim=-1);  A = A / (1e-9 + A.sum(dim=1, keepdim=True))        E, A [B,N,N], colsum [B,N] = the sums
 * forward in two launches, backward (dE from dA, A, colsum; h [B,N] scratch) in two, every reduction in a fixed order.
 * torch: four element-wise passes forward, nine backward.  N a multiple of 64, N <= 1024. */
int hitadv_offset_attention_supported(int N);
int hitadv_offset_attention_fwd(const float *E, int B, int N, float *A, float *colsum, void *stream);
int hitadv_offset_attention_bwd(const float *dA, const float *A, const float *colsum, int B, int N, float *h, float *dE,
                                void *stream);

/* G independent attacks STACKED (HiT_ADV.attack_many on the PointNet engine: one victim pass over the G*B clouds): the three
 * launches around that pass for all G groups at once.  Every per-cloud argument is the group-0 pointer of a buffer that
 * holds the G groups' rows one after the other (B clouds each); every per-group scalar or scratch likewise, with the stride
 * of its size query: loss / dist_loss / scaled_loss / iter_counter / step 1 element, scratch
 * hitadv_iteration_head_scratch_floats(B), reg_scratch hitadv_regulariser_scratch_floats(B), partials
 * hitadv_deform_bwd_scratch_floats(B, N, C).  Wlog / blog are shared.  Group g's results are the bits of the un-stacked
 * entry point called on its rows (the kernels move their arguments to the group's rows and run the same per-group code). */
int hitadv_iteration_head_reg_stack(int G, const float *logits, const int64_t *label, const float *perturb,
                                    const float *sigma, const float *adv, int B, int num_class, int N, int C, float *bestdist,
                                    int64_t *bestscore, float *o_bestdist, int64_t *o_bestscore, float *o_bestattack,
                                    int64_t *pred_out, float *dist_val_out, int32_t *iter_counter, int kind, float kappa,
                                    float *loss, float *dlogits, float *scratch, const float *ori, const float *hide_ref,
                                    const float *scale_const, float cd_w, float ker_w, float hide_w, float min_sigm,
                                    float max_sigm, float *reg_scratch, float *dist_loss, float *scaled_loss,
                                    const float *feat, const float *Wlog, const float *blog, int feat_dim, void *stream);
int hitadv_deform_bwd_partials_reg_stack(int G, const float *ori, const float *central, const float *perturb,
                                         const float *sigma, const float *adv, const float *inv_den, const float *g_victim,
                                         const float *reg_scratch, float cd_w, int B, int N, int C, float *partials,
                                         void *stream);
int hitadv_adam_step_partials_reg_stack(int G, float *perturb, float *sigma, const float *partials, int nslab,
                                        const float *hide_ref, const float *reg_scratch, float cd_w, float ker_w,
                                        float hide_w, float min_sigm, float max_sigm, float *m_perturb, float *v_perturb,
                                        float *m_sigma, float *v_sigma, int B, int C, float lr_perturb, float lo_perturb,
                                        float hi_perturb, float lr_sigma, float lo_sigma, float hi_sigma,
                                        const int32_t *step, void *stream);

/* k nearest neighbours in feature space for DGCNN's dynamic graph (model/dgcnn_cls.py:7-13: topk of
 * -|x_i|^2 + 2 x_i.x_j - |x_j|^2), fused: the scores come off the f32 matrix cores tile by tile and go straight into
 * per-lane sorted lists -- no [B,N,N] matrix.  X [B,N,D] points-major (D in {64,128}, 16-byte aligned), xx [B,N] = |x|^2,
 * K <= 20.  idx [B,N,K] int64, closest first (the point itself), ties -> lower index. */
int hitadv_knn_features(const float *X, const float *xx, int B, int N, int D, int K, int64_t *idx, void *stream);
/* xx[row] = sum_k X[row,k]^2 for X [rows,D] fp32 (D a multiple of 4, X 16-byte aligned), evaluated in a fixed order (the same
 * bits every run): the squared norms hitadv_knn_features takes.  Host-side helper of model/dgcnn_cls.py:9 (`xx = torch.sum(x ** 2, ...)`). */
int hitadv_row_sqnorm(const float *X, int64_t rows, int D, float *xx, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HITADV_H */
