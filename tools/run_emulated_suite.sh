#!/bin/bash
# The `-m gpu` parity tests that reach NO matrix-instruction kernel, UNMODIFIED, on the CPU wave emulator (tests/native/emu_plugin.py: the
# kernels' own source compiled for x86-64 against tests/native/emu/simt_emu.hpp).  ~12 min on 8 cores; the summary lands in
# tests/golden/emulated_suite_report.txt.  No GPU needed -- and nothing here says anything about the hardware: see the plugin's header.
cd "$(dirname "$0")/.."
export PYTHONPATH=tests/native
K_KERNELS="test_pairwise_direct_bit_exact or test_pairwise_gram_forms_bit_exact or test_pairwise_generic_dim or test_nn_min_bit_exact_both_directions or test_nn_min_reference_arithmetic_bit_exact or test_nn_min_ties_take_lowest_index or test_nn_min_generic_dim_q1_shape or test_set_distance_modules_vs_reference_vectors or test_set_distance_modules_reference_arithmetic_equal_reference_vectors or test_nn_min_backward_matches_autograd_of_direct_matrix or test_knn_points_bit_exact or test_knn_points_sizes_and_both_selection_kernels or test_knn_points_gram_knn_form_bit_exact or test_knn_points_full_batch_with_ties_and_falling_distances or test_knn_points_vs_independent_float64_top_k or test_knn_points_heavy_ties_and_log_compaction or test_knn_points_duplicates_ragged_and_gather or test_knn_points_backward or test_knn_dist_operators_vs_reference_vectors or test_knn_dist_grad_vs_direct_form_oracle or test_deform_forward_backward_vs_reference_vectors or test_deform_identity_and_ragged_sizes or test_fps_from_start_vs_reference_vector or test_fps_from_start_bit_exact_sizes or test_fps_pct_bit_exact_sizes or test_fps_both_samplers_on_a_lattice_of_exact_ties or test_fps_pct_reproduces_the_reference_table or test_query_ball_point_victim_bit_exact or test_query_ball_point_victim_reproduces_the_reference_table or test_knn_points_square_distance_form_bit_exact or test_fps_ext_bit_exact or test_fps_ext_known_answers or test_ball_query_group_gather_bit_exact or test_native_gradients_and_interpolation or test_best_update_and_adam_match_host_logic or test_adam_step_sum_and_projection or test_adam_single_matches_torch_adam or test_fused_adv_losses_match_reference_modules or test_fused_regulariser_matches_torch_composition or test_topk_rows_bit_exact"
K_EDGES="not edge_max and not pointnet_engine and not index_tables"
K_ATTACK="test_hit_adv_follows_reference_trajectory[False or test_hit_adv_bookkeeping_over_ten_binary_steps[False or test_hit_adv_wide_configuration_vs_reference or test_cwknn_follows_reference_trajectory or test_cwuknn_follows_reference_trajectory or test_clip_operators_match_reference_vectors or test_cwperturb_follows_reference_trajectory or test_cwperturbt or test_cwaof or test_cw_family_follows_reference_trajectories or test_cwadd_family"
OUT=tests/golden/emulated_suite_report.txt
{
  echo "# GPU parity tests run UNMODIFIED on the CPU wave emulator (tools/run_emulated_suite.sh); commit $(git rev-parse --short HEAD), $(date -u +%F)"
  echo "# kernels built for the emulator: pairwise knn sampling grouping deform regulariser attack_state iteration victim_bf3 (.hip, hit_adv_amd/csrc)"
  for spec in "tests/test_gpu_kernels.py|$K_KERNELS" "tests/test_gpu_edges.py tests/test_z_r06_edges.py|$K_EDGES" "tests/test_gpu_attack.py|$K_ATTACK"; do
    files=${spec%%|*}; k=${spec#*|}
    echo "## $files"
    python -m pytest -p emu_plugin --emulate $files -q -rA -k "$k" 2>&1 | grep -E "^(PASSED|FAILED|ERROR|SKIPPED)|passed|failed" | sed 's/ - .*//'
  done
  echo "## V1 (csrc/victim_bf3.hip; its 16x16x32 matrix instructions emulated with a summation order of the emulator's own): the deferred-search variant == the shipped kernel, bit for bit, on emulator-sized flat shapes; two of the suite's own V1 tests"
  python -m pytest -p emu_plugin --emulate tests/test_z_r06_v1_defer.py tests/test_gpu_kernels.py -q -rA -k "deferred_search_gives or test_linear_max_fwd_bf16x3_ties_keep_the_first_point or test_linear_max_fwd_f16x2_raises_its_range_flag" 2>&1 | grep -E "^(PASSED|FAILED|ERROR|SKIPPED)|passed|failed" | sed 's/ - .*//'
  echo "# (also passed once, 1,040 s: tests/test_gpu_kernels.py::test_linear_max_fwd_bf16x3_same_bits_for_every_grid -- the split / merge / flat forms of the bf16x3 kernel, same bits for every grid)"
  echo "## tests/test_z_r06_concurrency.py (24 launches per kernel, no noise streams: the test's own logic only)"
  HITADV_CONCURRENCY_LAUNCHES=24 HITADV_CONCURRENCY_NOISE=0 python -m pytest -p emu_plugin --emulate tests/test_z_r06_concurrency.py -q -rA 2>&1 | grep -E "^(PASSED|FAILED|ERROR|SKIPPED)|passed|failed" | sed 's/ - .*//'
} > $OUT
tail -3 $OUT
