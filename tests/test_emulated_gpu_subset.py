"""A slice of the `-m gpu` parity suite run UNMODIFIED on the CPU wave emulator, inside the CPU suite.

tests/native/emu_plugin.py builds the non-matrix kernel files of hit_adv_amd/csrc/ for the emulator (tests/native/emu_build.py), puts the
result where hit_adv_amd._lib keeps libhitadv_hip.so and maps the tests' `.cuda()` / `device='cuda'` onto the CPU -- the test bodies,
`hit_adv_amd/ops.py`, the attack classes and the kernels' source are the product's.  Here: the HiT-ADV attack itself on the toy victim
of fixture g5 (every kernel of the loop: deformation forward / backward with the Adam tail, regulariser, iteration head, best tracking,
bisection) following the REFERENCE's trajectory; ten binary steps of bookkeeping (g5c); CWKNN's trajectory (g7); the deformation against
g3; best tracking + Adam; the fused regulariser and adversarial losses; tie rules of nn_min / kNN; the extension sampler's known answers.
The whole emulable part of the suite (~230 tests, ~12 min): tools/run_emulated_suite.sh -> tests/golden/emulated_suite_report.txt.
It runs in a subprocess: the plugin patches torch.cuda for the process it lives in."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANGXX = "/opt/rocm/lib/llvm/bin/clang++"

SLICE = [
    "tests/test_gpu_attack.py::test_hit_adv_follows_reference_trajectory[False-True]",
    "tests/test_gpu_attack.py::test_hit_adv_bookkeeping_over_ten_binary_steps[False]",
    "tests/test_gpu_attack.py::test_cwknn_follows_reference_trajectory",
    "tests/test_gpu_kernels.py::test_deform_forward_backward_vs_reference_vectors",
    "tests/test_gpu_kernels.py::test_best_update_and_adam_match_host_logic",
    "tests/test_gpu_kernels.py::test_fused_regulariser_matches_torch_composition",
    "tests/test_gpu_kernels.py::test_fused_adv_losses_match_reference_modules",
    "tests/test_gpu_kernels.py::test_nn_min_ties_take_lowest_index",
    "tests/test_gpu_kernels.py::test_knn_points_heavy_ties_and_log_compaction",
    "tests/test_gpu_kernels.py::test_fps_ext_known_answers",
    "tests/test_z_r06_edges.py",
]


@pytest.mark.skipif(not os.path.exists(CLANGXX), reason="no host clang++")
def test_a_slice_of_the_gpu_parity_suite_passes_unmodified_on_the_cpu_wave_emulator():
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "tests", "native") + os.pathsep + os.environ.get("PYTHONPATH", ""),
               HITADV_EMU_STEMS="pairwise,knn,sampling,grouping,deform,regulariser,attack_state,iteration")
    r = subprocess.run([sys.executable, "-m", "pytest", "-p", "emu_plugin", "--emulate", "-q", "-p", "no:cacheprovider"] + SLICE,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 14 and "failed" not in tail and "error" not in tail, tail
