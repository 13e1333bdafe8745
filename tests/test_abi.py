"""The C-ABI library loads without a GPU and exports exactly what include/hitadv.h declares
(no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'hitadv.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(hitadv_\w+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from hit_adv_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 21
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.PROTOTYPES) == names  # the ctypes table and the header stay in sync
    assert lib.hitadv_version().decode().startswith('hitadv-hip')
    assert lib.hitadv_deform_bwd_scratch_floats(32, 1024, 192) == 32 * 16 * 4 * 192


def test_v1_launch_geometry_of_the_bench_shapes():
    """Host logic, no GPU: which form of the 128 -> 1024 layer + max a shape selects is a function of (B, N, Cout, blocks) that the
    scratch query exposes (S = point splits per cloud; csrc/victim_bf3.hip::bf3_split).  The stacked loop of the bench (stacks of 6-8
    attacks = 192-256 clouds on 128 workgroups, `HITADV_V1_BLOCKS_IN_FLIGHT`) must select S = 1 -- the FLAT kernel, the one the roofline
    notes, the ISA guards and HITADV_V1_DEFER are about; one attack alone (32 clouds on 256 workgroups) takes two splits per cloud."""
    from hit_adv_amd import _lib
    f = _lib.load().hitadv_linear_max_fwd_bf16x3_scratch
    splits = lambda B, N, blocks: f(B, N, 1024, blocks) // (B * 1024)  # noqa: E731
    for B in (192, 224, 256):
        assert splits(B, 1024, 128) == 1 and splits(B, 1024, 0) == 1
    assert splits(32, 1024, 0) == 2 and splits(32, 1024, 128) == 1
    assert splits(64, 2048, 0) == 1 and splits(13, 1280, 8) == 1
    assert splits(1, 1024, 0) == 16 and splits(1, 64, 0) == 1          # never more splits than 64-point tiles
    assert f(32, 1024, 1024, 7) == 0 and f(32, 1024, 1024, 300) == 0     # blocks outside 8..256: refused


def test_invalid_arguments_return_error_codes_not_crashes():
    from hit_adv_amd import _lib
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    assert lib.hitadv_pairwise_sqdist(null, null, null, 1, 1, 1, 3, 0, null) == -1
    assert lib.hitadv_knn_points(null, null, 1, 1, 1, 1, 0, null, null, 1, null) == -1
    assert lib.hitadv_deform_fwd(null, null, null, null, 1, 1, 1, null, null, null) == -1
    assert lib.hitadv_furthest_point_sampling(0, 0, 0, null, null, null, null) == -1


def test_product_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from hit_adv_amd import ops
    x = torch.zeros(1, 4, 3)
    with pytest.raises(RuntimeError):
        ops.nn_min(x, x)
    with pytest.raises(RuntimeError):
        ops.pairwise_sqdist(x, x)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'hit_adv_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert 'oracle' not in src.replace('CPU oracle', '').replace('the oracle', ''), os.path.join(dp, f)
