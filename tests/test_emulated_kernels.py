"""The index kernels' OWN SOURCE, executed on the CPU, against the oracle -- bit for bit, through the C ABI.

tests/native/emu_build.py takes hit_adv_amd/csrc/{pairwise,sampling,grouping,knn}.hip as they are, rewrites only what plain C++ cannot
parse (the `<<<...>>>` launch syntax, `extern __shared__` arrays, fps_lean's one LDS atomic written as inline asm) and compiles them for
x86-64 against a wave64 SIMT emulator (tests/native/emu/simt_emu.hpp: the threads of a block are fibres; __syncthreads and every
wave-level operation -- __shfl*, __ballot, DPP, readlane, readfirstlane -- are rendezvous of the live lanes, computed as the hardware
defines them).  The kernels' bodies, their launchers, their size dispatch and their extern "C" entry points are the product's text; the
"device" pointers are host pointers.  What runs here is therefore the kernels' LOGIC -- index arithmetic, tie rules, chunking, LDS
protocols, DPP reductions, barrier placement -- and it must give the C oracle's tables EXACTLY:

  K1 / K2  pairwise matrix, fused Chamfer / Hausdorff minima + arg-minima (util/set_distance.py:15-70), direct and Gram form
  K4       knn_points (HiT_ADV.py:78-80,320-336; dist_utils.py:136-175), K = 1 .. 32, four distance forms, one and several chunks
  K5       fps_from_start (HiT_ADV.py:489-510) by BOTH kernels (the 64-bit-key fps<> and fps_lean), PCT's sampler
           (other_utils.py:254-272), the CUDA extension's sampler with its thread-slot tie rule (sampling_gpu.cu:69-173)
  natives  ball_query / group_points / three_nn / three_interpolate (+ grads) of pointnet2_ops, the victims' query_ball_point

What this cannot show is the hardware: that gfx950's add / mul / fma round as IEEE says, that LDS and barriers order memory as the
kernels assume, how fast anything is.  The -m gpu tests are for that, and in round 6 -- which had no GPU -- this file is the strongest
statement available about the code that would run there.  (Matrix-instruction kernels are not emulated: MFMA's internal summation order
is the hardware's.)"""
import ctypes
import os
import sys

import pytest
import torch

from oracle import c_oracle as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "native"))
import emu_build  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(emu_build.CLANGXX), reason="no host clang++")


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


@pytest.fixture(scope="module")
def E(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("emu"))
    libs = {}

    def get(stem):
        if stem not in libs:
            libs[stem] = ctypes.CDLL(emu_build.build(stem, d))
        return libs[stem]
    return get


def cloud(b, n, seed, kind="gauss"):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, n, 3, generator=g)
    if kind == "sphere":
        x = x / x.norm(dim=2, keepdim=True) + 0.01 * torch.randn(b, n, 3, generator=g)
    elif kind == "lattice":  # exact ties everywhere
        x = torch.randint(-3, 4, (b, n, 3), generator=g).float() * 0.25
    else:
        x = x - x.mean(1, keepdim=True)
        x = x / x.norm(dim=2).max()
    return x.contiguous()


# ------------------------------------------------------------------------------------------------------------------ K1 / K2
@pytest.mark.parametrize("B,n,m", [(2, 300, 257), (1, 1024, 1024), (2, 5, 1030), (1, 2500, 2300), (3, 1, 1)])
@pytest.mark.parametrize("form", [0, 1])
def test_pairwise_matrix_and_fused_minima_from_the_kernel_source(E, B, n, m, form):
    lib = E("pairwise")
    for kind in ("gauss", "lattice"):
        x, y = cloud(B, n, 1, kind), cloud(B, m, 2, kind)
        mx, ax = torch.empty(B, n), torch.empty(B, n, dtype=torch.int32)
        my, ay = torch.empty(B, m), torch.empty(B, m, dtype=torch.int32)
        assert lib.hitadv_nn_min(P(x), P(y), B, n, m, 3, form, P(mx), P(ax), P(my), P(ay), None, None) == 0
        rx, rax = N.nn_min(x, y, form)
        ry, ray = N.nn_min(y, x, form)
        assert torch.equal(mx, rx) and torch.equal(ax, rax) and torch.equal(my, ry) and torch.equal(ay, ray), kind
        if n * m <= 1100 * 1100:
            M = torch.empty(B, n, m)
            assert lib.hitadv_pairwise_sqdist(P(x), P(y), P(M), B, n, m, 3, form, None) == 0
            assert torch.equal(M.view(torch.int32), N.pairwise(x, y, form).view(torch.int32)), kind


def test_generic_dimension_pairwise_and_its_minima(E):
    """Quirk Q1: HiT_ADV.py:230 feeds [B,3,N] tensors to ChamferDist -- a 3 x 3 matrix of distances between 1024-dimensional rows."""
    lib = E("pairwise")
    g = torch.Generator().manual_seed(3)
    x, y = torch.randn(2, 3, 1024, generator=g).contiguous(), torch.randn(2, 3, 1024, generator=g).contiguous()
    M = torch.empty(2, 3, 3)
    assert lib.hitadv_pairwise_sqdist(P(x), P(y), P(M), 2, 3, 3, 1024, 0, None) == 0
    ref = ((x[:, :, None, :].double() - y[:, None, :, :].double()) ** 2).sum(-1)
    assert float((M.double() - ref).abs().max() / ref.max()) < 1e-6  # lanes stride the feature axis: another summation order than torch's
    mx, ax, my, ay = torch.empty(2, 3), torch.empty(2, 3, dtype=torch.int32), torch.empty(2, 3), torch.empty(2, 3, dtype=torch.int32)
    scratch = torch.empty(2 * 3 * 3)
    assert lib.hitadv_nn_min(P(x), P(y), 2, 3, 3, 1024, 0, P(mx), P(ax), P(my), P(ay), P(scratch), None) == 0
    assert torch.equal(mx, M.min(2).values) and torch.equal(ax.long(), M.argmin(2)) and torch.equal(my, M.min(1).values)


def test_fused_minima_backward_is_the_gather_through_the_saved_arguments(E):
    lib = E("pairwise")
    B, n, m = 2, 300, 513
    x, y = cloud(B, n, 4), cloud(B, m, 5)
    g = torch.Generator().manual_seed(6)
    gx, gy = torch.randn(B, n, generator=g), torch.randn(B, m, generator=g)
    _, ax = N.nn_min(x, y)
    _, ay = N.nn_min(y, x)
    dx, dy = torch.empty(B, n, 3), torch.empty(B, m, 3)
    assert lib.hitadv_nn_min_bwd(P(x), P(y), P(ax), P(ay), P(gx), P(gy), B, n, m, 3, P(dx), P(dy), None) == 0
    xr, yr = x.clone().requires_grad_(), y.clone().requires_grad_()
    D = ((xr[:, :, None] - yr[:, None]) ** 2).sum(-1)
    (D.gather(2, ax.long()[..., None])[..., 0] * gx).sum().backward(retain_graph=True)
    (D.gather(1, ay.long()[:, None])[:, 0] * gy).sum().backward()
    assert float((dx - xr.grad).abs().max()) < 1e-5 and float((dy - yr.grad).abs().max()) < 1e-5


# ------------------------------------------------------------------------------------------------------------------ K4
@pytest.mark.parametrize("n,m,K", [(300, 300, 1), (256, 1024, 17), (1024, 1024, 6), (70, 3000, 17), (40, 64, 32), (9, 5, 5)])
def test_knn_points_from_the_kernel_source(E, n, m, K):
    lib = E("knn")
    for kind in ("gauss", "lattice", "sphere"):
        p = cloud(2, m, 7, kind)
        q = p[:, :n].contiguous() if n <= m else cloud(2, n, 8, kind)  # self-cloud queries: the zero distance and its ties
        for form in (0, 2, 3):
            d, ix = torch.empty(2, n, K), torch.empty(2, n, K, dtype=torch.int64)
            assert lib.hitadv_knn_points(P(q), P(p), 2, n, m, K, form, P(d), P(ix), 1, None) == 0
            rd, rix = N.knn_points(q, p, K, form)
            assert torch.equal(ix, rix) and torch.equal(d.view(torch.int32), rd.view(torch.int32)), (kind, form)
    d32 = torch.empty(2, n, K, dtype=torch.int32)
    assert lib.hitadv_knn_points(P(q), P(p), 2, n, m, K, 0, P(d), P(d32), 0, None) == 0  # the int32 table of the same call
    assert torch.equal(d32.long(), N.knn_points(q, p, K, 0)[1])


# ------------------------------------------------------------------------------------------------------------------ K5
@pytest.mark.parametrize("n,m", [(1024, 96), (2048, 96), (300, 300), (64, 5), (513, 70), (4500, 24)])
@pytest.mark.parametrize("form", [0, 1])
def test_fps_from_start_and_pct_sampler_by_both_kernels(E, n, m, form):
    """form 0 = fps<> (the default), 1 = fps_lean (HITADV_FPS_FORM=1; its ds_max_rtn_u64 exchange as the same operation in C++): the
    bit-pattern running distances, the DPP maximum, the ballot search for the holder, the three-word rotation -- everything but the asm."""
    lib = E("sampling")
    shipped = lib.hitadv_debug_fps_form(-1)
    try:
        lib.hitadv_debug_fps_form(form)
        for kind in ("gauss", "sphere", "lattice"):
            x = cloud(3, n, 9, kind)
            start = torch.tensor([0, n - 1, n // 2], dtype=torch.int64)
            out = torch.empty(3, m, dtype=torch.int64)
            assert lib.hitadv_fps_from_start(P(x), P(start), 3, n, m, P(out), None) == 0
            assert torch.equal(out, N.fps_from_start(x, m, start)), (kind, "direct")
            assert lib.hitadv_fps_pct(P(x), P(start), 3, n, m, P(out), None) == 0
            assert torch.equal(out, N.fps_pct(x, m, start)), (kind, "pct")
    finally:
        lib.hitadv_debug_fps_form(shipped)


@pytest.mark.parametrize("n,m", [(1024, 51), (1000, 128), (4096, 64), (37, 37), (600, 1)])
def test_extension_sampler_with_its_thread_slot_tie_rule_and_gather(E, n, m):
    lib = E("sampling")
    for kind in ("gauss", "lattice"):
        x = cloud(2, n, 11, kind)
        x[0, 3] = 0.0  # |p|^2 <= 1e-3: skipped by the extension (sampling_gpu.cu:100-101)
        out = torch.empty(2, m, dtype=torch.int32)
        assert lib.hitadv_furthest_point_sampling(2, n, m, P(x), None, P(out), None) == 0
        want = N.furthest_point_sampling(x, m)
        assert torch.equal(out, want), kind
        feats = x.transpose(1, 2).contiguous()
        got = torch.empty(2, 3, m)
        assert lib.hitadv_gather_points(2, 3, n, m, P(feats), P(want), P(got), None) == 0
        assert torch.equal(got, N.gather_points(feats, want))
        go = torch.randn(2, 3, m, generator=torch.Generator().manual_seed(1))
        gp = torch.empty(2, 3, n)
        assert lib.hitadv_gather_points_grad(2, 3, n, m, P(go), P(want), P(gp), None) == 0
        assert torch.equal(gp, N.gather_points_grad(go, want, n))


# ------------------------------------------------------------------------------------------------------------------ natives
def test_ball_query_grouping_and_interpolation_from_the_kernel_source(E):
    lib = E("grouping")
    x = cloud(2, 1024, 13)
    fidx = N.furthest_point_sampling(x, 51)
    new_xyz = torch.gather(x, 1, fidx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    feats = torch.randn(2, 7, 1024, generator=torch.Generator().manual_seed(2)).contiguous()
    for r, ns in ((0.22, 49), (0.05, 8), (0.6, 64), (1e-4, 4)):
        idx = torch.empty(2, 51, ns, dtype=torch.int32)
        assert lib.hitadv_query_ball_point(2, 1024, 51, ctypes.c_float(r), ns, P(new_xyz), P(x), P(idx), None) == 0
        want = N.ball_query(new_xyz, x, r, ns)
        assert torch.equal(idx, want), (r, ns)
        grouped = torch.empty(2, 7, 51, ns)
        assert lib.hitadv_group_points(2, 7, 1024, 51, ns, P(feats), P(want), P(grouped), None) == 0
        assert torch.equal(grouped, N.group_points(feats, want))
        go = torch.randn(2, 7, 51, ns, generator=torch.Generator().manual_seed(3))
        gp = torch.empty(2, 7, 1024)
        assert lib.hitadv_group_points_grad(2, 7, 1024, 51, ns, P(go), P(want), P(gp), None) == 0
        assert torch.equal(gp, N.group_points_grad(go, want, 1024))
    unknown, known = cloud(2, 300, 14), cloud(2, 77, 15)
    d2, ti = torch.empty(2, 300, 3), torch.empty(2, 300, 3, dtype=torch.int32)
    assert lib.hitadv_three_nn(2, 300, 77, P(unknown), P(known), P(d2), P(ti), None) == 0
    rd2, rti = N.three_nn(unknown, known)
    assert torch.equal(ti, rti) and torch.equal(d2.view(torch.int32), rd2.view(torch.int32))
    pts = torch.randn(2, 5, 77, generator=torch.Generator().manual_seed(4)).contiguous()
    w = torch.rand(2, 300, 3, generator=torch.Generator().manual_seed(5)).contiguous()
    out = torch.empty(2, 5, 300)
    assert lib.hitadv_three_interpolate(2, 5, 77, 300, P(pts), P(rti), P(w), P(out), None) == 0
    assert torch.equal(out.view(torch.int32), N.three_interpolate(pts, rti, w).view(torch.int32))
    go = torch.randn(2, 5, 300, generator=torch.Generator().manual_seed(6))
    gp = torch.empty(2, 5, 77)
    assert lib.hitadv_three_interpolate_grad(2, 5, 300, 77, P(go), P(rti), P(w), P(gp), None) == 0
    assert torch.equal(gp.view(torch.int32), N.three_interpolate_grad(go, rti, w, 77).view(torch.int32))


@pytest.mark.parametrize("n,s,radius,nsample", [(1024, 512, 0.2, 32), (512, 128, 0.4, 64), (2048, 512, 0.2, 32), (300, 40, 0.05, 16)])
def test_victims_ball_query_from_the_kernel_source(E, n, s, radius, nsample):
    """model/pointnet2_utils.py:87-107: Gram-form square_distance (form 3) and the direct form, `>` radius^2, ascending, padded."""
    lib = E("grouping")
    for kind in ("gauss", "sphere"):
        x = cloud(2, n, 17, kind)
        q = x[:, ::n // s][:, :s].contiguous()
        for form in (3, 0):
            idx = torch.empty(2, s, nsample, dtype=torch.int64)
            assert lib.hitadv_query_ball_point_victim(2, n, s, ctypes.c_float(N.radius_squared(radius)), nsample, form, P(q), P(x), P(idx), None) == 0
            assert torch.equal(idx, N.query_ball_point(radius, nsample, x, q, form)), (kind, form)
