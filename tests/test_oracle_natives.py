"""Known-answer tests for the C restatement of the pointnet2_ops kernels.

The reference ships no vectors for its CUDA extension, so every expectation
below is derived by hand from the kernel sources cited in
oracle/pointnet2_oracle.c (parity with the original binary stays unpinned).
Also cross-checks the C and torch statements of the canonical kNN / FPS rules.
"""
import numpy as np
import torch

from helpers import T, golden, synth_batch
from oracle import c_oracle as N
from oracle import hitadv_oracle as O


def test_fps_ext_collinear_points():
    # points on a line at x = 1..8 (all |p|^2 > 1e-3): start 0, then the far end,
    # then the point maximising min-distance to {1, 8}: x=4 (slot 3) and x=5 (slot 4) tie at
    # d^2 = 9.  With 8 "threads" of one point each the tree folds slot 4 into slot 0 and slot 3
    # into slot 1 (via 3->1 at stride 2); the last level keeps slot 0 on the tie -> index 4.
    xyz = torch.zeros(1, 8, 3)
    xyz[0, :, 0] = torch.arange(1, 9).float()
    idx = N.furthest_point_sampling(xyz, 4)
    assert idx.dtype == torch.int32
    assert idx[0].tolist()[:3] == [0, 7, 4]
    # next: min-dist^2 to {1, 8, 5}: x=2:1, x=3:4, x=4:1, x=6:1, x=7:1 -> x=3 (index 2)
    assert idx[0, 3].item() == 2


def test_fps_ext_skips_near_origin_points():
    # sampling_gpu.cu:100-101: points with |p|^2 <= 1e-3 never update temp nor compete
    xyz = torch.tensor([[[1., 0, 0], [0.01, 0, 0], [0, 0.02, 0], [-1., 0, 0], [0, 1., 0]]])
    idx = N.furthest_point_sampling(xyz, 3)
    assert idx[0].tolist() == [0, 3, 4]
    # all points skipped -> best stays -1 / besti 0 everywhere -> index 0 repeated
    z = torch.zeros(1, 4, 3)
    assert N.furthest_point_sampling(z, 3)[0].tolist() == [0, 0, 0]


def test_fps_ext_tie_rule_is_slot_order_not_index_order():
    # 1024 points, 512 slots: slot t owns k=t and k=t+512.  Put the two equal maxima at
    # k=1 (slot 1) and k=514 (slot 2): the last tree level compares slot 0 (even slots)
    # against slot 1 (odd slots) and keeps slot 0's winner on ties -> k=514 wins.
    xyz = torch.zeros(1, 1024, 3)
    xyz[0, :, 0] = 0.1  # |p|^2 = 0.01 > 1e-3, all coincide with the start point
    xyz[0, 1] = torch.tensor([0.1, 0.5, 0.0])
    xyz[0, 514] = torch.tensor([0.1, -0.5, 0.0])
    idx = N.furthest_point_sampling(xyz, 2)
    assert idx[0].tolist() == [0, 514]


def test_ball_query_rules():
    xyz = torch.tensor([[[0., 0, 0], [0.5, 0, 0], [1.0, 0, 0], [0.2, 0, 0], [5, 5, 5]]])
    q = torch.tensor([[[0., 0, 0], [9., 9, 9], [1.0, 0, 0]]])
    idx = N.ball_query(q, xyz, 0.5, 3)
    # strict '<': 0.5 away is outside; first hit pre-fills the row
    assert idx[0, 0].tolist() == [0, 3, 0]
    # empty ball keeps the zero initialisation
    assert idx[0, 1].tolist() == [0, 0, 0]
    assert idx[0, 2].tolist() == [2, 2, 2]
    idx = N.ball_query(q, xyz, 0.6, 2)
    assert idx[0, 0].tolist() == [0, 1]  # stops after nsample hits in ascending index order
    assert idx[0, 2].tolist() == [1, 2]


def test_three_nn_strict_less_tie_order():
    known = torch.tensor([[[1., 0, 0], [-1., 0, 0], [0, 1., 0], [0, 0, 2.]]])
    unknown = torch.tensor([[[0., 0, 0]]])
    d, ix = N.three_nn(unknown, known)
    assert ix[0, 0].tolist() == [0, 1, 2]  # three exact ties -> earlier index first
    assert d[0, 0].tolist() == [1.0, 1.0, 1.0]


def test_group_gather_interpolate_and_grads():
    g = torch.Generator().manual_seed(0)
    pts = torch.randn(2, 4, 10, generator=g)
    idx = torch.randint(0, 10, (2, 3, 5), generator=g).int()
    out = N.group_points(pts, idx)
    exp = torch.stack([pts[b][:, idx[b].long()] for b in range(2)])
    assert torch.equal(out, exp)
    go = torch.randn(2, 4, 3, 5, generator=g)
    gp = N.group_points_grad(go, idx, 10)
    ref = torch.zeros(2, 4, 10)
    for b in range(2):
        ref[b].index_add_(1, idx[b].reshape(-1).long(), go[b].reshape(4, -1))
    np.testing.assert_allclose(gp, ref, rtol=1e-6, atol=1e-6)

    gi = torch.randint(0, 10, (2, 6), generator=g).int()
    assert torch.equal(N.gather_points(pts, gi), torch.stack([pts[b][:, gi[b].long()] for b in range(2)]))
    gg = torch.randn(2, 4, 6, generator=g)
    ref = torch.zeros(2, 4, 10)
    for b in range(2):
        ref[b].index_add_(1, gi[b].long(), gg[b])
    np.testing.assert_allclose(N.gather_points_grad(gg, gi, 10), ref, rtol=1e-6, atol=1e-6)

    ti = torch.randint(0, 10, (2, 7, 3), generator=g).int()
    w = torch.rand(2, 7, 3, generator=g)
    out = N.three_interpolate(pts, ti, w)
    exp = torch.stack([(pts[b][:, ti[b].long()] * w[b][None]).sum(-1) for b in range(2)])
    np.testing.assert_allclose(out, exp, rtol=1e-6, atol=1e-6)
    g3 = torch.randn(2, 4, 7, generator=g)
    ref = torch.zeros(2, 4, 10)
    for b in range(2):
        for t in range(3):
            ref[b].index_add_(1, ti[b, :, t].long(), g3[b] * w[b, :, t][None])
    np.testing.assert_allclose(N.three_interpolate_grad(g3, ti, w, 10), ref, rtol=1e-5, atol=1e-6)


def test_c_and_torch_canonical_knn_agree_bit_exact():
    data, _ = synth_batch(2, 512, first=50)
    xyz = data[:, :, :3].contiguous()
    q = xyz[:, :100].contiguous()
    for K in (1, 5, 17):
        d_t, i_t = O.knn_points(q, xyz, K)
        d_c, i_c = N.knn_points(q, xyz, K)
        assert torch.equal(i_t, i_c)
        assert torch.equal(d_t, d_c)
    # duplicated points: ties resolve to the lower index in both statements
    dup = torch.cat([xyz[:, :8], xyz[:, :8]], 1)
    d_t, i_t = O.knn_points(dup, dup, 4)
    d_c, i_c = N.knn_points(dup, dup, 4)
    assert torch.equal(i_t, i_c) and (i_c[:, 8:, 0] == torch.arange(8)).all()


def test_c_fps_from_start_matches_reference_vector():
    fx = golden('g4_fps.npz')
    idx = N.fps_from_start(T(fx['xyz']), 256, T(fx['start']))
    assert (idx.numpy() == fx['idx']).all()


def test_c_nn_min_matches_direct_matrix():
    data, _ = synth_batch(2, 256, first=60)
    x = data[:, :, :3].contiguous()
    y = data[:, :200, 3:].contiguous()
    d, ix = N.nn_min(x, y)
    P = O.pairwise_sqdist_direct(x, y)
    assert torch.equal(d, P.min(2).values)
    assert torch.equal(ix.long(), P.argmin(2))


def test_uniform_loss_runs_on_natives():
    data, _ = synth_batch(2, 1024, first=70)
    v = O.uniform_loss(data[:, :, :3].contiguous(), N, k=5)
    assert v.ndim == 0 and torch.isfinite(v) and v.item() > 0
