"""The oracle's restatement of the reference's Gram-form fp32 arithmetic (oracle/pointnet2_oracle.c::pair_value, forms 1
and 2) against torch itself on this box's CPU: ``_Distance.batch_pairwise_dist`` (util/set_distance.py:15-32) and the
distance matrix of ``KNNDist`` (util/dist_utils.py:148-150) come out BIT FOR BIT, so the HIP kernels' Gram modes, which
are tested bit for bit against this C code on the GPU box, reproduce the reference's values and not merely its formula.
(Matrices smaller than torch's GEMM threshold take another code path in torch and are compared to rounding.)"""
import numpy as np
import pytest
import torch

from helpers import T, golden, synth_batch
from oracle import c_oracle as N
from oracle import hitadv_oracle as O
from oracle import victim_geometry as VG


def _clouds(b, n, first):
    return synth_batch(b, n, first=first)[0][:, :, :3].contiguous()


@pytest.mark.parametrize("n,m", [(1024, 1024), (256, 1024), (100, 1001), (64, 48)])
def test_set_distance_gram_matrix_is_reproduced_bit_for_bit(n, m):
    x, y = _clouds(2, n, 300), _clouds(2, m, 310)
    y = y + 0.01 * torch.randn(y.shape, generator=torch.Generator().manual_seed(1))
    ref = O.pairwise_sqdist_gram(x, y)  # op for op the reference's batch_pairwise_dist
    got = N.pairwise(x, y, N.FORM_GRAM)
    assert torch.equal(got, ref)
    # and so are the reductions the reference builds on it (set_distance.py:45-49, :62-68)
    mins, arg = N.nn_min(x, y, N.FORM_GRAM)
    tm = torch.min(ref, 2)
    assert torch.equal(mins, tm.values)
    assert torch.equal(ref.gather(2, arg.long().unsqueeze(-1)).squeeze(-1), tm.values)


@pytest.mark.parametrize("n", [1024, 333])
def test_knn_dist_gram_matrix_is_reproduced_bit_for_bit(n):
    pc = _clouds(2, n, 320).transpose(1, 2).contiguous()  # [B,3,K] as KNNDist holds it (dist_utils.py:145-147)
    inner = -2. * torch.matmul(pc.transpose(2, 1), pc)
    xx = torch.sum(pc ** 2, dim=1, keepdim=True)
    dist = xx + inner + xx.transpose(2, 1)
    pts = pc.transpose(1, 2).contiguous()
    assert torch.equal(N.pairwise(pts, pts, N.FORM_GRAM_KNN), dist)
    for k in (4, 5):
        neg, _ = (-dist).topk(k=k + 1, dim=-1)
        d, idx = N.knn_points(pts, pts, k + 1, N.FORM_GRAM_KNN)
        assert torch.equal(d, -neg)  # same k+1 smallest values, ascending (dist_utils.py:156-158)
        assert torch.equal(dist.gather(2, idx), d)


def test_gram_forms_reproduce_the_reference_vectors():
    """g1 (chamfer / hausdorff captured from util/set_distance.py) and g2 (KNNDist from util/dist_utils.py): the minima
    are bit-exact, so what is left is the order of the final mean over N values."""
    fx = golden('g1_set_distance.npz')
    adv, ori = T(fx['adv']), T(fx['ori'])
    to_gt, _ = N.nn_min(adv, ori, N.FORM_GRAM)    # for every pred its nearest gt (P = dist(gts, preds), min over dim 1)
    to_pred, _ = N.nn_min(ori, adv, N.FORM_GRAM)
    np.testing.assert_allclose(to_gt.mean(1), fx['chamfer_l1'], rtol=2e-7)
    np.testing.assert_allclose(to_pred.mean(1), fx['chamfer_l2'], rtol=2e-7)
    assert np.array_equal(to_gt.max(1).values.numpy(), fx['hausdorff_l1'])
    assert np.array_equal(to_pred.max(1).values.numpy(), fx['hausdorff_l2'])


# (N, S) of every sampling / grouping call inside the victims: PointNet++ SSG at 1024 and 2048 points (cfg4), its second
# set-abstraction level, PCT's two Local_op levels, and two odd sizes
VICTIM_SHAPES = [(1024, 512), (2048, 512), (512, 128), (512, 256), (300, 77), (64, 48)]


@pytest.mark.parametrize("n,s", VICTIM_SHAPES)
def test_victim_square_distance_and_its_consumers_bit_for_bit(n, s):
    """form 3 = the victims' ``square_distance`` (model/pointnet2_utils.py:19-41) as torch evaluates it, and on it the
    ball query (:87-107) and PCT's kNN grouping (model/pct_utils.py:98-109)."""
    xyz = _clouds(3, n, 400)
    new_xyz = xyz[:, :s].contiguous() + 0.001
    assert torch.equal(N.pairwise(new_xyz, xyz, N.FORM_SQUARE_DISTANCE), VG.square_distance(new_xyz, xyz))
    for radius, nsample in ((0.2, 32), (0.4, 64)):
        if nsample <= n:
            assert torch.equal(VG.c_query_ball_point(radius, nsample, xyz, new_xyz), VG.query_ball_point(radius, nsample, xyz, new_xyz))
    k = 32
    assert torch.equal(VG.c_pct_knn_point(k, xyz, new_xyz).sort(-1)[0], VG.pct_knn_point(k, xyz, new_xyz).sort(-1)[0])


@pytest.mark.parametrize("n,s", [v for v in VICTIM_SHAPES if v[0] >= 200])
def test_pct_sampler_bit_for_bit(n, s):
    """form 4 = ``get_dists`` of ONE point against the cloud (util/other_utils.py:237-251; a one-row matrix product takes a
    different code path in torch's BLAS than the GEMM of forms 1-3) and PCT's sampler on it (:254-272).  Clouds of fewer than
    134 points would take a third path (torch multiplies matrices with fewer than 400 multiply-adds itself, without
    BLAS); the reference's PCT samples from 1024 and 512 points (model/pct_cls.py:48-53)."""
    xyz = _clouds(3, n, 410)
    cur = xyz[:, 5:6].contiguous()
    got = N.pairwise(cur, xyz, N.FORM_PCT_DISTS)
    squared = VG.get_dists_squared(cur, xyz)  # everything in front of the reference's torch.sqrt: bit for bit
    assert torch.equal(got, torch.from_numpy(np.sqrt(squared.numpy())))  # numpy's fp32 sqrt is the correctly rounded one
    # torch.sqrt itself is MKL VML's vsSqrt (< 1 ulp, not correctly rounded): at most one ulp off, in a few values per thousand
    ref = VG.get_dists(cur, xyz)
    off = got != ref
    assert off.float().mean().item() < 0.02
    assert torch.equal(torch.where(off, torch.nextafter(ref, got), ref), got)
    torch.manual_seed(n + s)
    want = VG.pct_fps(xyz, s)
    torch.manual_seed(n + s)
    assert torch.equal(VG.c_pct_fps(xyz, s), want)


def test_ball_query_threshold_is_the_fp32_value_of_the_double_square():
    """``sqrdists > radius ** 2``: the Python double is rounded to fp32 for the comparison; 0.2f * 0.2f is one ulp above."""
    r2 = N.radius_squared(0.2)
    assert r2 == float(np.float32(0.2 ** 2)) and r2 != float(np.float32(0.2) * np.float32(0.2))
    d = torch.tensor([np.float32(0.2) * np.float32(0.2)], dtype=torch.float32)
    assert bool((d > 0.2 ** 2).item())  # torch agrees: the fp32 product lies outside the ball


def test_dgcnn_first_layer_score_is_the_negated_knn_form():
    """model/dgcnn_cls.py:7-13 on 3-D coordinates: score = (-|x_j|^2 - (-2 x_i.x_j)) - |x_i|^2 is, operation for operation, the
    negation of form 2 (negation and a - b = a + (-b) are exact), so its k largest entries are the k smallest form-2 values."""
    x = _clouds(2, 1024, 330).transpose(1, 2).contiguous()  # [B,3,N]
    inner = -2 * torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x ** 2, dim=1, keepdim=True)
    score = -xx - inner - xx.transpose(2, 1)
    pts = x.transpose(1, 2).contiguous()
    assert torch.equal(-N.pairwise(pts, pts, N.FORM_GRAM_KNN), score)
    d, idx = N.knn_points(pts, pts, 5, N.FORM_GRAM_KNN)
    assert torch.equal(-d, score.topk(k=5, dim=-1).values)
