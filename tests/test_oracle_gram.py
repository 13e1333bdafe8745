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
