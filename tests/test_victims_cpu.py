"""PointNet++ and PCT on the CPU: the plain nn.Module formulation of hit_adv_amd/model/{pointnet2,pct}.py with the oracle's
restatement of the reference's sampling / grouping functions (oracle/victim_geometry.py) against fixtures g11 / g12,
captured from the unmodified reference.  This pins BOTH halves of what the GPU parity tests use as their CPU target:
the module definitions (same parameters, same function) and the oracle geometry (same FPS / ball-query / kNN tables)."""
import argparse

import numpy as np
import pytest
import torch

from helpers import T, golden
from oracle import victim_geometry as VG


@pytest.mark.parametrize("torch_ops", [False, True], ids=["c", "torch"])
def test_pointnet2_cpu_equals_reference_vectors(torch_ops):
    from hit_adv_amd.model import pointnet2 as P2
    fx = golden('g11_pointnet2.npz')
    torch.manual_seed(int(fx['init_seed']))
    m = VG.CpuVictim(P2.get_model(40, normal_channel=False).eval(), torch_ops)
    x = T(fx['x']).clone().requires_grad_()
    pts = x.detach().transpose(1, 2).contiguous()
    torch.manual_seed(int(fx['fwd_seed']))
    fps1 = (VG.farthest_point_sample if torch_ops else VG.c_farthest_point_sample)(pts, 512)
    assert torch.equal(fps1, T(fx['fps1']))
    ball = VG.query_ball_point if torch_ops else VG.c_query_ball_point
    assert torch.equal(ball(0.2, 32, pts, VG.index_points(pts, fps1)), T(fx['ball1']))
    torch.manual_seed(int(fx['fwd_seed']))
    logits, l3 = m(x)
    assert l3.shape == (2, 1024, 1)
    np.testing.assert_allclose(logits.detach(), fx['logits'], rtol=1e-5, atol=1e-6)
    (logits * T(fx['grad_w'])).sum().backward()
    np.testing.assert_allclose(x.grad, fx['grad_x'], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("torch_ops", [False, True], ids=["c", "torch"])
def test_pct_cpu_equals_reference_vectors(torch_ops):
    from hit_adv_amd.model import pct as PCT
    fx = golden('g12_pct.npz')
    torch.manual_seed(int(fx['init_seed']))
    m = VG.CpuVictim(PCT.Pct(argparse.Namespace(dropout=0.2), output_channels=40).eval(), torch_ops)
    x = T(fx['x']).clone().requires_grad_()
    torch.manual_seed(int(fx['fwd_seed']))
    assert torch.equal((VG.pct_fps if torch_ops else VG.c_pct_fps)(x.detach().transpose(1, 2).contiguous(), 512), T(fx['fps1']))
    torch.manual_seed(int(fx['fwd_seed']))
    logits = m(x)
    np.testing.assert_allclose(logits.detach(), fx['logits'], rtol=1e-5, atol=1e-6)
    (logits * T(fx['grad_w'])).sum().backward()
    np.testing.assert_allclose(x.grad, fx['grad_x'], rtol=1e-4, atol=1e-7)


def test_start_feed_serves_the_draws_a_live_victim_would_make():
    """model/_sampling.py: a feed drawn up front holds, forward by forward, exactly the ``randint`` draws the reference's
    victims make in every forward pass (pointnet2_utils.py:75, other_utils.py:264)."""
    from hit_adv_amd.model import _sampling as S
    from hit_adv_amd.model import pointnet2 as P2
    m = P2.get_model(40, normal_channel=False).eval()
    assert S.plan_of(m, 2048) == [2048, 512] and S.plan_of(torch.nn.Linear(2, 2), 1024) == []
    torch.manual_seed(9)
    feed = S.feed_for(m, 3, 2048, 4, 'cpu')
    torch.manual_seed(9)
    for f in range(4):
        for high in (2048, 512):
            want = torch.randint(0, high, (3,), dtype=torch.long)
            with S.using(feed):
                assert torch.equal(S.next_start(3, high, 'cpu'), want)
    assert int(feed.cursor) == 4
    feed.seek(1)
    with S.using(feed):
        a = S.next_start(3, 2048, 'cpu')
    assert torch.equal(a, feed.table[1, 0])
    lazy = S.StartFeed.empty([2048, 512], 3, 6, 'cpu')
    torch.manual_seed(9)
    lazy.load(2, 4)
    assert torch.equal(lazy.table[2:], feed.table) and int(lazy.cursor) == 2 and int(lazy.table[:2].abs().sum()) == 0
    try:
        with S.using(feed):
            S.next_start(4, 2048, 'cpu')
        raise AssertionError("a feed drawn for another batch size must refuse")
    except RuntimeError:
        pass


def test_degrade_on_fp16_range_runs_the_call_again_in_full_range_arithmetic():
    """The decorator the attacks' entry points carry (model/_pointwise.py): a call that ends in ``Fp16RangeExceeded`` is made again
    with the CPU generator rewound and every fp16x2 form off for its duration, one warning per process; a nested call (an attack
    that calls another decorated entry point) is not wrapped twice; other exceptions pass through."""
    import warnings

    import pytest
    import torch
    from hit_adv_amd.model import _pointwise as PW
    from hit_adv_amd.model.dgcnn import FoldedDGCNN
    from hit_adv_amd.model.pointnet import FoldedPointNet

    seen = []

    class Attack:
        @PW.degrade_on_fp16_range
        def attack(self, fail_first):
            draw = torch.rand(3)
            seen.append((FoldedPointNet.matrix_mode, PW.FUSED_GROUP_MAX, PW.FUSED_EMBEDDING_POOL, FoldedDGCNN.fused_embedding, draw))
            if fail_first and len(seen) == 1:
                raise PW.Fp16RangeExceeded("an operand beyond fp16's range")
            return self.inner()

        @PW.degrade_on_fp16_range
        def inner(self):
            return FoldedPointNet.matrix_mode

        @PW.degrade_on_fp16_range
        def broken(self):
            raise ValueError("not a range problem")

    before = (FoldedPointNet.matrix_mode, PW.FUSED_GROUP_MAX, PW.FUSED_EMBEDDING_POOL, FoldedDGCNN.fused_embedding)
    assert before == ('fp16x2', True, True, True)
    PW._DEGRADE_WARNED = False
    torch.manual_seed(3)
    with pytest.warns(RuntimeWarning, match="fp32's range"):
        out = Attack().attack(True)
    assert out == 'bf16x3' and len(seen) == 2
    assert seen[0][:4] == before and seen[1][:4] == ('bf16x3', False, False, False)
    assert torch.equal(seen[0][4], seen[1][4])  # the second run took the same draws
    assert (FoldedPointNet.matrix_mode, PW.FUSED_GROUP_MAX, PW.FUSED_EMBEDDING_POOL, FoldedDGCNN.fused_embedding) == before
    del seen[:]
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # a second occurrence degrades silently
        assert Attack().attack(True) == 'bf16x3'
        del seen[:]
        assert Attack().attack(False) == 'fp16x2' and len(seen) == 1
    with pytest.raises(ValueError):
        Attack().broken()
    assert PW._FULL_RANGE_DEPTH == 0
