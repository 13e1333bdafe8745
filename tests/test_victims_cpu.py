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
