"""EMULATOR-ONLY (tools/run_emulated_suite.sh; ~5 min): the part of tests/test_gpu_attack.py::test_pointnet2_victim_on_gpu that needs no
hipGraph -- PointNet++ SSG on the HIP kernels (FPS, the victims' ball query, grouping, the fused group layers on emulated matrix
instructions, rows_linear) against the REFERENCE's own run (fixture g11): the FPS table and the ball-query table bit for bit, the logits,
and the input gradient -- with the fused group-max layers and with the GEMM + max form."""
import os

import pytest
import torch

from helpers import T, close, golden, gradient_close

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("HITADV_EMULATE") != "1", reason="emulator-only: on a GPU the suite's own tests cover this")]


def test_pointnet2_on_the_hip_kernels_reproduces_the_reference_tables_logits_and_gradient():
    from hit_adv_amd.model import _pointwise
    from hit_adv_amd.model import pointnet2 as P2
    fx = golden('g11_pointnet2.npz')
    torch.manual_seed(int(fx['init_seed']))
    m = P2.get_model(40, normal_channel=False).eval().cuda()
    x = T(fx['x']).cuda().requires_grad_()
    pts = x.detach().transpose(1, 2).contiguous()
    torch.manual_seed(int(fx['fwd_seed']))
    fps1 = P2.farthest_point_sample(pts, 512)
    assert torch.equal(fps1.cpu(), T(fx['fps1']))
    assert torch.equal(P2.query_ball_point(0.2, 32, pts, P2.index_points(pts, fps1)).cpu(), T(fx['ball1']))
    torch.manual_seed(int(fx['fwd_seed']))
    logits, l3 = m(x)
    assert l3.shape == (2, 1024, 1)
    close(logits, fx['logits'], rtol=1e-4, atol=1e-5, what='PointNet++ logits vs the reference (g11)')
    (logits * T(fx['grad_w']).cuda()).sum().backward()
    gradient_close(x.grad, fx['grad_x'], 'PointNet++ input gradient vs the reference (g11)', frac_bound=1e-2, l2_bound=1.5e-3)
    try:
        _pointwise.FUSED_GROUP_MAX = False
        x2 = T(fx['x']).cuda().requires_grad_()
        torch.manual_seed(int(fx['fwd_seed']))
        logits2, _ = m(x2)
        (logits2 * T(fx['grad_w']).cuda()).sum().backward()
    finally:
        _pointwise.FUSED_GROUP_MAX = True
    close(logits2, fx['logits'], rtol=1e-4, atol=1e-5, what='PointNet++ logits vs the reference (g11), GEMM + max form')
    gradient_close(x2.grad, fx['grad_x'], 'PointNet++ input gradient vs the reference (g11), GEMM + max form', frac_bound=1e-3, l2_bound=1e-4)
