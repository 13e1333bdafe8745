"""FoldedPointNet (the attack-time execution plan of the PointNet victim) is the same function as the
module: logits, feature transform and input gradient agree to fp32 rounding.  Pure torch -> runs on CPU."""
import numpy as np
import torch

from helpers import golden_json, synth_batch


def _model():
    from hit_adv_amd.model.pointnet import PointNetFeatureModel
    torch.manual_seed(0)
    m = PointNetFeatureModel(40, normal_channel=False)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():  # non-trivial running statistics and affine parameters
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
                mod.running_var.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
                mod.weight.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
                mod.bias.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
    return m.eval()


def test_state_dict_layout_matches_reference():
    shapes = golden_json('g8_state_dicts.json')
    m = _model()
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == shapes['pointnet']
    assert sum(p.numel() for p in m.parameters()) == shapes['pointnet_param_count']


def test_folded_view_equals_module():
    m = _model()
    view = m.attack_view()
    data, _ = synth_batch(3, 256, first=5)
    x = data[:, :, :3].transpose(1, 2).contiguous()
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    la, ta = m(xa)
    lb, tb = view(xb)
    np.testing.assert_allclose(lb.detach(), la.detach(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(tb.detach(), ta.detach(), rtol=1e-4, atol=1e-5)
    w = torch.randn(3, 40, generator=torch.Generator().manual_seed(2))
    ga, = torch.autograd.grad((la * w).sum(), xa)
    gb, = torch.autograd.grad((lb * w).sum(), xb)
    np.testing.assert_allclose(gb, ga, rtol=1e-3, atol=1e-5 * float(ga.abs().max()))
    assert (la.argmax(1) == lb.argmax(1)).all()
