"""DGCNN victim: layout compatibility with the reference's checkpoints and, on CPU (pure torch path),
equality with the reference's logits / input gradient for the same seeded initialisation (fixture g10).
The GPU path (HIP kNN / top-k) is covered in test_gpu_attack.py."""
import argparse

import numpy as np
import torch

from helpers import T, golden, golden_json


def build(seed=31):
    from hit_adv_amd.model.dgcnn import DGCNN_cls
    torch.manual_seed(seed)
    return DGCNN_cls(argparse.Namespace(k=5, emb_dims=1024, dropout=0.2), output_channels=40).eval()


def test_state_dict_layout_matches_reference():
    shapes = golden_json('g8_state_dicts.json')
    m = build()
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == shapes['dgcnn']


def test_cpu_forward_backward_equals_reference():
    from hit_adv_amd.model.dgcnn import get_graph_feature, knn
    fx = golden('g10_dgcnn.npz')
    m = build(int(fx['seed']))  # same construction order -> same RNG draws -> same weights as the reference model
    x = T(fx['x']).clone().requires_grad_()
    logits = m(x)
    np.testing.assert_allclose(logits.detach(), fx['logits'], rtol=1e-4, atol=1e-5)
    (logits * T(fx['grad_w'])).sum().backward()
    np.testing.assert_allclose(x.grad, fx['grad_x'], rtol=1e-3, atol=1e-6)
    assert (knn(T(fx['x']), 5).numpy() == fx['knn_layer1']).all()
    np.testing.assert_array_equal(get_graph_feature(T(fx['x']), k=5).numpy(), fx['edge_layer1'])


def test_attack_view_equals_module_and_reference_on_cpu():
    """FoldedDGCNN (BatchNorm folded, EdgeConv split into per-point products + neighbour max) computes the module's
    function: against the reference's logits / input gradient of fixture g10 and, in float64 with non-trivial BatchNorm
    statistics, against the module itself."""
    fx = golden('g10_dgcnn.npz')
    m = build(int(fx['seed']))
    view = m.attack_view()
    x = T(fx['x']).clone().requires_grad_()
    logits = view(x)
    np.testing.assert_allclose(logits.detach(), fx['logits'], rtol=1e-4, atol=2e-5)
    (logits * T(fx['grad_w'])).sum().backward()
    # fp32: the re-associated EdgeConv rounds differently, so a few near-tied maxima (over neighbours / points) route
    # their gradient through another point than in the reference; the float64 comparison below has no such ties
    g, r = x.grad.numpy(), fx['grad_x']
    assert np.linalg.norm(g - r) <= 1e-2 * np.linalg.norm(r)
    assert np.isclose(g, r, rtol=2e-3, atol=1e-5 * np.abs(r).max()).mean() > 0.95
    md = build(7).double()
    with torch.no_grad():
        for mod in md.modules():
            if isinstance(mod, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                mod.running_mean.normal_(0, 0.2)
                mod.running_var.uniform_(0.5, 1.5)
                mod.weight.normal_(0, 1.0)   # negative scales too: the max/LeakyReLU exchange must survive them
                mod.bias.normal_(0, 0.2)
    vd = md.attack_view()
    g = torch.Generator().manual_seed(3)
    xd = torch.randn(2, 3, 96, generator=g, dtype=torch.float64).requires_grad_()
    xe = xd.detach().clone().requires_grad_()
    wl = torch.randn(2, 40, generator=g, dtype=torch.float64)
    la, lb = md(xd), vd(xe)
    np.testing.assert_allclose(lb.detach(), la.detach(), rtol=1e-9, atol=1e-11)
    ga, = torch.autograd.grad((la * wl).sum(), xd)
    gb, = torch.autograd.grad((lb * wl).sum(), xe)
    np.testing.assert_allclose(gb, ga, rtol=1e-7, atol=1e-10)
    md.bn1.weight.data.mul_(2.0)
    lc = vd.refresh(md)(xe)
    np.testing.assert_allclose(lc.detach(), md(xd).detach(), rtol=1e-9, atol=1e-11)


def test_pointnet2_state_dict_layout_matches_reference():
    from hit_adv_amd.model.pointnet2 import get_model
    shapes = golden_json('g8_state_dicts.json')
    m = get_model(40, normal_channel=False)
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == shapes['pointnet++']


def test_pct_state_dict_layout_matches_reference():
    from hit_adv_amd.model.pct import Pct
    shapes = golden_json('g8_state_dicts.json')
    m = Pct(argparse.Namespace(dropout=0.2), output_channels=40)
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == shapes['pct']
