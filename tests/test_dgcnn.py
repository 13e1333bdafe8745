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
