#!/usr/bin/env python3
"""ms per HiT-ADV inner iteration for each victim at B=32, N=1024 (graph replay where capturable)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hit_adv_amd.Dataset.synthetic import synth_batch  # noqa: E402
from hit_adv_amd.ShapeAttack.HiT_ADV import HiT_ADV  # noqa: E402
from hit_adv_amd.util.adv_utils import UntargetedLogitsAdvLoss  # noqa: E402

HP = dict(attack_lr=1e-2, central_num=192, total_central_num=256, curv_loss_knn=16, max_sigm=1.2, min_sigm=0.1,
          budget=0.55, cd_weight=1e-4, ker_weight=1., hide_weight=1.)


def build(name, k):
    torch.manual_seed(0)
    if name == 'pointnet':
        from hit_adv_amd.model.pointnet import PointNetFeatureModel
        return PointNetFeatureModel(40, normal_channel=False)
    if name == 'dgcnn':
        from hit_adv_amd.model.dgcnn import DGCNN_cls
        return DGCNN_cls(argparse.Namespace(k=k, emb_dims=1024, dropout=0.2), output_channels=40)
    if name == 'pointnet++':
        from hit_adv_amd.model.pointnet2 import get_model
        return get_model(40, normal_channel=False)
    from hit_adv_amd.model.pct import Pct
    return Pct(argparse.Namespace(dropout=0.2), output_channels=40)


def main():
    out = {}
    data, _ = synth_batch(32, 1024)
    data = data.cuda()
    for name, k in (('pointnet', 0), ('dgcnn', 5), ('dgcnn', 20), ('pointnet++', 0), ('pct', 0)):
        m = build(name, k).cuda().eval()
        with torch.no_grad():
            o = m(data[:, :, :3].transpose(1, 2).contiguous())
            label = (o[0] if isinstance(o, tuple) else o).argmax(1)
        iters = 60
        att = HiT_ADV(m, UntargetedLogitsAdvLoss(30.), verbose=False, binary_step=1, num_iter=iters, **HP)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            att.attack(data, label)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            att.attack(data, label)
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        x = data[:, :, :3].transpose(1, 2).contiguous().requires_grad_()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            o = m(x)
            lo = o[0] if isinstance(o, tuple) else o
            torch.autograd.grad(lo.sum(), x)
        torch.cuda.synchronize()
        out['%s%s' % (name, '_k%d' % k if k else '')] = dict(graph=att.last_graph_used, attack_s=round(dt, 3),
                                                            ms_per_iter_incl_setup=round(dt / iters * 1e3, 3),
                                                            victim_fwd_bwd_eager_ms=round((time.perf_counter() - t1) * 100, 3))
        del att, m
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
