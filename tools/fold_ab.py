#!/usr/bin/env python3
"""A/B on one box: a HiT-ADV iteration (graph replay, cfg2 sizes) with the small fully connected layers folded into their
neighbours' launches (FoldedPointNet.fold_small_layers) and with one launch per layer."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hit_adv_amd.Dataset.synthetic import synth_batch  # noqa: E402
from hit_adv_amd.ShapeAttack.HiT_ADV import HiT_ADV  # noqa: E402
from hit_adv_amd.model.pointnet import FoldedPointNet, PointNetFeatureModel  # noqa: E402
from hit_adv_amd.util.adv_utils import UntargetedLogitsAdvLoss  # noqa: E402

HP = dict(attack_lr=1e-2, central_num=192, total_central_num=256, curv_loss_knn=16, max_sigm=1.2, min_sigm=0.1, budget=0.55,
          cd_weight=1e-4, ker_weight=1., hide_weight=1.)


def run(fold, data, label, model):
    FoldedPointNet.fold_small_layers = fold
    att = HiT_ADV(model, UntargetedLogitsAdvLoss(30.), verbose=False, binary_step=1, num_iter=20, iterations_per_graph=1, **HP)
    att.attack(data, label)
    ws = next(iter(att._ws.values()))
    att._prepare_graphs([ws])
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(300):
            ws.graph.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 300 * 1e6)
    return best


def main():
    torch.manual_seed(0)
    model = PointNetFeatureModel(40, normal_channel=False).cuda().eval()
    data, _ = synth_batch(32, 1024)
    data = data.cuda()
    with torch.no_grad():
        label = model(data[:, :, :3].transpose(1, 2).contiguous())[0].argmax(1)
    out = {}
    for rep in range(2):
        for fold in (True, False):
            out['%s_%d_us' % ('folded' if fold else 'per_layer', rep)] = round(run(fold, data, label, model), 2)
    print(json.dumps(out))




def micro():
    """The folded launches against the launches they replace, 20 per graph."""
    from hit_adv_amd import ops
    g = torch.Generator().manual_seed(0)
    cu = lambda t: t.cuda()  # noqa: E731
    B = 32
    dl, W3r, W2r, f2 = cu(torch.randn(B, 40, generator=g)), cu(torch.randn(40, 256, generator=g)), cu(torch.randn(256, 512, generator=g)), cu(torch.randn(B, 256, generator=g))
    dTp, W6r, W5r = cu(torch.randn(B, 16, 9, generator=g)), cu(torch.randn(9, 256, generator=g)), cu(torch.randn(256, 512, generator=g))
    x, f5, W6, b6 = cu(torch.randn(B, 3, 1024, generator=g)), cu(torch.randn(B, 256, generator=g)), cu(torch.randn(256, 9, generator=g)), cu(torch.randn(9, generator=g))
    W0, b0, W1, b1, W2, b2 = (cu(torch.randn(*s, generator=g)) for s in ((3, 64), (64,), (64, 64), (64,), (64, 128), (128,)))
    o0, o1, o2, T3 = (torch.empty(*s, device='cuda') for s in ((B * 1024, 64), (B * 1024, 64), (B * 1024, 128), (B, 9)))
    cases = {
        'head_folded': lambda: ops.fc_layer_pre(dl.unsqueeze(1), W3r, W2r, mask=f2),
        'head_two_launches': lambda: ops.fc_layer(ops.fc_layer(dl, W3r), W2r, mask=f2),
        'stn_folded': lambda: ops.fc_layer_pre(dTp, W6r, W5r, mask=f2),
        'stn_three_launches': lambda: ops.fc_layer(ops.fc_layer(ops.sum_partials(dTp), W6r), W5r, mask=f2),
        'stage1_folded': lambda: ops.pointnet_rowmlp_fwd_stn(B, 1024, x, f5, W6, b6, T3, W0, b0, W1, b1, W2, b2, o0, o1, o2),
        'stage1_two_launches': lambda: ops.pointnet_rowmlp_fwd(1, B, 1024, W2, b2, o2, x=x, T=ops.fc_layer(f5, W6, b6), W0=W0, b0=b0, W1=W1, b1=b1, o0=o0, o1=o1),
    }
    out = {}
    s = torch.cuda.Stream()
    for name, fn in cases.items():
        with torch.cuda.stream(s):
            fn()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for _ in range(20):
                    fn()
            gr.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                gr.replay()
            torch.cuda.synchronize()
            out[name + '_us'] = round((time.perf_counter() - t0) / 400 * 1e6, 2)
    print(json.dumps(out))


if __name__ == '__main__':
    micro() if len(sys.argv) > 1 and sys.argv[1] == 'micro' else main()
