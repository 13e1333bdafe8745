#!/usr/bin/env python3
"""ms per iteration of the CW-family attacks at B=32, N=1024 (cfg5 of BASELINE.json sweeps them): CWPerturb (L2), CWKNN
(Chamfer + kNN distance, clipped), CWAOF, per victim.  `python tools/cw_breakdown.py [victim ...]`"""
import json
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from hit_adv_amd.Dataset.synthetic import synth_batch  # noqa: E402
from victim_breakdown import build  # noqa: E402


ITERS = 300


def attacks(model):
    from hit_adv_amd.CW import CWAOF, CWKNN, CWPerturb
    from hit_adv_amd.util.adv_utils import LogitsAdvLoss
    from hit_adv_amd.util.clip_utils import ProjectInnerClipLinf
    from hit_adv_amd.util.dist_utils import ChamferkNNDist, L2Dist
    adv = LogitsAdvLoss(kappa=30.)
    yield 'CWPerturb', ITERS, CWPerturb(model, adv, L2Dist(), attack_lr=1e-2, binary_step=1, num_iter=ITERS, verbose=False)
    yield 'CWKNN', ITERS, CWKNN(model, adv, ChamferkNNDist(), ProjectInnerClipLinf(budget=0.18), attack_lr=1e-3, num_iter=ITERS,
                             verbose=False)
    yield 'CWAOF', ITERS, CWAOF(model, adv, L2Dist(), attack_lr=1e-2, binary_step=1, num_iter=ITERS, clip_func=ProjectInnerClipLinf(budget=0.18),
                             verbose=False)


def main():
    names = sys.argv[1:] or ['pointnet', 'dgcnn']
    data, _ = synth_batch(32, 1024)
    data = data.cuda()
    out = {}
    for name in names:
        m = build(name, 5).cuda().eval()
        with torch.no_grad():
            o = m(data[:, :, :3].transpose(1, 2).contiguous())
            target = ((o[0] if isinstance(o, tuple) else o).argmax(1) + 1) % 40
        for label, iters, att in attacks(m):
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                try:
                    att.attack(data, target)  # warm-up: library handles, solver search
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    att.attack(data, target)
                    torch.cuda.synchronize()
                    out['%s/%s' % (name, label)] = [round((time.perf_counter() - t0) / iters * 1e3, 3),
                                                    'graph' if getattr(att, 'last_graph_used', False) else 'eager']
                except Exception as e:  # noqa: BLE001
                    out['%s/%s' % (name, label)] = 'failed: %s' % (str(e)[:80],)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
