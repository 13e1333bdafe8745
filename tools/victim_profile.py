#!/usr/bin/env python3
"""A short HiT-ADV attack on one victim, meant to be run under rocprofv3 (kernel breakdown of an iteration):

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 tools/victim_profile.py dgcnn 5 40
    python tools/prof_summary.py gpurun_out/prof profiles/<name>.csv <iterations + 3>

victim in {pointnet, dgcnn, pointnet++, pct}; k = DGCNN's neighbour count (ignored otherwise); iterations of one binary step.
(The first eager iterations include MIOpen's solver search for victims that still call it -- ignore naive_conv / igemm /
ck rows with ~1 call per iteration when reading a steady-state breakdown.)"""
import os
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from hit_adv_amd.Dataset.synthetic import synth_batch  # noqa: E402
from hit_adv_amd.ShapeAttack.HiT_ADV import HiT_ADV  # noqa: E402
from hit_adv_amd.util.adv_utils import UntargetedLogitsAdvLoss  # noqa: E402
from victim_breakdown import HP, build  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'pointnet'
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    m = build(name, k).cuda().eval()
    data, _ = synth_batch(32, 1024)
    data = data.cuda()
    with torch.no_grad():
        o = m(data[:, :, :3].transpose(1, 2).contiguous())
        label = (o[0] if isinstance(o, tuple) else o).argmax(1)
    att = HiT_ADV(m, UntargetedLogitsAdvLoss(30.), verbose=False, binary_step=1, num_iter=iters, **HP)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        att.attack(data, label)
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()
