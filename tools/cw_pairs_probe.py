#!/usr/bin/env python3
"""Which of cfg5's attacks overlap with which when run through CW.attack_concurrently (PCT victim, B = 32): every set is
timed in sequence and in flight at once.   gpurun -- python tools/cw_pairs_probe.py"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from hit_adv_amd import CW  # noqa: E402
from hit_adv_amd.util.adv_utils import LogitsAdvLoss, UntargetedLogitsAdvLoss  # noqa: E402
from hit_adv_amd.util.clip_utils import ClipPointsLinf  # noqa: E402
from hit_adv_amd.util.dist_utils import ChamferkNNDist, L2Dist  # noqa: E402

cfg = bench.CONFIGS['cfg5']
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
model = bench.build_victim(cfg).to(dev)
torch.manual_seed(2)
ae = bench.ToyAE().eval().to(dev)
clip = ClipPointsLinf(budget=0.18)
data, _ = bench.synth(0, 32, 1024)
xyz = data[:, :, :3].contiguous().to(dev)
with torch.no_grad():
    label = bench.logits_of(model, xyz.transpose(1, 2).contiguous()).argmax(1)
target = (label + 1) % 40


def make(kind):
    kw = dict(verbose=False)
    if kind == 'advpc':
        return CW.CWAdvPC(model, ae, LogitsAdvLoss(kappa=0.), L2Dist(), clip_func=clip, binary_step=1, num_iter=16, **kw), (xyz, target, label)
    if kind == 'knn':
        return CW.CWKNN(model, LogitsAdvLoss(kappa=15.), ChamferkNNDist(), clip, num_iter=48, **kw), (xyz, target)
    if kind == 'knn_l2':  # the kNN loop without its distance kernels
        return CW.CWKNN(model, LogitsAdvLoss(kappa=15.), L2Dist(), clip, num_iter=48, **kw), (xyz, target)
    return CW.CWAOF(model, UntargetedLogitsAdvLoss(kappa=30.), L2Dist(), clip_func=clip, binary_step=1, num_iter=16, **kw), (xyz, label)


def timed(kinds, together):
    calls = [make(k) for k in kinds]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if together:
        CW.attack_concurrently(calls)
    else:
        for a, args in calls:
            a.attack(*args)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


res = {}
timed(['advpc', 'knn', 'aof'], False)
for kinds in (['knn', 'knn', 'knn'], ['knn_l2', 'knn_l2', 'knn_l2'], ['advpc', 'advpc', 'advpc'], ['aof', 'aof', 'aof'], ['advpc', 'knn', 'aof']):
    seq = timed(kinds, False)
    par = timed(kinds, True)
    res['+'.join(kinds)] = dict(sequence_s=round(seq, 3), in_flight_s=round(par, 3), ratio=round(par / seq, 3))
    print(json.dumps({'+'.join(kinds): res['+'.join(kinds)]}), flush=True)
