#!/usr/bin/env python3
"""HITADV_FPS_CHECK diagnostic: PointNet++ under HiT-ADV, short horizon, in four modes -- {eager, graphs} x {1, 2 attacks in
flight} -- counting the clouds whose FPS table differs between the two sampling kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
import bench
import fps_check
fps_check.install(order=os.environ.get('HITADV_FPS_ORDER', 'lean_first'), sync=os.environ.get('HITADV_FPS_SYNC'))
from hit_adv_amd.ShapeAttack.HiT_ADV import HiT_ADV
from hit_adv_amd.util.adv_utils import UntargetedLogitsAdvLoss
cfg = bench.CONFIGS['cfg4']
dev = torch.device('cuda', 0)
model = bench.build_victim(cfg).to(dev)
from hit_adv_amd.Dataset.synthetic import synth_batch
def batch(i):
    data, _ = synth_batch(cfg['B'], cfg['N'], first=100 * i)
    data = data.to(dev)
    with torch.no_grad():
        o = model(data[:, :, :3].transpose(1, 2).contiguous()); lab = (o[0] if isinstance(o, tuple) else o).argmax(1)
    return data, lab
bs = [batch(0), batch(1)]
for graph in (False,):
    for n in (1, 2):
        fps_check.reset()
        torch.manual_seed(5)
        att = HiT_ADV(model, adv_func=UntargetedLogitsAdvLoss(kappa=30.), binary_step=1, num_iter=40, verbose=False, use_graph=graph, **bench.HP)
        if n == 1:
            att.attack(*bs[0])
        else:
            att.attack_many(bs)
        torch.cuda.synchronize()
        if n == 2:
            os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
            torch.save(fps_check.captures(), os.path.join(ROOT, 'gpurun_out', 'fps_mismatch.pt'))
        print('graph', graph, 'in flight', n, fps_check.counts(), 'kernel counters', fps_check.diag_counters(), flush=True)
