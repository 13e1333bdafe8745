#!/usr/bin/env python3
"""After the packed-f32 finding in the FPS kernel (docs/kernels/round5.md section 8): do the kNN tables of DGCNN's forward pass
(knn_select / knn_feat_k, which use packed f32 instructions too) repeat when they are computed twice in a row with other attacks in
flight?  Every `ops.knn_features` / `ops.knn_points` call is issued twice and the index tables are compared on the device."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from hit_adv_amd import ops
from hit_adv_amd.ShapeAttack.HiT_ADV import HiT_ADV
from hit_adv_amd.util.adv_utils import UntargetedLogitsAdvLoss
from hit_adv_amd.Dataset.synthetic import synth_batch
counts = {}
def wrap(name):
    fn = getattr(ops, name)
    def twice(*a, **k):
        r1 = fn(*a, **k)
        r2 = fn(*a, **k)
        i1 = r1[1] if isinstance(r1, tuple) else r1
        i2 = r2[1] if isinstance(r2, tuple) else r2
        c = counts.setdefault(name, torch.zeros(2, dtype=torch.int64, device=i1.device))
        c[0] += (i1 != i2).reshape(i1.shape[0], -1).any(dim=1).sum()
        c[1] += i1.shape[0]
        return r1
    setattr(ops, name, twice)
for n in ('knn_features',):
    wrap(n)
cfg = bench.CONFIGS['cfg3']
dev = torch.device('cuda', 0)
model = bench.build_victim(cfg).to(dev)
def batch(i):
    data, _ = synth_batch(cfg['B'], cfg['N'], first=100 * i)
    data = data.to(dev)
    with torch.no_grad():
        o = model(data[:, :, :3].transpose(1, 2).contiguous()); lab = (o[0] if isinstance(o, tuple) else o).argmax(1)
    return data, lab
bs = [batch(i) for i in range(4)]
for n in (1, 4):
    for c in counts.values(): c.zero_()
    torch.manual_seed(5)
    att = HiT_ADV(model, adv_func=UntargetedLogitsAdvLoss(kappa=30.), binary_step=1, num_iter=40, verbose=False, use_graph=False, **bench.HP)
    if n == 1: att.attack(*bs[0])
    else: att.attack_many(bs[:n])
    torch.cuda.synchronize()
    print('in flight', n, {k: v.tolist() for k, v in counts.items()}, flush=True)
