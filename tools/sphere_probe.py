#!/usr/bin/env python3
"""One stack of eight HiT-ADV attacks (1 x 200 iterations, PointNet engine) on Gaussian or surface-like clouds:
    python tools/sphere_probe.py gauss|sphere
Round 5: surface-like clouds spread the max-pool winners over many more points (more than 64 per 128-point tile in half the
tiles): the backward chain's second launch, made for a rare case, took 6x its work (docs/kernels/round5.md)."""
import sys, torch, warnings, time
sys.path.insert(0, '/root/repo')
import bench
from hit_adv_amd.Dataset.synthetic import sphere_batch, synth_batch
from hit_adv_amd.ShapeAttack.HiT_ADV import HiT_ADV
from hit_adv_amd.util.adv_utils import UntargetedLogitsAdvLoss
kind = sys.argv[1]
cfg = bench.CONFIGS['cfg2']; dev = torch.device('cuda', 0)
model = bench.build_victim(cfg).to(dev)
batches = []
for s in range(8):
    data, _ = (sphere_batch if kind == 'sphere' else synth_batch)(32, 1024, first=10**6 + 32 * s)
    data = data.to(dev)
    with torch.no_grad():
        label = bench.logits_of(model, data[:, :, :3].transpose(1, 2).contiguous()).argmax(1)
    batches.append((data, label))
att = HiT_ADV(model, adv_func=UntargetedLogitsAdvLoss(kappa=30.), binary_step=1, num_iter=200, verbose=False, **bench.HP)
torch.manual_seed(1)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    att.attack_many(batches[:8])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = att.attack_many(batches[:8])
    torch.cuda.synchronize()
print(kind, "seconds for 8 x (1 x 200)", round(time.perf_counter() - t0, 3), "succ", [int(k) for _, k in res])
