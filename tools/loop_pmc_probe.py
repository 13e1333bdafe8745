#!/usr/bin/env python3
"""cfg2's hot loop as the headline runs it -- 24 attacks as THREE stacks of eight (256 clouds per victim pass), the default
fp16x2 engine -- for a handful of iterations, meant for rocprofv3 --pmc passes (tools/r05_measure.sh pmc_loop): per stack 2
eager warm-up iterations + 11 replayed ones = 39 launches of every loop kernel (setup kernels come in multiples of 24)."""
import os
import sys
import warnings

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hit_adv_amd.ShapeAttack.HiT_ADV import HiT_ADV  # noqa: E402
from hit_adv_amd.util.adv_utils import UntargetedLogitsAdvLoss  # noqa: E402

ITERATIONS = 13
cfg = bench.CONFIGS['cfg2']
dev = torch.device('cuda', 0)
model = bench.build_victim(cfg).to(dev)
batches = []
for s in range(24):
    data, _ = bench.synth(s * 32, 32, 1024)
    data = data.to(dev)
    with torch.no_grad():
        label = bench.logits_of(model, data[:, :, :3].transpose(1, 2).contiguous()).argmax(1)
    batches.append((data, label))
att = HiT_ADV(model, adv_func=UntargetedLogitsAdvLoss(kappa=30.), binary_step=1, num_iter=ITERATIONS - 2, verbose=False, **bench.HP)
torch.manual_seed(1)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    res = att.attack_many(batches)
torch.cuda.synchronize()
print("stacked:", att.stacks(), "graph:", att.last_graph_used, "successes:", [int(k) for _, k in res])
