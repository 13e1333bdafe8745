#!/usr/bin/env python3
"""A short slice of cfg5's sweep (16 + 48 + 16 captured iterations of CWAdvPC / CWKNN / CWAOF on PCT, B = 32), for a
kernel trace that fits:   rocprofv3 --kernel-trace --output-format csv -d /tmp/prof/c5 -- python3 tools/cfg5_short.py [--sequential-sweep]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

bench.SEQUENTIAL_SWEEP = '--sequential-sweep' in sys.argv
cfg = bench.CONFIGS['cfg5']
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
model = bench.build_victim(cfg).to(dev)
run, prewarm, info, _ = bench.make_runner(cfg, model, dev, 1)
data, _ = bench.synth(0, cfg['B'], cfg['N'])
data = data.to(dev)
with torch.no_grad():
    label = bench.logits_of(model, data[:, :, :3].transpose(1, 2).contiguous()).argmax(1)
prewarm((data, label))
torch.cuda.synchronize()
t0 = time.perf_counter()
prewarm.profiled((data, label))
torch.cuda.synchronize()
print("short sweep: %.3f s (%s)" % (time.perf_counter() - t0, "sequence" if bench.SEQUENTIAL_SWEEP else "three in flight"))
