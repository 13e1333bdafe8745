#!/usr/bin/env python3
"""ops.group_linear_max forward (last shared layer + max over the neighbours) at cfg4's / cfg5's shapes, us per call and the rate at
which it reads its input."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hit_adv_amd import ops  # noqa: E402


def timed(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        f()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e3 / n


out = {}
g = torch.Generator().manual_seed(0)
for (G, ns, Cin, Cout) in ((32768, 32, 64, 128), (8192, 64, 128, 256), (16384, 32, 128, 128)):
    x = torch.randn(G, ns, Cin, generator=g).relu().cuda()
    W = (torch.randn(Cout, Cin, generator=g) * 0.1).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    pieces = (ops.split_weights_f16x2(W, range_flag=flag), ops.split_weights_f16x2(W.t().contiguous(), range_flag=flag))
    with torch.no_grad():
        us = timed(lambda: ops.group_linear_max(x, W, b, flag, pieces=pieces))
    out['%dx%dx%d->%d' % (G, ns, Cin, Cout)] = dict(us=round(us, 1), read_GBps=round(G * ns * Cin * 4 / us / 1e3))
print(json.dumps(out, indent=1))
