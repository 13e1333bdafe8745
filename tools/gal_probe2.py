#!/usr/bin/env python3
"""hitadv_group_add_relu_linear against group_add_relu + rows_linear at cfg4's two levels, with the neighbour tables of real ball
queries (synthetic clouds) and with uniformly random ones."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hit_adv_amd import ops  # noqa: E402
from hit_adv_amd.Dataset.synthetic import synth_batch  # noqa: E402


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
    return round(t0.elapsed_time(t1) * 1e3 / n, 1)


out = {}
data, _ = synth_batch(64, 2048)
xyz = data[:, :, :3].contiguous().cuda()
g = torch.Generator().manual_seed(0)
flag = torch.zeros(1, dtype=torch.int32, device='cuda')
for (S, r, ns, C) in ((512, 0.2, 32, 64), (128, 0.4, 64, 128)):
    B, N, _ = xyz.shape
    fi = ops.fps_from_start(xyz, S, torch.zeros(B, dtype=torch.int64, device='cuda'))
    new = torch.gather(xyz, 1, fi.unsqueeze(-1).expand(-1, -1, 3))
    ball = ops.query_ball_point(r, ns, xyz, new)
    rnd = torch.randint(0, N, (B, S, ns), generator=g).cuda()
    U = torch.randn(B, N, C, generator=g).cuda()
    V = torch.randn(B, S, C, generator=g).cuda()
    W = (torch.randn(C, C, generator=g) * 0.1).cuda()
    b = torch.randn(C, generator=g).cuda()
    W2, Wt2 = ops.split_weights_f16x2(W), ops.split_weights_f16x2(W.t().contiguous())
    for name, idx in (('ball query', ball), ('uniform random', rnd)):
        with torch.no_grad():
            fused = timed(lambda: ops.GroupAddReLULinear.apply(U, V, idx, W2, Wt2, b, flag))
            gar = timed(lambda: ops.group_add_relu(U, V, idx))
            H = ops.group_add_relu(U, V, idx).reshape(-1, C)
            rl = timed(lambda: ops.rows_linear(H, W2, b, True, flag))
        out['N=%d S=%d ns=%d C=%d, %s' % (N, S, ns, C, name)] = dict(fused_us=fused, group_add_relu_us=gar, rows_linear_us=rl)
    xyz = new
print(json.dumps(out, indent=1))
