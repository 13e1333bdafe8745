#!/usr/bin/env python3
"""group_add_relu forward + backward (reverse table, dV, dU) and edge_max forward at the victims' shapes, eager, us per call."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hit_adv_amd import ops  # noqa: E402
from hit_adv_amd.Dataset.synthetic import synth_batch  # noqa: E402


def timed(f, n=10):
    for _ in range(2):
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
for (S, r, ns, C) in ((512, 0.2, 32, 64), (128, 0.4, 64, 128)):
    B, N, _ = xyz.shape
    fi = ops.fps_from_start(xyz, S, torch.zeros(B, dtype=torch.int64, device='cuda'))
    new = torch.gather(xyz, 1, fi.unsqueeze(-1).expand(-1, -1, 3))
    idx = ops.query_ball_point(r, ns, xyz, new)
    U = torch.randn(B, N, C, generator=g).cuda().requires_grad_()
    V = torch.randn(B, S, C, generator=g).cuda().requires_grad_()
    w = torch.randn(B, S, ns, C, generator=g).cuda()
    H = ops.group_add_relu(U, V, idx)
    out['group_add_relu fwd N=%d S=%d ns=%d C=%d' % (N, S, ns, C)] = timed(lambda: ops.group_add_relu(U.detach(), V.detach(), idx))
    out['group_add_relu bwd N=%d S=%d ns=%d C=%d' % (N, S, ns, C)] = timed(lambda: torch.autograd.grad(H, [U, V], w, retain_graph=True))
    xyz = new
B, N, k = 32, 1024, 5
for C in (64, 128, 256):
    UV = torch.randn(B, N, 2 * C, generator=g).cuda()
    idx = torch.randint(0, N, (B, N, k), generator=g).cuda()
    out['edge_max fwd C=%d' % C] = timed(lambda: ops.edge_max_fused(UV, idx, 0.2))
print(json.dumps(out, indent=1))
