#!/usr/bin/env python3
"""How many sample-and-group lists a point appears in (cfg4's two levels, synthetic clouds): the length of the chain
group_add_relu_du_k walks per target point."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hit_adv_amd import ops  # noqa: E402
from hit_adv_amd.Dataset.synthetic import synth_batch  # noqa: E402

data, _ = synth_batch(64, 2048)
xyz = data[:, :, :3].contiguous().cuda()
for (S, r, ns) in ((512, 0.2, 32), (128, 0.4, 64)):
    B, N, _ = xyz.shape
    fi = ops.fps_from_start(xyz, S, torch.zeros(B, dtype=torch.int64, device='cuda'))
    new = torch.gather(xyz, 1, fi.long().unsqueeze(-1).expand(-1, -1, 3))
    idx = ops.query_ball_point(r, ns, xyz, new)  # [B,S,ns]
    # lists containing point j (unique per list)
    deg = torch.zeros(B, N, dtype=torch.int64, device='cuda')
    for b in range(0, B, 8):
        for s in range(S):
            u = torch.unique(idx[b, s])
            deg[b, u] += 1
    d = deg.float()
    print('N=%d S=%d ns=%d: lists per point mean %.1f, median %.0f, p99 %.0f, max %.0f' % (N, S, ns, d[0:B:8].mean(), d[0:B:8].median(), d[0:B:8].quantile(0.99), d[0:B:8].max()))
    xyz = new
