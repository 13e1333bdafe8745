#!/usr/bin/env python3
"""How the max-pool routes gradient in the bench workload: per (cloud, 64-point tile) and per point, the number of
channels whose arg-max lands there (the K of the backward gather).  Diagnostic for csrc/pointnet.hip."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hit_adv_amd.Dataset.synthetic import synth_batch  # noqa: E402
from hit_adv_amd import ops  # noqa: E402
from hit_adv_amd.model.pointnet import PointNetFeatureModel  # noqa: E402

torch.manual_seed(0)
m = PointNetFeatureModel(40, normal_channel=False).cuda().eval()
v = m.attack_view()
data, _ = synth_batch(32, 1024)
x = data[:, :, :3].transpose(1, 2).contiguous().cuda()
B, _, N = x.shape
R = B * N
a1, a2 = torch.empty(R, 64, device='cuda'), torch.empty(R, 128, device='cuda')
ops.pointnet_rowmlp_fwd(0, B, N, v.s2_w, v.s2_b, a2, x=x, W0=v.s1_w, b0=v.s1_b, o0=a1)
g, idx = ops.linear_max_fwd(a2, v.s3_w, B, N, bias=v.s3_b, relu=True)
for name, ix, gg in (("stn3d", idx, g),):
    live = gg > 0
    tile = ix // 64
    per_tile = torch.zeros(B, 16, dtype=torch.long, device='cuda')
    per_tile.scatter_add_(1, tile, live.long())
    per_pt = torch.zeros(B, N, dtype=torch.long, device='cuda')
    per_pt.scatter_add_(1, ix, live.long())
    print(name, "live channels/cloud %.0f" % live.sum(1).float().mean().item(),
          "| per tile: mean %.1f max %d p95 %.0f" % (per_tile.float().mean().item(), per_tile.max().item(),
                                                    per_tile.float().flatten().quantile(0.95).item()),
          "| per point: critical points/cloud %.0f, max channels at one point %d" %
          ((per_pt > 0).sum(1).float().mean().item(), per_pt.max().item()))
    print("per-tile counts of cloud 0:", per_tile[0].tolist())
