#!/usr/bin/env python3
"""The PointNet engine's forward shared-layer kernel (V2) in its two forms -- 0: rowmlp_stream_k (round 5), 1: rowmlp_fwd16_k
(one tile per workgroup) -- at the stacked launch size (B = 256) and at one attack's (B = 32), mode 2 (packed pieces), each
stage as the engine launches it (stage 0 with the deformation inside, stage 1 with STN3d's last layer inside).  Prints JSON:
us per launch and the algorithmic bytes moved per second.     python tools/v2_probe.py [B ...]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hit_adv_amd import _lib, ops  # noqa: E402

lib = _lib.load()
p = bench._p
g = torch.Generator().manual_seed(0)
dev = torch.device('cuda', 0)
N, C = 1024, 192
out = {}
for B in [int(a) for a in sys.argv[1:]] or [256, 32]:
    R = B * N
    ori = (torch.randn(B, 3, N, generator=g) * 0.4).to(dev)
    central = ori[:, :, :C].contiguous()
    P, S = ((torch.rand(B, C, 3, generator=g) - 0.5) * 0.5).to(dev), (0.1 + 1.1 * torch.rand(B, C, generator=g)).to(dev)
    W0, b0 = torch.randn(3, 64, generator=g).to(dev), torch.randn(64, generator=g).to(dev)
    W1, b1 = (torch.randn(64, 64, generator=g) * 0.2).to(dev), torch.randn(64, generator=g).to(dev)
    W2, b2 = (torch.randn(64, 128, generator=g) * 0.2).to(dev), torch.randn(128, generator=g).to(dev)
    T64 = (torch.eye(64).repeat(B, 1, 1) + 0.05 * torch.randn(B, 64, 64, generator=g)).to(dev).contiguous()
    F5, W6, b6 = torch.randn(B, 256, generator=g).relu().to(dev), (torch.randn(256, 9, generator=g) * 0.05).to(dev), torch.randn(9, generator=g).to(dev)
    hin = torch.randn(R, 64, generator=g).relu().to(dev)
    adv, inv = torch.empty_like(ori), torch.empty(B, N, device=dev)
    o0, o1, o2, xp, Tout = (torch.empty(R, 64, device=dev), torch.empty(R, 64, device=dev), torch.empty(R, 128, device=dev),
                            torch.empty(R, 3, device=dev), torch.empty(B, 9, device=dev))
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    stages = {
        "stage0_deform": (lambda s: lib.hitadv_pointnet_rowmlp_fwd_deform(p(ori), p(central), p(P), p(S), C, p(adv), p(inv), p(W0), p(b0), p(W2), p(b2),
                                                                         p(o0), p(o2), B, N, 2, p(flag), s), R * (64 + 128) * 4 + R * 4 * 7),
        "stage0_plain": (lambda s: lib.hitadv_pointnet_rowmlp_fwd(0, p(ori), None, None, p(W0), p(b0), None, None, p(W2), p(b2), None, p(o0), None, p(o2),
                                                                  B, N, 2, p(flag), s), R * (64 + 128) * 4 + R * 12),
        "stage1_stn": (lambda s: lib.hitadv_pointnet_rowmlp_fwd_stn(p(ori), p(F5), p(W6), p(b6), p(Tout), p(W0), p(b0), p(W1), p(b1), p(W2), p(b2), p(xp),
                                                                    p(o0), p(o1), p(o2), B, N, 2, p(flag), s), R * (64 + 64 + 128) * 4 + R * 24),
        "stage2": (lambda s: lib.hitadv_pointnet_rowmlp_fwd(2, None, p(T64), p(hin), None, None, None, None, p(W2), p(b2), None, p(o0), None, p(o2),
                                                            B, N, 2, p(flag), s), R * (64 + 64 + 128) * 4),
    }
    for form in (1, 0):
        lib.hitadv_pointnet_rowmlp_form(form)
        for name, (launch, nbytes) in stages.items():
            us = bench.graph_timed(launch)
            out["B%d %s form%d" % (B, name, form)] = dict(us=round(us, 2), TBps=round(nbytes / us / 1e6, 3))
    lib.hitadv_pointnet_rowmlp_form(0)
print(json.dumps(out, indent=1))
