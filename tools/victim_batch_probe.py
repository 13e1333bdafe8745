#!/usr/bin/env python3
"""How the PointNet engine's forward + input-gradient pass scales with the batch: one captured pass at B = 32, 64, 128, 256
clouds of 1024 points, microseconds per pass and per 32 clouds.  (Latency-bound kernels -- the shared-layer chains, the
FC stacks -- cost the same for 128 clouds as for 32; the matrix-bound 128 -> 1024 layers scale linearly.)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hit_adv_amd.Dataset.synthetic import synth_batch  # noqa: E402
from hit_adv_amd.model.pointnet import PointNetFeatureModel  # noqa: E402


def main():
    torch.manual_seed(0)
    model = PointNetFeatureModel(40, normal_channel=False).cuda().eval()
    view = model.attack_view()
    out = {}
    for B in ([int(sys.argv[1])] if len(sys.argv) > 1 else (32, 64, 128, 256)):
        data, _ = synth_batch(B, 1024, first=0)
        x = data[:, :, :3].transpose(1, 2).contiguous().cuda().requires_grad_()
        w = torch.randn(B, 40, device='cuda')

        def body():
            logits, _ = view(x)
            g, = torch.autograd.grad(logits, x, grad_outputs=w)
            return g
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            body(); body()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(5):
                body()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(40):
            g.replay()
        t1.record()
        torch.cuda.synchronize()
        us = t0.elapsed_time(t1) * 1e3 / 200
        out[B] = dict(us_per_pass=round(us, 1), us_per_32_clouds=round(us * 32 / B, 1))
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
