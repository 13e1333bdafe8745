#!/usr/bin/env python3
"""ops.rows_linear against torch's f32 GEMM (+ bias + ReLU epilogue) at cfg4's two middle layers, forward and input gradient."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from hit_adv_amd import _lib, ops  # noqa: E402


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


def main():
    out = {}
    for rows, C in ((1048576, 64), (524288, 128)):
        g = torch.Generator().manual_seed(0)
        x = torch.randn(rows, C, generator=g).relu().cuda()
        W = (torch.randn(C, C, generator=g) * 0.1).cuda()
        b = torch.randn(C, generator=g).cuda()
        W2 = ops.split_weights_f16x2(W)
        us = timed(lambda: ops.rows_linear(x, W2, b, True))
        ut = timed(lambda: torch._addmm_activation(b, x, W.t(), use_gelu=False))
        um = timed(lambda: x @ W)
        byt = 2.0 * rows * C * 4
        out['%dx%d->%d' % (rows, C, C)] = dict(rows_linear_us=round(us, 1), GBps=round(byt / us / 1e3, 0), torch_addmm_relu_us=round(ut, 1),
                                                torch_mm_us=round(um, 1))
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
