#!/usr/bin/env python3
"""What a cloud boundary costs inside V1 (fp16x2 form).  A workgroup that takes several clouds in turn drains its tile pipeline
at the end of each one (scan merge, store, first tile of the next cloud fetched and waited for).  The same number of 64-point
tiles per workgroup is timed as many short clouds and as few long ones; the difference is the boundaries."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from hit_adv_amd import _lib, ops  # noqa: E402


def main():
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    _p = bench._p
    g = torch.Generator().manual_seed(1)
    rows = 256 * 1024
    h2 = torch.randn(rows, 128, generator=g).relu().to(dev)
    Wt = (torch.randn(128, 1024, generator=g) * 0.1).to(dev)
    bias = torch.randn(1024, generator=g).to(dev)
    W2 = ops.split_weights_f16x2(Wt.t().contiguous())
    tk = torch.zeros(4096, device=dev, dtype=torch.int32)
    out = {}
    for blocks in (128, 0):
        for B in (256, 128, 64, 32):
            N = rows // B
            n3 = lib.hitadv_linear_max_fwd_bf16x3_scratch(B, N, 1024, blocks)
            pv, pi = torch.empty(n3, device=dev), torch.empty(n3, device=dev, dtype=torch.int32)
            mo, mi = torch.empty(B, 1024, device=dev), torch.empty(B, 1024, device=dev, dtype=torch.int64)
            us = bench.graph_timed(lambda st: lib.hitadv_linear_max_fwd_f16x2(
                _p(h2), _p(W2), _p(bias), B, N, 128, 1024, 1, blocks, _p(pv), _p(pi), _p(mo), _p(mi), _p(tk), None, st))
            out['blocks=%d B=%d N=%d' % (blocks or 256, B, N)] = round(us, 2)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
