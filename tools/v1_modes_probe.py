#!/usr/bin/env python3
"""V1 (128 -> 1024 shared layer + max over the points) in its three forms -- f32 MFMA, bf16x3, fp16x2 -- on the bench shape:
error against float64 (max over the [B,Cout] maxima, and the error of the whole product on a sample) and time per launch."""
import ctypes
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
    out = {}
    for scale_x, scale_w, tag in ((1.0, 0.1, 'unit'), (1e-3, 0.1, 'small activations'), (30.0, 1.0, 'large')):
        B, N, Cin, Cout = 32, 1024, 128, 1024
        g = torch.Generator().manual_seed(7)
        x = (torch.randn(B * N, Cin, generator=g).relu() * scale_x)
        Wt = torch.randn(Cin, Cout, generator=g) * scale_w
        y = (x.double() @ Wt.double()).view(B, N, Cout)
        ref = y.max(dim=1).values
        scale = float(ref.abs().max())
        xc, Wc = x.to(dev), Wt.to(dev)
        W3 = ops.split_weights_bf16x3(Wc.t().contiguous())
        flag = torch.zeros(1, device=dev, dtype=torch.int32)
        W2 = ops.split_weights_f16x2(Wc.t().contiguous(), range_flag=flag)
        res = {}
        v32, i32 = ops.linear_max_fwd(xc, Wc, B, N)
        vb3, ib3 = ops.linear_max_fwd_bf16x3(xc, W3, B, N)
        vh2, ih2 = ops.linear_max_fwd_f16x2(xc, W2, B, N, range_flag=flag)
        for name, v, i in (('f32', v32, i32), ('bf16x3', vb3, ib3), ('fp16x2', vh2, ih2)):
            e = (v.cpu().double() - ref).abs()
            res[name] = dict(max_err=float(e.max()), max_err_over_scale=float(e.max()) / scale, rms_err_over_scale=float(e.pow(2).mean().sqrt()) / scale,
                             argmax_equal_f64=float((i.cpu() == y.argmax(dim=1)).double().mean()))
        res['range_flag'] = int(flag.item())
        out[tag] = res
    # timing, bench shape
    B, N = 32, 1024
    g = torch.Generator().manual_seed(1)
    h2 = torch.randn(B * N, 128, generator=g).relu().to(dev)
    Wt = (torch.randn(128, 1024, generator=g) * 0.1).to(dev)
    bias = torch.randn(1024, generator=g).to(dev)
    _p = bench._p
    mo, mi = torch.empty(B, 1024, device=dev), torch.empty(B, 1024, device=dev, dtype=torch.int64)
    tk = torch.zeros(4096, device=dev, dtype=torch.int32)
    W3 = ops.split_weights_bf16x3(Wt.t().contiguous())
    W2 = ops.split_weights_f16x2(Wt.t().contiguous())
    times = {}
    for blocks in (0, 128):
        n3 = lib.hitadv_linear_max_fwd_bf16x3_scratch(B, N, 1024, blocks)
        pv, pi = torch.empty(n3, device=dev), torch.empty(n3, device=dev, dtype=torch.int32)
        times['bf16x3 blocks=%d' % blocks] = round(bench.graph_timed(lambda st: lib.hitadv_linear_max_fwd_bf16x3(
            _p(h2), _p(W3), _p(bias), B, N, 128, 1024, 1, blocks, _p(pv), _p(pi), _p(mo), _p(mi), _p(tk), st)), 2)
        times['fp16x2 blocks=%d' % blocks] = round(bench.graph_timed(lambda st: lib.hitadv_linear_max_fwd_f16x2(
            _p(h2), _p(W2), _p(bias), B, N, 128, 1024, 1, blocks, _p(pv), _p(pi), _p(mo), _p(mi), _p(tk), None, st)), 2)
    n = lib.hitadv_linear_max_fwd_scratch(B, N, 1024)
    pv, pi = torch.empty(n, device=dev), torch.empty(n, device=dev, dtype=torch.int32)
    times['f32'] = round(bench.graph_timed(lambda st: lib.hitadv_linear_max_fwd(_p(h2), _p(Wt), _p(bias), B, N, 128, 1024, 1, _p(pv), _p(pi),
                                                                                _p(mo), _p(mi), _p(tk), st)), 2)
    out['us_per_launch'] = times
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
