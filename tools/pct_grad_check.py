#!/usr/bin/env python3
"""Diagnostic: PCT's input gradient through three formulations on the SAME sampling / grouping tables --
(a) the GPU fast path (points-major GEMMs, hitadv_group_add_relu, hitadv_lrelu_pool), (b) the plain nn.Module on the GPU
(fp32), (c) the plain nn.Module in float64 on the CPU -- and how far apart they are."""
import argparse
import copy
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hit_adv_amd.Dataset.synthetic import synth_batch  # noqa: E402
from hit_adv_amd.model import _pointwise, _sampling  # noqa: E402
from hit_adv_amd.model import pct as PCT  # noqa: E402


def main():
    torch.manual_seed(29)
    m = PCT.Pct(argparse.Namespace(dropout=0.2), output_channels=40).eval()
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.normal_(0, 0.1)
                mod.running_var.uniform_(0.7, 1.3)
    data, _ = synth_batch(2, 1024, first=12000)
    x = data[:, :, :3].transpose(1, 2).contiguous()
    w = torch.randn(2, 40, generator=torch.Generator().manual_seed(4))
    gm = copy.deepcopy(m).cuda()
    torch.manual_seed(31)
    feed = _sampling.feed_for(gm, 2, 1024, 1, 'cuda')
    log = {'fps': [], 'knn_point': []}
    saved = {n: getattr(PCT, n) for n in log}
    for n in log:
        setattr(PCT, n, (lambda n: lambda *a, **k: (log[n].append(saved[n](*a, **k)), log[n][-1])[1])(n))
    xa = x.cuda().requires_grad_()
    with _sampling.using(feed):
        la = gm(xa)
    (la * w.cuda()).sum().backward()
    for n in log:
        setattr(PCT, n, saved[n])

    def replay(device):
        its = {n: iter([t.to(device) for t in rows]) for n, rows in log.items()}
        for n in log:
            setattr(PCT, n, (lambda n: lambda *a, **k: next(its[n]))(n))

    fast = _pointwise._fast
    _pointwise._fast = lambda conv, bn, x: False  # the modules themselves (MIOpen / plain ops), fp32, GPU
    PCT.fast_pm = _pointwise.fast_pm
    replay('cuda')
    xb = x.cuda().requires_grad_()
    lb = gm(xb)
    (lb * w.cuda()).sum().backward()
    _pointwise._fast = fast
    replay('cpu')
    xc = x.double().requires_grad_()
    lc = copy.deepcopy(m).double()(xc)
    (lc * w.double()).sum().backward()
    for n in log:
        setattr(PCT, n, saved[n])

    def cmp(a, b):
        a, b = a.detach().cpu().double(), b.detach().cpu().double()
        scale = float(b.abs().max())
        bad = ((a - b).abs() > 1e-3 * b.abs() + 1e-5 * scale).double().mean().item()
        return dict(rel_l2=float((a - b).norm() / b.norm()), frac_off=bad, max_abs_over_scale=float((a - b).abs().max() / scale))
    out = dict(logits_fast_vs_f64=cmp(la, lc), logits_module_vs_f64=cmp(lb, lc), grad_fast_vs_f64=cmp(xa.grad, xc.grad),
               grad_module_vs_f64=cmp(xb.grad, xc.grad), grad_fast_vs_module=cmp(xa.grad, xb.grad),
               grad_norm=float(xc.grad.norm()), grad_absmax=float(xc.grad.abs().max()))
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
