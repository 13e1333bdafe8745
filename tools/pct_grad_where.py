#!/usr/bin/env python3
"""Diagnostic: WHERE PCT's fp32 fast-path input gradient differs from the float64 module's (same tables)."""
import argparse
import copy
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hit_adv_amd import ops  # noqa: E402
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
    with _sampling.using(feed), torch.no_grad():
        gm(x.cuda())
    for n in log:
        setattr(PCT, n, saved[n])

    def replay(device):
        its = {n: iter([t.to(device) for t in rows]) for n, rows in log.items()}
        for n in log:
            setattr(PCT, n, (lambda n: lambda *a, **k: next(its[n]))(n))

    # record the intermediate tensors that feed a max: hook torch.Tensor.max? simpler: capture via module-level wrappers
    caps = {}

    def capture_run(model, xin, tag, plain):
        fast = _pointwise._fast
        if plain:
            _pointwise._fast = lambda conv, bn, t: False
        real_sup = ops.lrelu_pool_supported
        ops.lrelu_pool_supported = lambda C: False
        acts = []
        real_max = torch.Tensor.max

        def spy(self, *a, **k):
            out = real_max(self, *a, **k)
            if a or k:
                acts.append((tuple(self.shape), a, out[1].detach().cpu(), self.detach().cpu().double() if self.numel() < 1 << 20 else None))
            return out
        torch.Tensor.max = spy
        real_amp = F.adaptive_max_pool1d

        def spy_amp(t, o, *a, **k):
            acts.append((tuple(t.shape), 'amp', t.detach().argmax(dim=2).cpu(), t.detach().cpu().double() if t.numel() < 1 << 20 else None))
            return real_amp(t, o, *a, **k)
        F.adaptive_max_pool1d = spy_amp
        try:
            la = model(xin)
        finally:
            torch.Tensor.max = real_max
            F.adaptive_max_pool1d = real_amp
            ops.lrelu_pool_supported = real_sup
            _pointwise._fast = fast
        caps[tag] = acts
        return la

    replay('cpu')
    xc = x.double().requires_grad_()
    lc = capture_run(copy.deepcopy(m).double(), xc, 'f64', True)
    (lc * w.double()).sum().backward()
    gd = xc.grad
    replay('cuda')
    xa = x.cuda().requires_grad_()
    la = capture_run(gm, xa, 'fast', False)
    (la * w.cuda()).sum().backward()
    replay('cuda')
    xb = x.cuda().requires_grad_()
    lb = capture_run(gm, xb, 'plain', True)
    (lb * w.cuda()).sum().backward()
    for n in log:
        setattr(PCT, n, saved[n])
    out = {}
    for tag, g in (('fast', xa.grad), ('plain', xb.grad)):
        d = (g.cpu().double() - gd)
        per_point = d.pow(2).sum(1)  # [B,N]
        tot = per_point.sum()
        top = torch.topk(per_point.flatten(), 10)
        out[tag] = dict(rel_l2=float(d.norm() / gd.norm()), per_cloud=[float(per_point[b].sum() / tot) for b in range(2)],
                        top10_share=float(top.values.sum() / tot), top10_points=[(int(i) // 1024, int(i) % 1024) for i in top.indices],
                        points_for_90pct=int((torch.cumsum(torch.sort(per_point.flatten(), descending=True).values, 0) < 0.9 * tot).sum()) + 1)
    # arg-max tables of every max in the three runs
    out['max_sites'] = {tag: [(list(s), str(a)[:30]) for s, a, _, _ in acts] for tag, acts in caps.items()}
    def winners(tag, i):
        return caps[tag][i][2]
    try:
        n_sites = len(caps['f64'])
        out['winner_mismatch_vs_f64'] = {}
        for tag in ('fast', 'plain'):
            rows = []
            for i in range(min(n_sites, len(caps[tag]))):
                a, b = winners('f64', i), winners(tag, i)
                if a.shape != b.shape and a.dim() == b.dim() and a.dim() >= 2:
                    b = b.transpose(-1, -2) if b.transpose(-1, -2).shape == a.shape else b
                rows.append(float((a != b).float().mean()) if a.shape == b.shape else 'shape %s vs %s' % (list(a.shape), list(b.shape)))
            out['winner_mismatch_vs_f64'][tag] = rows
    except Exception as e:  # noqa: BLE001
        out['winner_error'] = repr(e)
    zf, zd, zp = caps['fast'][2][3], caps['f64'][2][3].transpose(1, 2), caps['plain'][2][3].transpose(1, 2)  # [B,256,1024]
    wf, wd = caps['fast'][2][2], caps['f64'][2][2]
    rows = []
    for b, c in (wf != wd).nonzero().tolist():
        i_d, i_f = int(wd[b, c]), int(wf[b, c])
        rows.append(dict(b=b, c=c, f64_winner=i_d, fast_winner=i_f,
                         f64_vals=[float(zd[b, i_d, c]), float(zd[b, i_f, c])], fast_vals=[float(zf[b, i_d, c]), float(zf[b, i_f, c])],
                         plain_vals=[float(zp[b, i_d, c]), float(zp[b, i_f, c])]))
    out['final_pool_flips'] = rows
    out['z_err_fast_vs_f64'] = float((zf - zd).abs().max() / zd.abs().max())
    out['z_err_plain_vs_f64'] = float((zp - zd).abs().max() / zd.abs().max())
    out['z_rowdiff'] = dict(fast=float((zf - zd).norm() / zd.norm()), plain=float((zp - zd).norm() / zd.norm()))
    del out['max_sites']
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
