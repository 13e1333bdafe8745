#!/usr/bin/env python3
"""Diagnostic: which piece of PCT's GPU fast path loses input-gradient accuracy.  Same sampling / grouping tables for every
run; one piece of the fast path at a time is replaced by its plain torch composition, and the input gradient is compared
with the float64 module on the CPU (relative L2 error)."""
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

    replay('cpu')
    xc = x.double().requires_grad_()
    lc = copy.deepcopy(m).double()(xc)
    (lc * w.double()).sum().backward()
    gd = xc.grad

    def run(model=gm, dtype=torch.float32):
        replay('cuda')
        xa = x.cuda().to(dtype).requires_grad_()
        la = model(xa)
        (la * w.cuda().to(dtype)).sum().backward()
        g = xa.grad.cpu().double()
        return dict(grad_rel_l2=float((g - gd).norm() / gd.norm()),
                    logits_rel=float((la.detach().cpu().double() - lc.detach()).abs().max() / lc.detach().abs().max()))

    out = {}
    out['fast'] = run()
    # (1) group_add_relu as a torch composition
    real_gar = ops.group_add_relu
    ops.group_add_relu = lambda U, V, idx: F.relu(PCT.index_points(U, idx) + V.unsqueeze(2))
    out['fast, group_add_relu as torch ops'] = run()
    ops.group_add_relu = real_gar
    # (2) lrelu_pool as torch ops
    real_sup = ops.lrelu_pool_supported
    ops.lrelu_pool_supported = lambda C: False
    out['fast, lrelu_pool as torch ops'] = run()
    ops.lrelu_pool_supported = real_sup
    # (3) linear + ReLU epilogue as torch ops
    real_lr = PCT.linear_relu_pm
    plain = lambda conv, bn, t: F.relu(_pointwise.linear_pm(conv, bn, t))  # noqa: E731
    PCT.linear_relu_pm = plain
    out['fast, linear_relu as torch ops'] = run()
    # (4) all three
    ops.group_add_relu = lambda U, V, idx: F.relu(PCT.index_points(U, idx) + V.unsqueeze(2))
    ops.lrelu_pool_supported = lambda C: False
    out['fast, all three as torch ops'] = run()
    ops.group_add_relu, ops.lrelu_pool_supported, PCT.linear_relu_pm = real_gar, real_sup, real_lr
    # (5) the split first layer replaced by gather / subtract / concat + GEMMs (Local_op.forward on the grouped tensor)
    real_fast = PCT.Local_op.fast
    real_from = PCT.Local_op.from_points

    def from_points_plain(self, xyz, points, npoint, nsample):
        new_xyz, grouped = PCT.sample_and_group(npoint, 0., nsample, xyz, points)
        return new_xyz, self.forward(grouped).permute(0, 2, 1)
    PCT.Local_op.from_points = from_points_plain
    out['fast, Local_op on the grouped tensor'] = run()
    PCT.Local_op.from_points = real_from
    # (5b) the offset-attention layers through their channels-major form (permute in, permute out)
    real_sa = PCT.SA_Layer.forward_pm
    PCT.SA_Layer.forward_pm = lambda self, t: self.forward(t.permute(0, 2, 1)).permute(0, 2, 1)
    out['fast, SA layers channels-major'] = run()
    PCT.SA_Layer.forward_pm = real_sa
    # (5c) single pieces of the attention layer
    def sa_variant(which):
        def f(self, x):
            q = _pointwise.linear_pm(self.q_conv, None, x)
            k = _pointwise.linear_pm(self.k_conv, None, x) if which == 'separate_qk' else q
            energy = torch.bmm(q, k.transpose(1, 2))
            attention = self.softmax(energy)
            attention = attention / (1e-9 + attention.sum(dim=1, keepdim=True))
            v = _pointwise.linear_pm(self.v_conv, None, x)
            if which == 'xr_cm':
                x_r = torch.bmm(v.transpose(1, 2), attention).transpose(1, 2)
            else:
                x_r = torch.bmm(attention.transpose(1, 2), v)
            return x + PCT.linear_relu_pm(self.trans_conv, self.after_norm, x - x_r)
        return f
    for which in ('same', 'separate_qk', 'xr_cm'):
        PCT.SA_Layer.forward_pm = sa_variant(which)
        out['fast, SA variant ' + which] = run()
    PCT.SA_Layer.forward_pm = real_sa
    # (5d) the whole transformer block channels-major
    real_pt = PCT.Point_Transformer_Last.forward_pm
    PCT.Point_Transformer_Last.forward_pm = lambda self, t: self.forward(t.permute(0, 2, 1)).permute(0, 2, 1)
    out['fast, pt_last channels-major'] = run()
    PCT.Point_Transformer_Last.forward_pm = real_pt
    # (5e) the two shared layers in front, and the fusion layer, through the channels-major matmul
    real_fpm = PCT.Pct._forward_points_major

    def variant(first_cm, fuse_cm):
        def f(self, x):
            xyz = x.permute(0, 2, 1).contiguous()
            if first_cm:
                h = F.relu(_pointwise.conv1x1(self.conv2, self.bn2, F.relu(_pointwise.conv1x1(self.conv1, self.bn1, x)))).permute(0, 2, 1)
            else:
                h = PCT.linear_relu_pm(self.conv2, self.bn2, PCT.linear_relu_pm(self.conv1, self.bn1, xyz))
            new_xyz, p0 = self.gather_local_0.from_points(xyz, h, 512, 32)
            new_xyz, p1 = self.gather_local_1.from_points(new_xyz, p0, 256, 32)
            cat = torch.cat([self.pt_last.forward_pm(p1), p1], dim=2)
            if fuse_cm:
                z = _pointwise.conv1x1(self.conv_fuse[0], self.conv_fuse[1], cat.permute(0, 2, 1))
                g = F.leaky_relu(z, negative_slope=0.2).max(dim=2)[0]
            else:
                z = _pointwise.linear_pm(self.conv_fuse[0], self.conv_fuse[1], cat)
                g = F.leaky_relu(z, negative_slope=0.2).max(dim=1)[0]
            g = self.dp1(F.leaky_relu(self.bn6(self.linear1(g)), negative_slope=0.2))
            g = self.dp2(F.leaky_relu(self.bn7(self.linear2(g)), negative_slope=0.2))
            return self.linear3(g)
        return f
    for a, b in ((True, False), (False, True), (True, True)):
        PCT.Pct._forward_points_major = variant(a, b)
        out['fast, first layers channels-major=%s, fusion channels-major=%s' % (a, b)] = run()
    PCT.Pct._forward_points_major = real_fpm
    # (5f) the input-gradient GEMM of the first layer alone: [B*N,64] @ [64,3]
    g64 = torch.randn(2048, 64, device='cuda')
    W = torch.randn(64, 3, device='cuda')
    ref = (g64.double() @ W.double())
    out['gemm [2048,64]@[64,3] fp32 rel err'] = float(((g64 @ W).double() - ref).norm() / ref.norm())
    out['gemm W^T[3,64]@[64,2048] fp32 rel err'] = float(((W.t() @ g64.t()).double() - ref.t()).norm() / ref.norm())
    # (6) the plain module in fp32, and the fast path in float64 on the GPU (separates formulation from precision)
    fast = _pointwise._fast
    _pointwise._fast = lambda conv, bn, t: False
    out['plain module fp32'] = run()
    _pointwise._fast = fast
    ops.lrelu_pool_supported = lambda C: False
    ops.group_add_relu = lambda U, V, idx: F.relu(PCT.index_points(U, idx) + V.unsqueeze(2))
    PCT.linear_relu_pm = plain
    out['fast formulation (torch ops) in float64 on the GPU'] = run(copy.deepcopy(m).double().cuda(), torch.float64)
    ops.group_add_relu, ops.lrelu_pool_supported, PCT.linear_relu_pm = real_gar, real_sup, real_lr
    for n in log:
        setattr(PCT, n, saved[n])
    PCT.Local_op.fast = real_fast
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
