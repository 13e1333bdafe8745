"""BASELINE.json configs 3, 4 and 5 at workload size on the GPU, against the CPU oracle.

cfg3  ModelNet40-shaped 1024-point clouds, DGCNN victim (k = 5, eval.py:48), one rank's shard of 32 clouds
cfg4  ShapeNetPart-shaped 2048-point clouds, batch 64, PointNet++ SSG victim
cfg5  1024-point clouds, batch 32, PCT victim under the AdvPC + kNN + AOF sweep (CW/AdvPC.py, CW/kNN.py, CW/AOF.py)

The oracle (oracle/hitadv_oracle.py) drives the victims' plain nn.Module on the CPU with the reference's own sampling /
grouping arithmetic (oracle/victim_geometry.py; pinned to fixtures g10-g12 in tests/test_dgcnn.py and
tests/test_victims_cpu.py).  Short runs: the trajectories are chaotic over hundreds of Adam steps, not over a handful.
Every tolerance is what the comparison needs, stated where it is asserted; what was ACHIEVED is recorded by
helpers.close() into gpurun_out/parity_report_gpu.json.
"""
import argparse
import contextlib
import copy
import io
import warnings

import numpy as np
import pytest
import torch

from helpers import close, gradient_close, note, synth_batch
from oracle import c_oracle as N
from oracle import hitadv_oracle as O
from oracle import victim_geometry as VG

pytestmark = pytest.mark.gpu

HP = dict(attack_lr=1e-2, init_weight=10., max_weight=80., cd_weight=1e-4, ker_weight=1., hide_weight=1.,
          curv_loss_knn=16, central_num=192, total_central_num=256, max_sigm=1.2, min_sigm=0.1, budget=0.55)


def _shake_bn(model, spread=0.1):
    with torch.no_grad():
        for mod in model.modules():
            if isinstance(mod, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                mod.running_mean.normal_(0, spread)
                mod.running_var.uniform_(0.7, 1.3)
    return model


def _labels(model, data):
    with torch.no_grad():
        out = model(data[:, :, :3].transpose(1, 2).contiguous())
    return (out[0] if isinstance(out, tuple) else out).argmax(1)


class _Iterates:
    """Wraps an attacker's iteration: the deformed clouds and predictions of every pass (eager loop)."""

    def __init__(self, att):
        self.rows, self.inner = [], att._iteration
        att._iteration = self

    def __call__(self, ws):
        self.inner(ws)
        torch.cuda.synchronize()
        self.rows.append(dict(adv=ws.adv.cpu().numpy().copy(), pred=ws.state['pred'].cpu().numpy().copy()))


def test_cfg3_dgcnn_batch32_hit_adv_vs_cpu_oracle():
    """One rank's shard of cfg3: B = 32, N = 1024, DGCNN(k=5).  Same 192 centres per cloud as the oracle, iterates of a
    five-iteration run against the oracle's, graph run == eager run."""
    from hit_adv_amd.ShapeAttack.HiT_ADV import HiT_ADV
    from hit_adv_amd.model.dgcnn import DGCNN_cls
    from hit_adv_amd.util.adv_utils import UntargetedLogitsAdvLoss
    torch.manual_seed(11)
    cpu_model = _shake_bn(DGCNN_cls(argparse.Namespace(k=5, emb_dims=1024, dropout=0.2), output_channels=40).eval())
    data, _ = synth_batch(32, 1024, first=8000)
    label = _labels(cpu_model, data)
    hp = dict(binary_step=1, num_iter=5, **HP)
    trace = []
    oracle = O.HiTADVOracle(cpu_model, lambda l, t: O.untargeted_logits_adv_loss(l, t, 30.), **hp)
    torch.manual_seed(3)
    with contextlib.redirect_stdout(io.StringIO()):
        obest, osucc = oracle.attack(data, label, trace=trace)

    att = HiT_ADV(copy.deepcopy(cpu_model), UntargetedLogitsAdvLoss(30.), verbose=False, use_graph=False, **hp)
    rec = _Iterates(att)
    torch.manual_seed(3)
    best, succ = att.attack(data, label)
    assert att._view is not None  # the folded EdgeConv view (csrc: knn_features, edge_max, lrelu_pool) is what ran
    ws = next(iter(att._ws.values()))
    same = (ws.central.cpu() == oracle.state['central']).all(dim=1).float().mean().item()
    # the centre ranking uses 0.001 * normalised saliency: a victim gradient that differs in its last bits (the view
    # re-associates EdgeConv; feature-space neighbour near-ties) may swap two all-but-tied candidates
    assert same >= 0.995, same
    # achieved on MI355X (profiles/r02_parity_report.json): |gpu - oracle| <= 2.1e-6 on clouds of unit scale over the five
    # iterations -- the folded view, the MFMA feature-space kNN and the HIP EdgeConv kernels against the plain module
    for i, row in enumerate(rec.rows):
        close(row['adv'], trace[i]['adv'], rtol=1e-4, atol=2e-5, what='cfg3 iterate %d' % i)
        assert (row['pred'] == trace[i]['pred']).all()
    close(best, obest, rtol=1e-4, atol=2e-5, what='cfg3 result')
    att = HiT_ADV(copy.deepcopy(cpu_model), UntargetedLogitsAdvLoss(30.), verbose=False, use_graph=True, **hp)
    torch.manual_seed(3)
    gbest, gsucc = att.attack(data, label)
    assert att.last_graph_used and np.array_equal(gbest, best) and int(gsucc) == int(succ)  # no float atomics left


def _record_tables(mod, names):
    """Record the index tables the named geometry functions of ``mod`` return (GPU run), for replay in float64."""
    log = {n: [] for n in names}
    saved = {n: getattr(mod, n) for n in names}

    def wrap(n):
        def f(*a, **k):
            out = saved[n](*a, **k)
            log[n].append(out.detach().cpu())
            return out
        return f
    for n in names:
        setattr(mod, n, wrap(n))
    return log, saved


def _replay_tables(mod, log):
    saved = {n: getattr(mod, n) for n in log}
    its = {n: iter(rows) for n, rows in log.items()}
    for n in log:
        setattr(mod, n, (lambda n: lambda *a, **k: next(its[n]))(n))
    return saved


def _restore(mod, saved):
    for n, f in saved.items():
        setattr(mod, n, f)


def test_cfg4_pointnet2_batch64_2048_points_tables_and_float64_module():
    """cfg4's forward / input gradient at B = 64, N = 2048: FPS tables and ball-query tables bit-exact against the C oracle
    (the reference's Gram-form rule, pinned to torch and fixture g11 on the CPU) on the same start indices, logits and input
    gradient against the float64 module evaluated on the SAME tables."""
    from hit_adv_amd.model import _sampling
    from hit_adv_amd.model import pointnet2 as P2
    torch.manual_seed(13)
    m = _shake_bn(P2.get_model(16, normal_channel=False).eval())  # ShapeNetPart: 16 object categories
    data, _ = synth_batch(64, 2048, first=9000)
    x = data[:, :, :3].transpose(1, 2).contiguous()
    gm = copy.deepcopy(m).cuda()
    torch.manual_seed(17)
    feed = _sampling.feed_for(gm, 64, 2048, 1, 'cuda')
    log, saved = _record_tables(P2, ['farthest_point_sample', 'query_ball_point'])
    try:
        xg = x.cuda().requires_grad_()
        with _sampling.using(feed):
            logits, _ = gm(xg)
        w = torch.randn(64, 16, generator=torch.Generator().manual_seed(4))
        (logits * w.cuda()).sum().backward()
    finally:
        _restore(P2, saved)
    fps1, fps2 = log['farthest_point_sample']
    ball1, ball2 = log['query_ball_point']
    starts = feed.table.cpu()[0]  # [2, B]
    pts = x.transpose(1, 2).contiguous()
    assert torch.equal(fps1, N.fps_from_start(pts, 512, starts[0]))
    l1_xyz = VG.index_points(pts, fps1)
    assert torch.equal(fps2, N.fps_from_start(l1_xyz, 128, starts[1]))
    assert torch.equal(ball1, VG.c_query_ball_point(0.2, 32, pts, l1_xyz))
    assert torch.equal(ball2, VG.c_query_ball_point(0.4, 64, l1_xyz, VG.index_points(l1_xyz, fps2)))
    # float64 module on the SAME tables (clouds are independent: the first 8 of the 64 keep the CPU time in seconds)
    nb = 8
    md = copy.deepcopy(m).double()
    saved = _replay_tables(P2, {k: [t[:nb] for t in v] for k, v in log.items()})
    try:
        xd = x[:nb].double().requires_grad_()
        ld, _ = md(xd)
        (ld * w[:nb].double()).sum().backward()
    finally:
        _restore(P2, saved)
    close(logits[:nb], ld, rtol=1e-4, atol=1e-5, what='cfg4 logits vs float64 module')
    _gradient_vs_float64(xg.grad[:nb], xd.grad, 'cfg4 input gradient')


def _gradient_vs_float64(g, gd, what, frac_bound=2e-3, l2_bound=3e-2):
    gradient_close(g, gd, what, frac_bound, l2_bound)


def test_dgcnn_gradient_vs_float64_module_on_the_same_graphs():
    """DGCNN (module path: HIP kNN in 3-D, MFMA scores + HIP top-k in feature space) at B = 4, N = 1024: logits and
    input gradient against the float64 module evaluated on the SAME four neighbour tables."""
    from hit_adv_amd.model import dgcnn as DG
    torch.manual_seed(23)
    m = _shake_bn(DG.DGCNN_cls(argparse.Namespace(k=5, emb_dims=1024, dropout=0.2), output_channels=40).eval())
    data, _ = synth_batch(4, 1024, first=11000)
    x = data[:, :, :3].transpose(1, 2).contiguous()
    w = torch.randn(4, 40, generator=torch.Generator().manual_seed(4))
    log, saved = _record_tables(DG, ['knn'])
    try:
        xg = x.cuda().requires_grad_()
        logits = copy.deepcopy(m).cuda()(xg)
        (logits * w.cuda()).sum().backward()
    finally:
        _restore(DG, saved)
    assert len(log['knn']) == 4
    saved = _replay_tables(DG, log)
    try:
        xd = x.double().requires_grad_()
        ld = copy.deepcopy(m).double()(xd)
        (ld * w.double()).sum().backward()
    finally:
        _restore(DG, saved)
    close(logits, ld, rtol=1e-4, atol=1e-5, what='DGCNN logits vs float64 module (same graphs)')
    _gradient_vs_float64(xg.grad, xd.grad, 'DGCNN input gradient')


@contextlib.contextmanager
def _pct_pool_winners(winners, pre_pool=None):
    """While active, every max-pool of a PCT pass on the GPU fast path appends its arg-max table to ``winners``, in the
    order the passes run them: Local_op 1 [B,512,128], Local_op 2 [B,256,256], the final pool over the points [B,1024]."""
    from hit_adv_amd import ops
    real_max, real_pool, real_group, real_g16 = torch.Tensor.max, ops.lrelu_pool, ops.group_linear_max, ops.group_linear_max_g16
    real_fused = ops.linear_lrelu_pool

    def spy_g16(xx, Wp, Wtp, bias, flag=None, return_arg=False, **kw):  # the second Local_op's last layer: the tiled GEMM core
        out, arg = real_g16(xx, Wp, Wtp, bias, flag, return_arg=True, **kw)
        winners.append(arg.detach().cpu().long())
        return (out, arg) if return_arg else out

    def spy_group(xx, Wr, bias, flag=None, return_arg=False, **kw):  # Local_op's last layer + max over the neighbours, fused
        out, arg = real_group(xx, Wr, bias, flag, return_arg=True, **kw)
        winners.append(arg.detach().cpu().long())
        return (out, arg) if return_arg else out

    def spy_max(self, *a, **k):  # the max over the neighbours of Local_op.from_points where the fused layer does not apply
        out = real_max(self, *a, **k)
        if (a or k) and self.dim() == 4:  # [B,S,nsample,C]; the attacks' own .max calls (logits, losses) are 1-D / 2-D
            winners.append(out[1].detach().cpu())
        return out

    def spy_pool(Z, slope=0.2):
        out, arg = real_pool(Z, slope, return_arg=True)
        winners.append(arg.detach().cpu().long())
        if pre_pool is not None:
            pre_pool.append(Z.detach().cpu())
        return out

    def spy_fused(xx, Wp, Wtp, bias, Bn, npts, slope=0.2, flag=None, return_arg=False):  # conv_fuse + LeakyReLU + max pool, fused
        out, arg = real_fused(xx, Wp, Wtp, bias, Bn, npts, slope, flag, return_arg=True)
        winners.append(arg.detach().cpu().long())
        if pre_pool is not None:
            pieces = Wp.view(torch.float16).float()  # the layer's pre-activation never exists on this path: rebuilt here from the
            W = pieces[0] + pieces[1] / 2048.0       # operands the kernel was given (fp32 GEMM: the same values to fp32 rounding)
            pre_pool.append((xx.detach() @ W.t() + bias).view(Bn, npts, -1).cpu())
        return (out, arg) if return_arg else out
    try:
        torch.Tensor.max, ops.lrelu_pool, ops.group_linear_max, ops.group_linear_max_g16 = spy_max, spy_pool, spy_group, spy_g16
        ops.linear_lrelu_pool = spy_fused
        yield winners
    finally:
        torch.Tensor.max, ops.lrelu_pool, ops.group_linear_max, ops.group_linear_max_g16 = real_max, real_pool, real_group, real_g16
        ops.linear_lrelu_pool = real_fused


@contextlib.contextmanager
def _imposed_pool_winners(winners, own=None, z=None):
    """While active, the plain PCT module's ``F.adaptive_max_pool1d`` calls take the recorded winners, in order, instead of
    their own arg-max (a no-op once the recording is used up): the module's max-pools are evaluated where the GPU pass
    evaluated them, as its sampling tables already are."""
    import torch.nn.functional as F
    queue = iter(winners)
    real_amp = F.adaptive_max_pool1d

    def amp(t, o):
        if own is not None:
            own.append(t.detach().argmax(dim=2))
        if z is not None:
            z.append(t.detach())
        idx = next(queue, None)                                  # [B,S,C] (neighbour maxima) or [B,C] (final pool)
        if idx is None:
            return real_amp(t, o)
        return t.gather(2, idx.reshape(t.shape[0], t.shape[1], 1))
    try:
        F.adaptive_max_pool1d = amp
        yield
    finally:
        F.adaptive_max_pool1d = real_amp


def test_pct_gradient_vs_float64_module_on_the_same_tables():
    """PCT at B = 2, N = 1024 on the SAME FPS and kNN grouping tables: (a) the GPU fast path (points-major GEMMs,
    hitadv_group_add_relu, the fused last layers and pooled embedding layer), (b) the plain nn.Module in fp32 on the GPU, (c) the plain nn.Module in
    float64 on the CPU.

    What round 2 read as "the fast path loses gradient accuracy" (tools/pct_grad_bisect.py, tools/pct_grad_where.py;
    gpurun_out/r03_pct_*.json): PCT's offset attention makes the rows of the fused feature map nearly equal, so the final
    max over the points has many near-ties -- in this input six of the 2048 (cloud, channel) maxima have a runner-up within
    5e-8 ... 1e-6 (relative) of the winner in float64.  ANY fp32 evaluation carries ~5e-7 of rounding there (fast path
    4.4e-7, plain module 3.3e-7, relative L2 of the pre-pool activation against float64); which candidate wins such a
    channel is decided by that rounding, and each flipped winner re-routes a whole channel's gradient (8 % of the gradient's
    L2 norm for six flips; the plain fp32 module happened to flip none here, on other inputs it flips too: 4.3 % in round
    2).  The fast path's formulation evaluated in float64 reproduces the float64 module to 1e-15.  So the gradient is
    compared where it is a function: (1) the pre-pool activations agree with float64 to fp32 rounding, (2) every winner that
    differs IS a near-tie within that rounding, (3) with the GPU run's winners imposed on the float64 module (as the
    sampling tables already are) the input gradients agree to the bounds every other victim is held to."""
    import torch.nn.functional as F
    from hit_adv_amd import ops
    from hit_adv_amd.model import _pointwise, _sampling
    from hit_adv_amd.model import pct as PCT
    torch.manual_seed(29)
    m = _shake_bn(PCT.Pct(argparse.Namespace(dropout=0.2), output_channels=40).eval())
    data, _ = synth_batch(2, 1024, first=12000)
    x = data[:, :, :3].transpose(1, 2).contiguous()
    w = torch.randn(2, 40, generator=torch.Generator().manual_seed(4))
    gm = copy.deepcopy(m).cuda()
    torch.manual_seed(31)
    feed = _sampling.feed_for(gm, 2, 1024, 1, 'cuda')
    log, saved = _record_tables(PCT, ['fps', 'knn_point'])
    winners, pre_pool = [], []
    try:
        with _pct_pool_winners(winners, pre_pool):
            xg = x.cuda().requires_grad_()
            with _sampling.using(feed):
                logits = gm(xg)
            (logits * w.cuda()).sum().backward()
    finally:
        _restore(PCT, saved)
    assert len(log['fps']) == 2 and len(log['knn_point']) == 2
    assert [tuple(t.shape) for t in winners] == [(2, 512, 128), (2, 256, 256), (2, 1024)] and len(pre_pool) == 1
    fast = _pointwise._fast
    saved = _replay_tables(PCT, {k: [t.cuda() for t in v] for k, v in log.items()})
    try:
        _pointwise._fast = lambda conv, bn, t: False  # the nn.Conv1d / BatchNorm1d modules themselves, fp32, GPU
        xm = x.cuda().requires_grad_()
        lm = gm(xm)
        (lm * w.cuda()).sum().backward()
    finally:
        _pointwise._fast = fast
        _restore(PCT, saved)

    def float64_run(impose):
        """The plain module in float64 on the CPU, same tables; ``impose``: its three max-pools take the GPU run's winners."""
        z64, own = [], []
        saved = _replay_tables(PCT, log)
        try:
            with _imposed_pool_winners(winners if impose else [], own, z64):
                xd = x.double().requires_grad_()
                ld = copy.deepcopy(m).double()(xd)
                (ld * w.double()).sum().backward()
        finally:
            _restore(PCT, saved)
        return ld.detach(), xd.grad, z64, own
    ld, gd, z64, own = float64_run(False)
    close(logits, ld, rtol=1e-4, atol=1e-5, what='PCT logits vs float64 module (same tables)')
    close(lm, ld, rtol=2e-4, atol=1e-5, what='PCT fp32 module logits vs float64 module (same tables)')
    # (1) the activation in front of the final pool, [B,256,1024] points-major on the GPU, [B,1024,256] in the module
    zd = z64[2].transpose(1, 2)
    zf = F.leaky_relu(pre_pool[0].double(), negative_slope=0.2)  # the module pools the activated map; the kernel activates inside
    close(float((zf - zd).norm() / zd.norm()), 0., rtol=0, atol=2e-6, what='PCT pre-pool activation, fast path: relative L2 error vs float64')
    close(float((zf - zd).abs().max() / zd.abs().max()), 0., rtol=0, atol=6e-6,
          what='PCT pre-pool activation, fast path: max error over scale vs float64')
    # (2) winners: the GPU's differ from float64's only where float64's own margin is inside fp32 rounding
    flips = (winners[2] != own[2]).nonzero().tolist()
    assert len(flips) <= 0.01 * winners[2].numel(), len(flips)
    for b, c in flips:
        mine, theirs = zd[b, winners[2][b, c], c], zd[b, own[2][b, c], c]
        assert float(theirs - mine) <= 8e-6 * float(zd[b, :, c].abs().max()), (b, c, float(mine), float(theirs))
    # (a channel whose ReLU'd maximum is 0 has no winner to speak of: the module's arg-max of a row of zeros is row 0, the fused
    # layer reports the row of the largest pre-activation; neither passes any gradient)
    neighbour_flips = [float(((winners[i].reshape(own[i].shape) != own[i]) & (z64[i].max(dim=2).values > 0)).double().mean())
                       for i in (0, 1)]
    assert max(neighbour_flips) <= 1e-3, neighbour_flips
    # (3) the gradient with the winners imposed
    ld2, gd2, _, _ = float64_run(True)
    close(logits, ld2, rtol=1e-4, atol=1e-5, what='PCT logits vs float64 module (same tables, same winners)')
    gradient_close(xg.grad, gd2, 'PCT input gradient, fast path vs float64 on the same tables and max-pool winners',
                   frac_bound=3e-2, l2_bound=5e-4)  # achieved 7e-3 / 1.04e-4 (8.2e-2 with the six flipped winners left in)
    # for the record: the raw comparison (flipped winners included), fast path and plain fp32 module
    l2_fast = float((xg.grad.cpu().double() - gd).norm() / gd.norm())
    l2_mod = float((xm.grad.cpu().double() - gd).norm() / gd.norm())
    close(l2_fast, 0., rtol=0, atol=0.2, what='PCT input gradient, fast path: raw relative L2 error vs float64 (%d flipped winners)' % len(flips))
    close(l2_mod, 0., rtol=0, atol=0.2, what='PCT input gradient, plain fp32 module: raw relative L2 error vs float64')


def test_cfg4_pointnet2_batch64_hit_adv_vs_cpu_oracle_and_graph():
    """HiT-ADV on cfg4 (B = 64, N = 2048, PointNet++ SSG): two iterations against the CPU oracle (same random draws in
    the reference's order: the victim's FPS starts come from the attack's pre-drawn feed), and the loop is captured."""
    from hit_adv_amd.ShapeAttack.HiT_ADV import HiT_ADV
    from hit_adv_amd.model import pointnet2 as P2
    from hit_adv_amd.util.adv_utils import UntargetedLogitsAdvLoss
    torch.manual_seed(13)
    cpu_model = _shake_bn(P2.get_model(16, normal_channel=False).eval())
    data, _ = synth_batch(64, 2048, first=9000)
    victim = VG.CpuVictim(cpu_model)
    torch.manual_seed(2)
    label = _labels(victim, data)
    hp = dict(binary_step=1, num_iter=2, **HP)
    trace = []
    oracle = O.HiTADVOracle(victim, lambda l, t: O.untargeted_logits_adv_loss(l, t, 30.), **hp)
    torch.manual_seed(7)
    with contextlib.redirect_stdout(io.StringIO()):
        obest, osucc = oracle.attack(data, label, trace=trace)
    att = HiT_ADV(copy.deepcopy(cpu_model), UntargetedLogitsAdvLoss(30.), verbose=False, use_graph=False, **hp)
    rec = _Iterates(att)
    torch.manual_seed(7)
    best, succ = att.attack(data, label)
    ws = next(iter(att._ws.values()))
    assert ws.feed is not None and ws.feed.table.shape == (2, 2, 64)
    same = (ws.central.cpu() == oracle.state['central']).all(dim=1).float().mean().item()
    assert same >= 0.995, same
    # achieved on MI355X: |gpu - oracle| <= 1.1e-6 (the victim's FPS and ball-query tables are the oracle's bit for bit: both
    # evaluate the reference's own Gram-form expressions, test_cfg4_pointnet2_batch64_2048_points_tables_and_float64_module)
    for i, row in enumerate(rec.rows):
        close(row['adv'], trace[i]['adv'], rtol=1e-4, atol=2e-5, what='cfg4 iterate %d' % i)
    close(best, obest, rtol=1e-4, atol=2e-5, what='cfg4 result')
    att = HiT_ADV(copy.deepcopy(cpu_model), UntargetedLogitsAdvLoss(30.), verbose=False, use_graph='always',
                  binary_step=1, num_iter=4, **HP)
    with warnings.catch_warnings():
        warnings.simplefilter('error')  # "not capturable, running the eager loop" would be a failure here
        torch.manual_seed(7)
        gbest, _ = att.attack(data, label)
    assert att.last_graph_used and np.isfinite(gbest).all()


class _ToyAE(torch.nn.Module):
    """Point-wise auto-encoder stand-in (the reference ships no auto-encoder for AdvPC): [B,3,K] -> [B,3,K]."""

    def __init__(self):
        super().__init__()
        self.enc = torch.nn.Conv1d(3, 16, 1)
        self.dec = torch.nn.Conv1d(16, 3, 1)

    def forward(self, x):
        return x + 0.05 * self.dec(torch.tanh(self.enc(x)))


def _pct():
    from hit_adv_amd.model import pct as PCT
    torch.manual_seed(19)
    return _shake_bn(PCT.Pct(argparse.Namespace(dropout=0.2), output_channels=40).eval())


def _direct_chamfer_knn(adv, ori):
    """ChamferkNNDist in the DIRECT form ((dx^2 + dy^2) + dz^2), the form the product evaluates by default.  Round 5 tried the
    reference's Gram form on both sides here (``ChamferkNNDist(reference_arithmetic=True)`` against ``O.chamfer_knn_dist``,
    VERDICT r04 #3): the 99th percentile of |gpu - oracle| went from 2e-6 to 2e-3 in the FIRST iterate.  The Gram form's
    squared distances of near neighbours (1e-4) are differences of numbers of order 1: which of two near-tied neighbours
    ranks fifth is decided by the last bit of a 3-term dot product, and torch's CPU ``bmm`` on the GPU box's host does not
    sum it the way the build host's does (the kernel reproduces the latter bit for bit: tests/test_oracle_gram.py, which
    runs where the fixtures were made).  The Gram form is pinned where it can be: fixture g7 / g24 (the reference's own
    outputs), the C restatement against torch on the build host, the kernel against the C restatement."""
    P = O.pairwise_sqdist_direct(ori, adv)
    cham = P.min(dim=1).values.mean(dim=1)
    S = torch.sort(O.pairwise_sqdist_direct(adv, adv), dim=-1, stable=True).values[..., 1:6].mean(-1)
    with torch.no_grad():
        mask = (S > (S.mean(-1) + 1.05 * S.std(-1))[:, None]).float()
    return cham * 5. + (S * mask).mean(1) * 3.


@pytest.mark.parametrize("which", ["advpc", "knn", "aof"])
def test_cfg5_pct_batch32_cw_sweep_vs_cpu_oracle(which):
    """cfg5 at B = 32, N = 1024 with the real PCT victim: each attack of the sweep for a few iterations against its oracle
    restatement (O.cw_family_attack / O.cw_knn_attack / O.cw_aof_attack) driving the plain PCT module on the CPU."""
    from hit_adv_amd import CW
    from hit_adv_amd.util.adv_utils import LogitsAdvLoss, UntargetedLogitsAdvLoss
    from hit_adv_amd.util.clip_utils import ClipPointsLinf
    from hit_adv_amd.util.dist_utils import ChamferkNNDist, L2Dist
    cpu_model = _pct()
    victim = VG.CpuVictim(cpu_model)
    data, _ = synth_batch(32, 1024, first=10000)
    xyz = data[:, :, :3].contiguous()
    torch.manual_seed(1)
    label = _labels(victim, xyz)
    target = (label + 1) % 40
    clip_o = lambda pc, ori: O.clip_points_linf(pc, ori, 0.18)  # noqa: E731
    trace, otrace = [], []
    clip = ClipPointsLinf(budget=0.18)

    def recording_clip(pc, ori):
        out = clip(pc, ori)
        trace.append(out.detach().cpu().numpy().copy())
        return out

    gpu_model = copy.deepcopy(cpu_model)
    iters = 3
    winners = []

    from hit_adv_amd.CW import _family
    from hit_adv_amd.model import pct as PCT
    tables, bases = {}, []

    def gpu_run():
        """The attack on the GPU (three iterations: the eager loop, so that the spies see every pass): besides the iterates it
        leaves the discrete choices of every victim pass -- FPS and kNN-grouping tables, max-pool winners -- and, for AOF, the
        eigenbasis of the graph Laplacian it split the cloud in."""
        del trace[:]
        del winners[:]
        del bases[:]
        torch.manual_seed(23)
        log, saved = _record_tables(PCT, ['fps', 'knn_point'])
        tables.clear()
        tables.update(log)
        real_basis = _family.get_Laplace_from_pc

        def basis(pc, *a, **k):
            e, v = real_basis(pc, *a, **k)
            bases.append((e.detach().cpu(), v.detach().cpu()))
            return e, v
        _family.get_Laplace_from_pc = basis
        try:
            return _gpu_attack()
        finally:
            _family.get_Laplace_from_pc = real_basis
            _restore(PCT, saved)

    def _gpu_attack():
        with _pct_pool_winners(winners):
            if which == "knn":
                att = CW.CWKNN(gpu_model, LogitsAdvLoss(kappa=15.), ChamferkNNDist(), recording_clip, attack_lr=1e-2,
                               num_iter=iters, verbose=False)
                return att.attack(xyz, target)
            if which == "advpc":
                att = CW.CWAdvPC(gpu_model, copy.deepcopy(ae), LogitsAdvLoss(kappa=0.), L2Dist(), attack_lr=1e-2, binary_step=1,
                                 num_iter=iters, GAMMA=0.25, clip_func=recording_clip, verbose=False)
                return att.attack(xyz, target, label)[1:]
            att = CW.CWAOF(gpu_model, UntargetedLogitsAdvLoss(kappa=30.), L2Dist(), attack_lr=1e-2, binary_step=1,
                           num_iter=iters, GAMMA=0.25, low_pass=100, clip_func=recording_clip, verbose=False)
            return att.attack(xyz, label)

    def oracle_run(impose):
        """The oracle restatement driving the plain PCT module on the CPU; ``impose``: every discrete choice of the GPU run is
        replayed into it, pass for pass (both loops run the victim in the reference's order) -- the module's samplers return
        the recorded FPS / kNN tables, its max-pools take the recorded winners, AOF's split uses the recorded eigenbasis
        (rocSOLVER's and LAPACK's bases differ inside near-degenerate eigenspaces at the low-pass cut)."""
        del otrace[:]
        torch.manual_seed(23)
        if not impose:
            return _oracle_attack(victim)
        saved = _replay_tables(PCT, {k: list(v) for k, v in tables.items()})
        real_eig, queue = O.laplace_eig, iter(list(bases))
        O.laplace_eig = lambda pc, k=30: next(queue)
        try:
            with _imposed_pool_winners(list(winners)):
                return _oracle_attack(cpu_model)  # the plain module: CpuVictim would put its own samplers in place
        finally:
            O.laplace_eig = real_eig
            _restore(PCT, saved)

    def _oracle_attack(victim):
        with contextlib.nullcontext():
            if which == "knn":
                return O.cw_knn_attack(victim, lambda l, t: O.logits_adv_loss(l, t, 15.), _direct_chamfer_knn, clip_o,
                                       xyz, target, attack_lr=1e-2, num_iter=iters, trace=otrace)
            if which == "advpc":
                return O.cw_family_attack(victim, lambda l, t: O.logits_adv_loss(l, t, 0.), clip_o, xyz, target,
                                          y_truth=label, ae_model=ae, targeted=True, fresh=True, attack_lr=1e-2,
                                          binary_step=1, num_iter=iters, GAMMA=0.25, trace=otrace)[1:]
            return O.cw_aof_attack(victim, lambda l, t: O.untargeted_logits_adv_loss(l, t, 30.), clip_o, xyz, label,
                                   attack_lr=1e-2, binary_step=1, num_iter=iters, GAMMA=0.25, low_pass=100, trace=otrace)

    torch.manual_seed(5)
    ae = _ToyAE().eval()
    final, succ = gpu_run()
    rows = [r.copy() for r in trace[:iters]]
    passes = dict(knn=1, advpc=4, aof=4)[which] * iters  # victim passes inside the loop (gradient + fresh views)
    assert len(rows) == iters and len(winners) >= 3 * passes
    assert [tuple(t.shape) for t in winners[:3]] == [(32, 512, 128), (32, 256, 256), (32, 1024)]
    ori = xyz.transpose(1, 2).numpy()
    lr = 1e-2
    # (1) PARITY: the oracle evaluated where the GPU pass was evaluated -- same sampling tables (bit for bit, by
    # construction of the victim's samplers) and the same max-pool winners, pass for pass.  PCT's final max over the points
    # has near-ties inside fp32 rounding in a few channels of every pass (test_pct_gradient_vs_float64_module_on_the_same_
    # tables: six of 2048 in one pass; each flipped winner re-routes a channel's gradient, and AdvPC / AOF have no distance term:
    # Adam divides the victim's gradient by its own magnitude).  With the winners imposed the iterates are a function of the
    # inputs again and every iterate is held to 1e-5.
    ofinal, osucc = oracle_run(impose=True)
    orows = [r['adv'] for r in otrace]
    assert len(orows) >= iters
    # Adam's step is lr * m / (sqrt(v) + 1e-8): a coordinate whose gradient is itself below the evaluation error (|g| of a
    # few 1e-9 against an fp32 gradient error of ~1e-4 relative L2) may take a step of the other sign, a difference of up to
    # 2 lr that no evaluation order can remove.  So the bar is 1e-5 on the 99th percentile of |gpu - oracle| over all
    # 98,304 coordinates, and the SHARE of coordinates beyond 1e-5 is held to OFF (4x what MI355X achieves).
    # With every discrete choice replayed, EVERY iterate is held to 1e-5 at the 99th percentile of |gpu - oracle| over the
    # 98,304 coordinates (achieved 1.1e-6 ... 5.1e-6); the share of coordinates beyond 1e-5 -- the tiny-gradient coordinates
    # whose Adam step can take the other sign, and what later iterates inherit from them -- is held to 4 x what MI355X
    # achieves (profiles/r04_parity_report.json: knn 4.4e-4 ... 6.9e-4; advpc 1.1e-3, 2.0e-3, 3.4e-3; aof 3.2e-4, 1.9e-3, 3.2e-3).
    BOUNDS = dict(knn=[(1e-5, 2.8e-3)] * 3, advpc=[(1e-5, 4.6e-3), (1e-5, 8e-3), (1e-5, 1.4e-2)],
                  aof=[(1e-5, 1.3e-3), (1e-5, 7.5e-3), (1e-5, 1.3e-2)])[which]
    for i in range(iters):
        assert np.abs(rows[i] - ori).max() <= 0.18 + 1e-6
        err = np.abs(rows[i] - orows[i])
        close(np.quantile(err, 0.99), 0., rtol=0, atol=BOUNDS[i][0],
              what='cfg5 %s iterate %d, same discrete choices: 99th percentile |gpu - oracle|' % (which, i))
        close(float((err > 1e-5).mean()), 0., rtol=0, atol=BOUNDS[i][1],
              what='cfg5 %s iterate %d, same discrete choices: share of coordinates off by > 1e-5' % (which, i))
        note('cfg5 %s iterate %d, same discrete choices: median |gpu - oracle|' % (which, i), np.median(err))
        note('cfg5 %s iterate %d, same discrete choices: max |gpu - oracle|' % (which, i), err.max())
    close(float((np.abs(final - ofinal) > 1e-5).mean()), 0., rtol=0, atol=BOUNDS[-1][1],
          what='cfg5 %s returned clouds, same discrete choices: share off by > 1e-5' % which)
    assert final.shape == ofinal.shape and int(succ) == int(osucc)
    # (2) for the record: the free-running oracle (its own winners).  Statistics only -- what the flipped winners cost.
    oracle_run(impose=False)
    for i in range(iters):
        err = np.abs(rows[i] - otrace[i]['adv'])
        assert err.max() <= 2 * lr * (i + 1) + 1e-6, (i, err.max())        # Adam's reach: a sanity check, not a parity bound
        note('cfg5 %s iterate %d, free-running oracle: median |gpu - oracle|' % (which, i), np.median(err))
        note('cfg5 %s iterate %d, free-running oracle: 99th percentile |gpu - oracle|' % (which, i), np.quantile(err, 0.99))
    if which == "knn":  # the distance term conditions every point: the free-running iterates stay on the oracle's too
        for i in range(iters):
            close(np.median(np.abs(rows[i] - otrace[i]['adv'])), 0., rtol=0, atol=1e-6,
                  what='cfg5 knn iterate %d, free-running: median |gpu - oracle|' % i)


def _tuned_victim(name):
    """The victims of bench.py's cfg3 / cfg4 / cfg5 (bench.VICTIM_TUNING: seeded init, every weight x 1.5, PointNet++ / PCT with
    shaken BatchNorm statistics) -- victims a bounded attack can move, so that some clouds succeed and some do not."""
    from hit_adv_amd.Dataset.synthetic import shake_bn, sharpen
    torch.manual_seed(0)
    if name == 'dgcnn':
        from hit_adv_amd.model.dgcnn import DGCNN_cls
        return sharpen(DGCNN_cls(argparse.Namespace(k=5, emb_dims=1024, dropout=0.2), output_channels=40).eval(), 1.5)
    if name == 'pointnet++':
        from hit_adv_amd.model.pointnet2 import get_model
        return shake_bn(sharpen(get_model(16, normal_channel=False).eval(), 1.5), seed=2)
    from hit_adv_amd.model.pct import Pct
    return shake_bn(sharpen(Pct(argparse.Namespace(dropout=0.2), output_channels=40).eval(), 1.5), seed=2)


@pytest.mark.parametrize("name", ["dgcnn", "pointnet++", "pct"])
def test_other_victims_twenty_iterations_with_both_bookkeeping_branches_vs_cpu_oracle(name):
    """VERDICT r04 #1c: the other configurations' victims (as bench.py builds them) under HiT-ADV for 20 iterations, B = 8,
    N = 1024, against the CPU oracle -- on victims where some clouds succeed and some do not (5 / 8, 6 / 8, 2 / 8 on MI355X, the
    same clouds on both sides), so that the success branch of the best tracking and BOTH directions of the bisection run.
    Free-running: the oracle's victim draws its own sampling tables (PointNet++ / PCT: from the same CPU generator state),
    nothing is imposed.  Achieved on MI355X: |gpu - oracle| of the returned clouds <= 1.9e-4 (DGCNN), 6e-7 (PointNet++),
    5.1e-5 (PCT); asserted: the same centres, the same success flags up to one cloud inside fp32 noise, the 99th percentile
    of the result at 1e-3 (5x the worst achieved) and its maximum at one Adam step."""
    from hit_adv_amd.ShapeAttack.HiT_ADV import HiT_ADV
    from hit_adv_amd.util.adv_utils import UntargetedLogitsAdvLoss
    cpu_model = _tuned_victim(name)
    victim = VG.CpuVictim(cpu_model) if name in ('pointnet++', 'pct') else cpu_model
    data, _ = synth_batch(8, 1024, first=7000)
    torch.manual_seed(1)
    label = _labels(victim, data)
    hp = dict(binary_step=1, num_iter=20, **HP)
    oracle = O.HiTADVOracle(victim, lambda l, t: O.untargeted_logits_adv_loss(l, t, 30.), **hp)
    torch.manual_seed(21)
    with contextlib.redirect_stdout(io.StringIO()):
        obest, osucc = oracle.attack(data, label)
    att = HiT_ADV(copy.deepcopy(cpu_model), UntargetedLogitsAdvLoss(30.), verbose=False, **hp)
    torch.manual_seed(21)
    best, succ = att.attack(data, label)
    ws = next(iter(att._ws.values()))
    same = (ws.central.cpu() == oracle.state['central']).all(dim=1).float().mean().item()
    assert same >= 0.99, same
    ok_gpu = att.last_lower_bound.numpy() > 0
    ok_cpu = oracle.state['lower'].numpy() > 0
    note('%s: successes on the GPU' % name, float(ok_gpu.sum()))
    note('%s: successes in the oracle' % name, float(ok_cpu.sum()))
    note('%s: clouds whose success flag differs' % name, float((ok_gpu != ok_cpu).sum()))
    assert 0 < int(succ) < 8 and 0 < int(osucc) < 8          # both branches ran, on both sides
    assert int((ok_gpu != ok_cpu).sum()) <= 1                  # (a cloud that flips inside fp32 noise in the last iterations)
    agree = ok_gpu == ok_cpu
    err = np.abs(best - obest)[agree]
    note('%s: result, largest |gpu - oracle| over the agreeing clouds' % name, float(err.max()))
    note('%s: result, 99th percentile' % name, float(np.quantile(err, 0.99)))
    assert float(np.quantile(err, 0.99)) <= 1e-3 and float(err.max()) <= 5e-2


def test_fps_tables_with_two_attacks_in_flight_match_the_other_sampling_kernel():
    """Round 5: two HiT-ADV attacks on PointNet++ in flight on two streams, every FPS table of every forward pass computed by both
    sampling kernels (tools/fps_check.py: fps_lean first, again right behind it, the 64-bit-key kernel, fps_lean once more).  The
    first fps_lean build -- distances on packed f32 instructions -- passed every bit-exact test alone on the GPU and returned a
    different table for 1-30 % of the clouds here (a lane missing one update of its running distance); on plain instructions the
    four tables of a call are the same bits, and the inputs do not change under the kernels."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import fps_check
    from hit_adv_amd import ops
    from hit_adv_amd.ShapeAttack.HiT_ADV import HiT_ADV
    from hit_adv_amd.util.adv_utils import UntargetedLogitsAdvLoss
    keep = ops.fps_from_start
    try:
        fps_check.install()
        fps_check.reset()
        model = _tuned_victim('pointnet++').cuda()
        batches = []
        for i in range(2):
            data, _ = synth_batch(16, 2048, first=300 * i)
            data = data.cuda()
            with torch.no_grad():
                o = model(data[:, :, :3].transpose(1, 2).contiguous())
            batches.append((data, (o[0] if isinstance(o, tuple) else o).argmax(1)))
        torch.manual_seed(5)
        att = HiT_ADV(model, adv_func=UntargetedLogitsAdvLoss(kappa=30.), binary_step=1, num_iter=25, verbose=False, use_graph=False,
                      attack_lr=1e-2, central_num=192, total_central_num=256, init_weight=10., max_weight=80., cd_weight=1e-4, ker_weight=1.,
                      hide_weight=1., curv_loss_knn=16, max_sigm=1.2, min_sigm=0.1, budget=0.55)
        att.attack_many(batches)
        torch.cuda.synchronize()
        c = fps_check.counts()
    finally:
        ops.fps_from_start = keep
    assert c['tables'][1] >= 2 * 16 * 25 * 2  # two FPS calls per forward pass, both attacks
    assert c['tables'][0] == 0 and c['more'] == [0, 0, 0], c
    assert c['inputs_changed'] == [0, 0], c
