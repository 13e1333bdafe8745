#!/usr/bin/env python3
"""Per-kernel timing of libhitadv_hip at the cfg2 problem sizes (B=32, N=1024, C=192), HIP events on
the launch stream.  Prints one JSON object; numbers feed DESIGN.md's kernel table.

    python tools/kbench.py [--reps 200]
"""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hit_adv_amd import _lib, ops  # noqa: E402


def timed_eager(fn, reps=20):
    """Average duration of a torch-level function (its own launches + allocator), events on the current stream."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e3 / reps


def timed(fn, reps):
    """Average launch duration: 20 back-to-back launches captured into a hipGraph (so the host's ctypes call
    rate cannot open gaps between them), replayed reps/20 times between two events."""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(20):
            fn(s)
    n = max(1, reps // 20)
    for _ in range(n):  # untimed: lets the clocks settle under this kernel's load
        g.replay()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0.record()
    for _ in range(n):
        g.replay()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e3 / (n * 20)  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=1000)
    ap.add_argument('-B', type=int, default=32)
    ap.add_argument('-N', type=int, default=1024)
    ap.add_argument('-C', type=int, default=192)
    a = ap.parse_args()
    B, N, C = a.B, a.N, a.C
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, N, 3, generator=g).cuda()
    y = torch.randn(B, N, 3, generator=g).cuda()
    P = torch.empty(B, N, N, device='cuda')
    s0 = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    out = {'B': B, 'N': N, 'C': C}

    xu = (torch.rand(B, N, 3, generator=g) * 2 - 1).cuda()
    yu = (torch.rand(B, N, 3, generator=g) * 2 - 1).cuda()
    us = timed(lambda s=s0: lib.hitadv_pairwise_sqdist(p(xu), p(yu), p(P), B, N, N, 3, 1, s), a.reps)
    out['pairwise_gram_uniform_inputs'] = dict(us=round(us, 2), GBps=round((4 * N * N + 24 * N) * B / us / 1e3, 1))
    for form, name in ((0, 'direct'), (1, 'gram')):
        us = timed(lambda s=s0: lib.hitadv_pairwise_sqdist(p(x), p(y), p(P), B, N, N, 3, form, s), a.reps)
        by = (4 * N * N + 24 * N) * B
        out['pairwise_' + name] = dict(us=round(us, 2), GBps=round(by / us / 1e3, 1), frac_of_8TBps=round(by / us / 1e3 / 8000, 4))
    mx, my = torch.empty(B, N, device='cuda'), torch.empty(B, N, device='cuda')
    ax, ay = torch.empty(B, N, device='cuda', dtype=torch.int32), torch.empty(B, N, device='cuda', dtype=torch.int32)
    us = timed(lambda s=s0: lib.hitadv_nn_min(p(x), p(y), B, N, N, 3, 0, p(mx), p(ax), p(my), p(ay), None, s), a.reps)
    pairs = 2 * B * N * N  # both directions evaluated independently
    out['nn_min'] = dict(us=round(us, 2), Gpairs_per_s=round(pairs / us / 1e3, 1),
                         valu_TFLOPs_at_11_ops=round(pairs * 11 / us / 1e6, 2))
    gx = torch.empty_like(x)
    gy = torch.empty_like(y)
    gm = torch.randn(B, N, device='cuda')
    us = timed(lambda s=s0: lib.hitadv_nn_min_bwd(p(x), p(y), p(ax), p(ay), p(gm), p(gm), B, N, N, 3, p(gx), p(gy), s), a.reps)
    out['nn_min_bwd'] = dict(us=round(us, 2))
    for K in (1, 6, 17):
        d = torch.empty(B, N, K, device='cuda')
        ix = torch.empty(B, N, K, device='cuda', dtype=torch.int64)
        us = timed(lambda s=s0: lib.hitadv_knn_points(p(x), p(x), B, N, N, K, 0, p(d), p(ix), 1, s), max(20, a.reps // 4))
        out['knn_K%d' % K] = dict(us=round(us, 2), Gpairs_per_s=round(B * N * N / us / 1e3, 1))
        if K in (6, 17):  # its backward (query-side gather + reference-side owner-computes scatter)
            gd = torch.randn(B, N, K, generator=g).cuda()
            g1, g2 = torch.empty_like(x), torch.empty_like(x)
            us = timed(lambda s=s0: lib.hitadv_knn_points_bwd(p(x), p(x), p(ix), 1, p(gd), B, N, N, K, p(g1), p(g2), s), max(20, a.reps // 4))
            out['knn_bwd_K%d' % K] = dict(us=round(us, 2))
    ori = x.transpose(1, 2).contiguous()
    central = ori[:, :, :C].contiguous()
    Pm = (torch.rand(B, C, 3, generator=g) * 0.55).cuda()
    sig = (0.1 + torch.rand(B, C, generator=g) * 1.1).cuda()
    adv, inv = torch.empty_like(ori), torch.empty(B, N, device='cuda')
    us = timed(lambda s=s0: lib.hitadv_deform_fwd(p(ori), p(central), p(Pm), p(sig), B, N, C, p(adv), p(inv), s), a.reps)
    out['deform_fwd'] = dict(us=round(us, 2), Gpairs_per_s=round(B * N * C / us / 1e3, 2))
    part = torch.empty(lib.hitadv_deform_bwd_scratch_floats(B, N, C), device='cuda')
    gp, gs = torch.empty_like(Pm), torch.empty_like(sig)
    up = torch.randn(B, 3, N, device='cuda')
    us = timed(lambda s=s0: lib.hitadv_deform_bwd(p(ori), p(central), p(Pm), p(sig), p(adv), p(inv), p(up), B, N, C,
                                             p(part), p(gp), p(gs), s), a.reps)
    out['deform_bwd(+reduce)'] = dict(us=round(us, 2), Gpairs_per_s=round(B * N * C / us / 1e3, 2))
    start = torch.zeros(B, dtype=torch.int64, device='cuda')
    fi = torch.empty(B, 256, dtype=torch.int64, device='cuda')
    us = timed(lambda s=s0: lib.hitadv_fps_from_start(p(x), p(start), B, N, 256, p(fi), s), 20)
    out['fps_from_start_m256'] = dict(us=round(us, 1), us_per_step=round(us / 256, 3))
    # the step latency at the sizes the victims use (PointNet++: 1024 -> 512 -> 128, PCT: 1024 -> 512 -> 256), old kernel beside the new
    lib.hitadv_debug_fps_form.restype = ctypes.c_int
    shipped_form = lib.hitadv_debug_fps_form(-1)
    for nm, fn, n_, m_ in (('fps_from_start', lib.hitadv_fps_from_start, 2048, 512), ('fps_from_start', lib.hitadv_fps_from_start, 1024, 512),
                           ('fps_from_start', lib.hitadv_fps_from_start, 512, 128), ('fps_pct', lib.hitadv_fps_pct, 1024, 512),
                           ('fps_pct', lib.hitadv_fps_pct, 512, 256)):
        xs = torch.randn(B, n_, 3, generator=g).cuda()
        fo = torch.empty(B, m_, dtype=torch.int64, device='cuda')
        row = {}
        for form in (0, 1):
            lib.hitadv_debug_fps_form(form)
            us = timed(lambda s=s0: fn(p(xs), p(start), B, n_, m_, p(fo), s), 20)
            row['key64' if form == 0 else 'lean'] = dict(us=round(us, 1), us_per_step=round(us / m_, 3))
        out[f'{nm}_N{n_}_m{m_}'] = row
    lib.hitadv_debug_fps_form(shipped_form)
    f32 = torch.empty(B, 51, dtype=torch.int32, device='cuda')
    us = timed(lambda s=s0: lib.hitadv_furthest_point_sampling(B, N, 51, p(x), None, p(f32), s), 20)
    out['fps_ext_m51'] = dict(us=round(us, 1), us_per_step=round(us / 50, 3))
    new_xyz = x[:, :51].contiguous()
    bq = torch.empty(B, 51, 49, dtype=torch.int32, device='cuda')
    us = timed(lambda s=s0: lib.hitadv_query_ball_point(B, N, 51, ctypes.c_float(0.22), 49, p(new_xyz), p(x), p(bq), s), a.reps)
    out['ball_query_m51_ns49'] = dict(us=round(us, 2))
    # victim helper: fused 128->1024 shared layer + max over points on the f32 matrix cores, and what it replaced
    h2 = torch.randn(B * N, 128, generator=g).cuda()
    Wt = (torch.randn(128, 1024, generator=g) * 0.1).cuda()
    bias = torch.randn(1024, generator=g).cuda()
    n = lib.hitadv_linear_max_fwd_scratch(B, N, 1024)
    pv, pi = torch.empty(n, device='cuda'), torch.empty(n, device='cuda', dtype=torch.int32)
    mo, mi = torch.empty(B, 1024, device='cuda'), torch.empty(B, 1024, device='cuda', dtype=torch.int64)
    tk = torch.zeros(4096, device='cuda', dtype=torch.int32)  # split tickets (self-resetting)
    us = timed(lambda s=s0: lib.hitadv_linear_max_fwd(p(h2), p(Wt), p(bias), B, N, 128, 1024, 1, p(pv), p(pi), p(mo),
                                                      p(mi), p(tk), s), a.reps)
    out['linear_max_fwd_128x1024(+merge)'] = dict(us=round(us, 2), TFLOPs=round(2 * B * N * 128 * 1024 / us / 1e6, 1),
                                                  frac_of_157TF=round(2 * B * N * 128 * 1024 / us / 1e6 / 157.3, 3))
    yb = torch.empty(B * N, 1024, device='cuda')
    n2 = lib.hitadv_max_over_points_scratch(B, 1024)
    pv2, pi2 = torch.empty(n2, device='cuda'), torch.empty(n2, device='cuda', dtype=torch.int32)

    def unfused(s=s0):
        torch.mm(h2, Wt, out=yb)  # torch's current stream is the one `s` names, eagerly and under capture
        lib.hitadv_max_over_points(p(yb), B, N, 1024, p(bias), 1, p(pv2), p(pi2), p(mo), p(mi), s)
    us = timed(unfused, a.reps)
    out['mm+max_over_points_128x1024'] = dict(us=round(us, 2))
    # PointNet engine building blocks
    fcs = torch.zeros(lib.hitadv_fc_layer_scratch_floats(B, 4096, 4096) + (1 << 20), device='cuda')
    for K, NOUT in ((1024, 512), (512, 256), (256, 9), (256, 4096), (256, 40), (40, 256), (256, 512), (512, 1024),
                    (4096, 256), (9, 256)):
        xin = torch.randn(B, K, generator=g).cuda()
        wt = torch.randn(K, NOUT, generator=g).cuda()
        bo = torch.randn(NOUT, generator=g).cuda()
        oo = torch.empty(B, NOUT, device='cuda')
        us = timed(lambda s=s0: lib.hitadv_fc_layer(p(xin), p(xin), p(wt), p(bo), B, K, NOUT, 1, p(oo), p(fcs), s), a.reps)
        out['fc_layer_%dx%d' % (K, NOUT)] = dict(us=round(us, 2))
    # rowmlp_bwd stage 0 with the arg-max table of a real forward pass (hot points) and with a uniform one
    from hit_adv_amd.model.pointnet import PointNetFeatureModel
    torch.manual_seed(0)
    view = PointNetFeatureModel(40, normal_channel=False).cuda().eval().attack_view()
    xin3 = x.transpose(1, 2).contiguous()
    R = B * N
    a1s, a2s = torch.empty(R, 64, device='cuda'), torch.empty(R, 128, device='cuda')
    ops.pointnet_rowmlp_fwd(0, B, N, view.s2_w, view.s2_b, a2s, x=xin3, W0=view.s1_w, b0=view.s1_b, o0=a1s)
    gs, js = ops.linear_max_fwd(a2s, view.s3_w, B, N, bias=view.s3_b, relu=True)
    dgs = torch.randn(B, 1024, generator=g).cuda()
    dpin, dx = torch.zeros(B, 3, N, device='cuda'), torch.empty(B, 3, N, device='cuda')
    ju = (torch.arange(1024, device='cuda') % N).expand(B, 1024).contiguous()
    ones = torch.ones(B, 1024, device='cuda')
    for name, jj, gm in (('real_argmax', js, gs), ('uniform_argmax', ju, ones)):
        pres = torch.zeros(B, (N + 63) // 64, device='cuda', dtype=torch.int64)  # no incoming rows: gather hits only ([B,tiles,words])
        over = torch.zeros(B, (N + 63) // 64, device='cuda', dtype=torch.int32)
        for mode, tag in ((0, '_f32'), (1, '_fp16x2')):
            us = timed(lambda s=s0: lib.hitadv_pointnet_rowmlp_bwd(0, p(dgs), p(gm), p(jj), p(view.s3_wr), 1024, p(a2s),
                                                                   p(view.s2_wr), p(a1s), None, None, None, p(view.s1_wr),
                                                                   None, None, p(dpin), None, p(dx), None, None, p(over), 1, B, N, mode, s),
                       a.reps)
            out['rowmlp_bwd0_' + name + '_dense_rows' + tag] = dict(us=round(us, 2))
            us = timed(lambda s=s0: lib.hitadv_pointnet_rowmlp_bwd(0, p(dgs), p(gm), p(jj), p(view.s3_wr), 1024, p(a2s),
                                                                   p(view.s2_wr), p(a1s), None, None, None, p(view.s1_wr),
                                                                   None, None, p(dpin), None, p(dx), p(pres), None, p(over), 1, B, N, mode, s),
                       a.reps)
            out['rowmlp_bwd0_' + name + tag] = dict(us=round(us, 2))
    for mode, tag in ((0, '_f32'), (1, '_fp16x2')):
        us = timed(lambda s=s0: lib.hitadv_pointnet_rowmlp_fwd(0, p(xin3), None, None, p(view.s1_w), p(view.s1_b), None, None,
                                                               p(view.s2_w), p(view.s2_b), None, p(a1s), None, p(a2s), B, N,
                                                               mode, p(view.range_flag), s), a.reps)
        out['rowmlp_fwd0' + tag] = dict(us=round(us, 2))
    for D in (64, 128):
        feat = torch.randn(B, N, D, generator=g).cuda()
        fx = (feat * feat).sum(2)
        fi = torch.empty(B, N, 5, dtype=torch.int64, device='cuda')
        us = timed(lambda s=s0: lib.hitadv_knn_features(p(feat), p(fx), B, N, D, 5, p(fi), s), a.reps)
        out['knn_features_D%d_K5' % D] = dict(us=round(us, 2), TFLOPs=round(2.0 * B * N * N * D / us / 1e6, 1))
    # fp32-accurate GEMMs on the fp16 cores against the library's f32 GEMM, and DGCNN's fused embedding layer
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    for M, K, N in ((32768, 512, 1024), (32768, 1024, 512), (131072, 128, 256), (524288, 128, 128), (32768, 256, 256)):
        xa = torch.randn(M, K, generator=g).cuda()
        Wn = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
        Wt_ = Wn.t().contiguous()
        Wp = ops.split_rows_f16x2(Wn, flag)
        cbuf = torch.empty(M, N, device='cuda')
        # the ring kernel and the staged kernel (operands through registers, two LDS stages) alternately, the better of two each:
        # whichever is measured first after a change of kernel runs ~10 % slow (clocks)
        us, us_staged = 1e30, 1e30
        for _ in range(2):
            lib.hitadv_debug_g16_ring(0)
            us_staged = min(us_staged, timed(lambda s=s0: lib.hitadv_gemm_f16x2(p(xa), None, p(Wp), None, M, N, K, 0, p(cbuf), p(flag), s), 200))
            lib.hitadv_debug_g16_ring(1)
            us = min(us, timed(lambda s=s0: lib.hitadv_gemm_f16x2(p(xa), None, p(Wp), None, M, N, K, 0, p(cbuf), p(flag), s), 200))
        us_lib = timed_eager(lambda: torch.mm(xa, Wt_, out=cbuf))
        out['gemm_f16x2_%dx%dx%d' % (M, K, N)] = dict(us=round(us, 1), useful_TFLOPs=round(2.0 * M * K * N / us / 1e6, 1),
                                                      executed_f16_TFLOPs=round(6.0 * M * K * N / us / 1e6, 1),
                                                      staged_kernel_us=round(us_staged, 1), torch_mm_f32_us=round(us_lib, 1))
        del xa, Wn, Wt_, Wp, cbuf
    Bc, Nc, Cin, Cc = 32, 1024, 512, 1024
    xa = torch.randn(Bc * Nc, Cin, generator=g).cuda().requires_grad_()
    Wn = (torch.randn(Cc, Cin, generator=g) / Cin ** 0.5).cuda()
    bc = torch.randn(Cc, generator=g).cuda()
    Wp, Wtp = ops.split_rows_f16x2(Wn, flag), ops.split_rows_f16x2(Wn.t().contiguous(), flag)
    wgt = torch.randn(Bc, 2 * Cc, generator=g).cuda()
    o = ops.linear_lrelu_pool(xa, Wp, Wtp, bc, Bc, Nc, 0.2, flag)
    us_f = timed_eager(lambda: ops.linear_lrelu_pool(xa.detach(), Wp, Wtp, bc, Bc, Nc, 0.2, flag))
    us_b = timed_eager(lambda: torch.autograd.grad(o, xa, wgt, retain_graph=True))
    Wt_ = Wn.t().contiguous()

    def lib_fwd():
        return ops.lrelu_pool(torch.addmm(bc, xa.detach(), Wt_).view(Bc, Nc, Cc), 0.2)
    o2 = ops.lrelu_pool(torch.addmm(bc, xa, Wt_).view(Bc, Nc, Cc), 0.2)
    out['linear_lrelu_pool_32x1024_512x1024'] = dict(fwd_us=round(us_f, 1), bwd_us=round(us_b, 1),
                                                     addmm_plus_lrelu_pool_fwd_us=round(timed_eager(lib_fwd), 1),
                                                     addmm_plus_lrelu_pool_bwd_us=round(timed_eager(
                                                         lambda: torch.autograd.grad(o2, xa, wgt, retain_graph=True)), 1))
    print(json.dumps(out))


if __name__ == '__main__':
    main()
