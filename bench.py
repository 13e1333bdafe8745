#!/usr/bin/env python3
"""Benchmarks of the HiT-ADV hot path on MI355X.  Headline (default, ``--config cfg2``): attacked point-clouds/sec of
HiT-ADV on the PointNet victim, N=1024, 500 iterations x 10 binary steps, batch 32 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|cfg4|cfg5]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full attack over one batch of synthetic clouds, inputs resident in HBM before the timed region:
  cfg2  HiT_ADV.attack(), 32 clouds x 1024 points, PointNet victim           (BASELINE.json configs[1], the metric's config)
  cfg3  HiT_ADV.attack(), 32 clouds x 1024 points per GPU, DGCNN victim, k=5 (configs[2]: 256 clouds over 8 GPUs)
  cfg4  HiT_ADV.attack(), 64 clouds x 2048 points, PointNet++ SSG victim     (configs[3])
  cfg5  CWAdvPC + CWKNN + CWAOF, one after the other, 32 clouds x 1024 points, PCT victim (configs[4])
With N > 1 every rank attacks its own batches (independent shards, weak scaling, no data-path collective); the only
collectives are the barrier / MAX for timing and one SUM all-reduce of the success counters.

Rank 0 prints ONE JSON line; DESIGN.md section "Measurement" defines every field.
"""
import argparse
import ctypes
import json
import os
import sys
import time
import warnings

# Independent stacks of attacks run on their own HIP streams; the runtime multiplexes streams onto 4 hardware queues by
# default, and three or four streams on four queues shared with everything else serialise again (measured, 12 attacks in
# three stacks: 30.3 clouds/s with 4 queues, 37.4 with 8).  Must be set before the HIP runtime starts.
_QUEUES_PRESET = os.environ.get("GPU_MAX_HW_QUEUES")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# What the line reports as `hip_hardware_queues`: the variable's value only if the runtime cannot have started before it was
# in the environment -- i.e. it was exported by the caller, or nothing had initialised the GPU when this file set it (a
# profiler's preloaded library does: under rocprofv3 export GPU_MAX_HW_QUEUES=8 in the shell, the measurement scripts, e.g. tools/r05_measure.sh).
if _QUEUES_PRESET is not None:
    HW_QUEUES = int(_QUEUES_PRESET)
elif torch.cuda.is_initialized() or os.environ.get("ROCPROFILER_LIBRARY_CTOR") or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
    HW_QUEUES = "unknown (the HIP runtime may have started before GPU_MAX_HW_QUEUES was set: 4 unless exported earlier)"
else:
    HW_QUEUES = 8

SEQUENTIAL_SWEEP = False  # cfg5: --sequential-sweep
NUM_ITER, BINARY_STEP = 500, 10
HP = dict(attack_lr=1e-2, central_num=192, total_central_num=256, init_weight=10., max_weight=80.,
          cd_weight=1e-4, ker_weight=1., hide_weight=1., curv_loss_knn=16, max_sigm=1.2, min_sigm=0.1,
          budget=0.55)  # eval.py:126-133 / :48-62 defaults
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md, HBM3E spec peak
F32_MFMA_PEAK = 157.3   # TFLOP/s, dense f32-input MFMA (= the f32 vector peak), same guide

CONFIGS = {
    'cfg2': dict(victim='pointnet', B=32, N=1024, classes=40, attack='hit_adv', steps=24, warmup=12, concurrent=24,
                 metric="attacked point-clouds/sec (HiT-ADV, PointNet, N=1024, 500 iters)",
                 workload="cfg2: synthetic ModelNet40-shaped clouds, 1024 pts, batch 32 per GPU, PointNet victim (random "
                          "init, eval mode), HiT-ADV eval.py hyper-parameters, num_iter=500 x binary_step=10 = 5000 inner "
                          "iterations per attack()"),
    'cfg3': dict(victim='dgcnn', B=32, N=1024, classes=40, attack='hit_adv', steps=4, warmup=2, concurrent=4,
                 metric="attacked point-clouds/sec (HiT-ADV, DGCNN k=5, N=1024, 500 iters)",
                 workload="cfg3: synthetic ModelNet40-shaped clouds, 1024 pts, batch 32 per GPU (256 over 8 GPUs), DGCNN "
                          "victim k=5 (seeded init, weights x 1.5, eval mode), HiT-ADV eval.py hyper-parameters, 500 x 10 iterations"),
    'cfg4': dict(victim='pointnet++', B=64, N=2048, classes=16, attack='hit_adv', steps=4, warmup=0, concurrent=4,
                 metric="attacked point-clouds/sec (HiT-ADV, PointNet++ SSG, N=2048, 500 iters)",
                 workload="cfg4: synthetic ShapeNetPart-shaped clouds, 2048 pts, batch 64, PointNet++ SSG victim (16 object "
                          "categories, seeded init, weights x 1.5, shaken BN statistics, eval mode), HiT-ADV eval.py hyper-parameters, 500 x 10 iterations"),
    'cfg5': dict(victim='pct', B=32, N=1024, classes=40, attack='cw_sweep', steps=2, warmup=1, concurrent=1,  # (one sweep = ~38 s)
                 metric="point-clouds/sec through the AdvPC + kNN + AOF sweep (PCT, N=1024)",
                 workload="cfg5: synthetic ModelNet40-shaped clouds, 1024 pts, batch 32, PCT victim (seeded init, weights x 1.5, shaken BN statistics, eval mode); "
                          "every cloud is attacked by CWAdvPC (2 x 200 iterations, point-wise stand-in auto-encoder: the "
                          "reference ships none), CWKNN (2500 iterations, ChamferkNNDist) and CWAOF (2 x 200 iterations, "
                          "low_pass 100), constructor defaults of CW/AdvPC.py, CW/kNN.py, CW/AOF.py, ClipPointsLinf(0.18)"),
}


def synth(first, count, npoint):
    from hit_adv_amd.Dataset.synthetic import synth_batch
    return synth_batch(count, npoint, first=first)


# A default-initialised deep victim in eval mode answers with its last layer's bias: every cloud lands in one class and nothing a
# bounded attack does moves it (round 4: 0 / 256 successes on PointNet++, 0 / 96 on PCT, so the success branch of the
# bookkeeping never ran in those lines).  cfg3 - cfg5 therefore sharpen the seeded victim (every weight x 1.5) and, for
# PointNet++ / PCT, move its BatchNorm statistics off 0 / 1 (Dataset/synthetic.py; chosen with tools/explore_success.py so that
# some clouds succeed and some do not).  Same kernels, same shapes: throughput is unaffected.  cfg2 keeps round 1's victim
# (168 - 172 of 640 succeed), so its numbers stay comparable across rounds.
VICTIM_TUNING = {'dgcnn': dict(gain=1.5, shake=None), 'pointnet++': dict(gain=1.5, shake=2), 'pct': dict(gain=1.5, shake=2)}


def build_victim(cfg):
    from hit_adv_amd.Dataset.synthetic import shake_bn, sharpen
    torch.manual_seed(0)
    name = cfg['victim']
    if name == 'pointnet':
        from hit_adv_amd.model.pointnet import PointNetFeatureModel
        return PointNetFeatureModel(cfg['classes'], normal_channel=False).eval()
    if name == 'dgcnn':
        from hit_adv_amd.model.dgcnn import DGCNN_cls
        model = DGCNN_cls(argparse.Namespace(k=5, emb_dims=1024, dropout=0.2), output_channels=cfg['classes']).eval()
    elif name == 'pointnet++':
        from hit_adv_amd.model.pointnet2 import get_model
        model = get_model(cfg['classes'], normal_channel=False).eval()
    else:
        from hit_adv_amd.model.pct import Pct
        model = Pct(argparse.Namespace(dropout=0.2), output_channels=cfg['classes']).eval()
    tune = VICTIM_TUNING[name]
    sharpen(model, tune['gain'])
    if tune['shake'] is not None:
        shake_bn(model, seed=tune['shake'])
    return model


class ToyAE(torch.nn.Module):
    """Point-wise stand-in for AdvPC's auto-encoder ([B,3,K] -> [B,3,K]); the reference ships no auto-encoder."""

    def __init__(self):
        super().__init__()
        self.enc, self.dec = torch.nn.Conv1d(3, 16, 1), torch.nn.Conv1d(16, 3, 1)

    def forward(self, x):
        return x + 0.05 * self.dec(torch.tanh(self.enc(x)))


def logits_of(model, x):
    out = model(x)
    return out[0] if isinstance(out, tuple) else out


# --------------------------------------------------------------------------------------------- kernel timing
def graph_timed(launch, per_graph=20, reps=50):
    """Average duration (us) of one launch: `per_graph` back-to-back launches captured into a hipGraph (the host's ctypes
    call rate must not open gaps between launches), replayed `reps` times between two events recorded on the stream the
    replays run on, after an untimed pass that lets the clocks settle under this kernel's load."""
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(per_graph):
            launch(stream)
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        g.replay()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e3 / (reps * per_graph)


def _traffic(kernel_key):
    """HBM bytes per launch from the newest committed PMC summary (separate rocprofv3 --pmc passes of tools/kbench.py at
    the same sizes, corrected per MI355X_MICROARCH.md: WRITE_SIZE exact, FETCH_SIZE x2).  Not measured in this run: the
    counters need the profiler; `traffic_source` in the line names the file."""
    import glob
    best = (None, None)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*kbench_traffic.json"))):
        with open(path) as f:
            v = json.load(f).get(kernel_key, {}).get("hbm_bytes_per_launch")
        if v is not None:
            best = (v, os.path.relpath(path, ROOT))
    return best


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def roofline_pairwise(dev, B=32, N=1024):
    """K1: the materialising 1024x1024 pairwise kernel (the kernel north_star's >= 70 %-of-HBM target names)."""
    from hit_adv_amd import _lib, ops
    lib = _lib.load()
    x, y = torch.randn(B, N, 3, device=dev), torch.randn(B, N, 3, device=dev)
    P = torch.empty(B, N, N, device=dev)
    us = graph_timed(lambda s: lib.hitadv_pairwise_sqdist(_p(x), _p(y), _p(P), B, N, N, 3, ops.FORM_GRAM, s))
    alg = (4 * N * N + 12 * (N + N)) * B  # 4,218,880 B per cloud pair
    ach = alg / (us * 1e-6) / 1e9
    traffic, src = _traffic("hitadv::pairwise3_vec4<1>")
    return dict(kernel="pairwise3_vec4<gram> (hitadv_pairwise_sqdist, B=%d, %dx%d)" % (B, N, N), bound="hbm",
                achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4), traffic=traffic,
                traffic_source=src, us_per_launch=round(us, 2), algorithmic_bytes=alg)


def roofline_knn_features(dev, B=32, N=1024, D=64, k=5):
    """cfg3's dominant own kernel: DGCNN's feature-space kNN graph (scores on the bf16 matrix cores at fp32 accuracy --
    both operands as three bf16 pieces, six MFMAs per 16 values of k, fp32 accumulator -- selection in the same launch);
    2*B*N*N*D useful flop per launch (6x that executed), three launches per forward (D = 64, 64, 128).  `achieved` /
    `peak` / `frac` price the USEFUL flops against the f32 matrix peak, the yardstick of an fp32-accurate product; the
    executed bf16 rate is given beside it."""
    from hit_adv_amd import _lib
    lib = _lib.load()
    f = torch.randn(B, N, D, device=dev)
    xx = (f * f).sum(-1).contiguous()
    idx = torch.empty(B, N, k, device=dev, dtype=torch.int64)
    us = graph_timed(lambda s: lib.hitadv_knn_features(_p(f), _p(xx), B, N, D, k, _p(idx), s))
    flops = 2.0 * B * N * N * D
    ach = flops / us / 1e6
    return dict(kernel="knn_feat_k<%d,%d> (hitadv_knn_features, B=%d, N=%d)" % (D, k, B, N), bound="mfma",
                achieved=round(ach, 1), peak=F32_MFMA_PEAK, unit="TFLOP/s", frac=round(ach / F32_MFMA_PEAK, 4), traffic=None,
                us_per_launch=round(us, 2), flops_per_launch=flops, dtype="3 x bf16 -> f32",
                executed_bf16_tflops=round(6 * ach, 1), peak_bf16=2500.0, frac_of_bf16_peak=round(6 * ach / 2500.0, 4))


def roofline_group_add_relu(dev, B, N, S, ns, C, what):
    """cfg4 / cfg5: first layer of a sample-and-group block, H[b,s,j,:] = relu(U[b, idx[b,s,j], :] + V[b,s,:]).
    Algorithmic bytes: H written once (4*B*S*ns*C), U and V read once (4*B*(N+S)*C), idx read once (8*B*S*ns)."""
    from hit_adv_amd import _lib
    lib = _lib.load()
    U, V = torch.randn(B, N, C, device=dev), torch.randn(B, S, C, device=dev)
    idx = torch.randint(0, N, (B, S, ns), device=dev, dtype=torch.int64)
    H = torch.empty(B, S, ns, C, device=dev)
    us = graph_timed(lambda s: lib.hitadv_group_add_relu_fwd(_p(U), _p(V), _p(idx), B, N, S, ns, C, _p(H), s))
    alg = 4 * B * S * ns * C + 4 * B * (N + S) * C + 8 * B * S * ns
    ach = alg / (us * 1e-6) / 1e9
    traffic, src = _traffic("hitadv::group_add_relu_fwd_k@" + ("cfg4" if "PointNet++" in what else "cfg5"))
    return dict(kernel="group_add_relu_fwd_k (%s: B=%d, N=%d, S=%d, nsample=%d, C=%d)" % (what, B, N, S, ns, C), bound="hbm",
                achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4), traffic=traffic,
                traffic_source=src, us_per_launch=round(us, 2), algorithmic_bytes=alg)


def roofline_group_add_relu_linear(dev, B, N, S, ns, C, what):
    """cfg4: the first two shared layers of a sample-and-group block in one kernel, Y[(b,s,j), :] = relu(relu(U[b, idx[b,s,j], :] +
    V[b,s,:]) W^T + bias) (hitadv_group_add_relu_linear).  Algorithmic bytes: Y written once (4*B*S*ns*C), U and V read once
    (4*B*(N+S)*C), idx read once (8*B*S*ns); the weights (2 pieces x C x C x 2 bytes) are noise."""
    from hit_adv_amd import _lib, ops
    lib = _lib.load()
    U, V = torch.randn(B, N, C, device=dev), torch.randn(B, S, C, device=dev)
    idx = torch.randint(0, N, (B, S, ns), device=dev, dtype=torch.int64)
    W2 = ops.split_weights_f16x2(torch.randn(C, C, device=dev) * 0.1)
    bias = torch.randn(C, device=dev)
    Y = torch.empty(B, S, ns, C, device=dev)
    us = graph_timed(lambda s: lib.hitadv_group_add_relu_linear(_p(U), _p(V), _p(idx), B, N, S, ns, C, _p(W2), _p(bias), C, 1, _p(Y),
                                                                None, s))
    alg = 4 * B * S * ns * C + 4 * B * (N + S) * C + 8 * B * S * ns
    ach = alg / (us * 1e-6) / 1e9
    traffic, src = _traffic("hitadv::rows_linear_gather_k@cfg4")
    return dict(kernel="rows_linear_gather_k<%d, %d> (%s: B=%d, N=%d, S=%d, nsample=%d; hitadv_group_add_relu_linear)" % (C, C, what, B, N, S, ns),
                bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4), traffic=traffic,
                traffic_source=src, us_per_launch=round(us, 2), algorithmic_bytes=alg)


def hot_loop_kernels(dev, B=32, N=1024):
    """Informational: kernels that ARE on cfg2's loop -- the deformation pair and V1 (the victim's 128->1024 shared layer
    fused with the max over points, on the f32 matrix cores)."""
    from hit_adv_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(1)
    C = HP['central_num']
    ori = torch.randn(B, 3, N, generator=g).to(dev)
    central = ori[:, :, :C].contiguous()
    P = (torch.rand(B, C, 3, generator=g) * 0.55).to(dev)
    sig = (0.1 + torch.rand(B, C, generator=g) * 1.1).to(dev)
    up = torch.randn(B, 3, N, generator=g).to(dev)
    adv, inv = torch.empty_like(ori), torch.empty(B, N, device=dev)
    part = torch.empty(lib.hitadv_deform_bwd_scratch_floats(B, N, C), device=dev)
    gp, gs = torch.empty_like(P), torch.empty_like(sig)
    out = {"deform_fwd_us": round(graph_timed(lambda s: lib.hitadv_deform_fwd(_p(ori), _p(central), _p(P), _p(sig), B, N, C,
                                                                               _p(adv), _p(inv), s)), 2),
           "deform_bwd_us": round(graph_timed(lambda s: lib.hitadv_deform_bwd(_p(ori), _p(central), _p(P), _p(sig), _p(adv),
                                                                               _p(inv), _p(up), B, N, C, _p(part), _p(gp),
                                                                               _p(gs), s)), 2),
           "pairs_per_launch": B * N * C}
    h2 = torch.randn(B * N, 128, generator=g).to(dev)
    Wt = (torch.randn(128, 1024, generator=g) * 0.1).to(dev)
    bias = torch.randn(1024, generator=g).to(dev)
    n = lib.hitadv_linear_max_fwd_scratch(B, N, 1024)
    pv, pi = torch.empty(n, device=dev), torch.empty(n, device=dev, dtype=torch.int32)
    mo, mi = torch.empty(B, 1024, device=dev), torch.empty(B, 1024, device=dev, dtype=torch.int64)
    tk = torch.zeros(4096, device=dev, dtype=torch.int32)  # split tickets (self-resetting)
    us = round(graph_timed(lambda st: lib.hitadv_linear_max_fwd(_p(h2), _p(Wt), _p(bias), B, N, 128, 1024, 1, _p(pv), _p(pi),
                                                                _p(mo), _p(mi), _p(tk), st)), 2)
    flops = 2.0 * B * N * 128 * 1024
    out["linear_max_fwd"] = {"bound": "mfma", "us_per_launch": us, "achieved": round(flops / us / 1e6, 1),
                             "peak": F32_MFMA_PEAK, "unit": "TFLOP/s", "frac": round(flops / us / 1e6 / F32_MFMA_PEAK, 4),
                             "dtype": "f32", "flops_per_launch": flops}
    # the form the loop runs by default: both operands as three bf16 pieces, six bf16 MFMAs per 16 values of k in an
    # fp32 accumulator (fp32-accurate; csrc/victim_bf3.hip).  Priced against the SAME 2*B*N*Cin*Cout useful flops; the
    # matrix cores execute 6/8 x 16 = 12x fewer cycles per useful flop than the f32 MFMA form would
    W3 = torch.empty(3, 1024, 128, device=dev, dtype=torch.int16)
    lib.hitadv_split_weights_bf16x3(_p(Wt.t().contiguous()), 1024, 128, _p(W3), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    n3 = lib.hitadv_linear_max_fwd_bf16x3_scratch(B, N, 1024, 0)
    pv3, pi3 = torch.empty(n3, device=dev), torch.empty(n3, device=dev, dtype=torch.int32)
    us3 = round(graph_timed(lambda st: lib.hitadv_linear_max_fwd_bf16x3(_p(h2), _p(W3), _p(bias), B, N, 128, 1024, 1, 0, _p(pv3),
                                                                        _p(pi3), _p(mo), _p(mi), _p(tk), st)), 2)
    out["linear_max_fwd_bf16x3"] = {"bound": "mfma", "us_per_launch": us3, "useful_tflops": round(flops / us3 / 1e6, 1),
                                    "executed_bf16_tflops": round(6 * flops / us3 / 1e6, 1), "peak_bf16": 2500.0,
                                    "frac_of_bf16_peak": round(6 * flops / us3 / 1e6 / 2500.0, 4), "dtype": "3 x bf16 -> f32",
                                    "flops_per_launch": flops}
    # what each form of the layer delivers against float64 on this input (max and rms error of the [B,1024] maxima over the
    # largest of them): the precision claim behind `dtype`
    ref = (h2.double() @ Wt.double()).view(B, N, 1024).max(dim=1).values + bias.double()
    ref = ref.clamp_min(0.)
    top = float(ref.abs().max())

    def err(v):
        e = (v.double() - ref).abs()
        return {"max_over_scale": float(e.max()) / top, "rms_over_scale": float(e.pow(2).mean().sqrt()) / top}
    acc = {"f32_mfma": err(mo.clone())}
    lib.hitadv_linear_max_fwd(_p(h2), _p(Wt), _p(bias), B, N, 128, 1024, 1, _p(pv), _p(pi), _p(mo), _p(mi), _p(tk),
                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    acc["f32_mfma"] = err(mo.clone())
    lib.hitadv_linear_max_fwd_bf16x3(_p(h2), _p(W3), _p(bias), B, N, 128, 1024, 1, 0, _p(pv3), _p(pi3), _p(mo), _p(mi), _p(tk),
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    acc["bf16x3"] = err(mo.clone())
    W2 = torch.empty(2, 1024, 128, device=dev, dtype=torch.int16)
    lib.hitadv_split_weights_f16x2(_p(Wt.t().contiguous()), 1024, 128, _p(W2), None, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    us2 = round(graph_timed(lambda st: lib.hitadv_linear_max_fwd_f16x2(_p(h2), _p(W2), _p(bias), B, N, 128, 1024, 1, 0, _p(pv3),
                                                                       _p(pi3), _p(mo), _p(mi), _p(tk), None, st)), 2)
    lib.hitadv_linear_max_fwd_f16x2(_p(h2), _p(W2), _p(bias), B, N, 128, 1024, 1, 0, _p(pv3), _p(pi3), _p(mo), _p(mi), _p(tk), None,
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    acc["fp16x2"] = err(mo.clone())
    out["accuracy_vs_float64"] = acc
    out["linear_max_fwd_f16x2"] = {"bound": "mfma", "us_per_launch": us2, "useful_tflops": round(flops / us2 / 1e6, 1),
                                   "executed_f16_tflops": round(3 * flops / us2 / 1e6, 1), "peak_f16": 2500.0,
                                   "frac_of_f16_peak": round(3 * flops / us2 / 1e6 / 2500.0, 4), "dtype": "2 x fp16 -> f32",
                                   "flops_per_launch": flops}
    # ... and as the headline's loop launches it: the victim passes of a stack of eight attacks in one call (256 clouds), on 128
    # workgroups (three stacks share the chip) and on the whole chip
    Bs = 8 * B
    hs = torch.randn(Bs * N, 128, generator=g).relu().to(dev)
    hi = hs.half()  # the activation as V2 hands it over: one word per value, fp16 hi | fp16 lo << 16
    hs = ((hi.view(torch.int16).int() & 0xffff) | (((hs - hi.float()) * 2048.).half().view(torch.int16).int() << 16)).contiguous()
    del hi
    mos, mis = torch.empty(Bs, 1024, device=dev), torch.empty(Bs, 1024, device=dev, dtype=torch.int64)
    stack = {"clouds_per_launch": Bs, "flops_per_launch": 8 * flops, "dtype": "2 x fp16 -> f32 (packed pieces from V2)", "bound": "mfma",
             "peak_f16": 2500.0}
    for blocks in (128, 256):
        ns = lib.hitadv_linear_max_fwd_bf16x3_scratch(Bs, N, 1024, blocks)
        pvs, pis = torch.empty(ns, device=dev), torch.empty(ns, device=dev, dtype=torch.int32)
        uss = round(graph_timed(lambda st: lib.hitadv_linear_max_fwd_f16x2_packed(_p(hs), _p(W2), _p(bias), Bs, N, 128, 1024, 1, blocks,
                                                                                  _p(pvs), _p(pis), _p(mos), _p(mis), _p(tk), st)), 2)
        stack["workgroups_%d" % blocks] = {"us_per_launch": uss, "us_per_32_clouds": round(uss / 8, 2),
                                           "executed_f16_tflops": round(3 * 8 * flops / uss / 1e6, 1),
                                           "frac_of_f16_peak": round(3 * 8 * flops / uss / 1e6 / 2500.0, 4),
                                           "frac_of_f16_peak_of_the_cus_used": round(3 * 8 * flops / uss / 1e6 / (2500.0 * blocks / 256), 4)}
    out["linear_max_fwd_f16x2_stack_of_8"] = stack
    return out


def pointnet_forward_flops(B, N, classes=40):
    """Dense flop count of one PointNetFeatureModel forward (model/feature_models.py:71-230): per point the three
    3/64 -> 64 -> 128 -> 1024 stacks and the two learned transforms, per cloud the nine FC layers."""
    per_point = 2 * (3 * 64 + 64 * 128 + 128 * 1024)          # STN3d shared layers
    per_point += 2 * (3 * 3 + 3 * 64)                          # input transform, first encoder layer
    per_point += 2 * (64 * 64 + 64 * 128 + 128 * 1024)         # STNkd shared layers
    per_point += 2 * (64 * 64 + 64 * 128 + 128 * 1024)         # feature transform, encoder tail
    per_cloud = 2 * (1024 * 512 + 512 * 256) * 3 + 2 * 256 * (9 + 4096 + classes)
    return float(B) * (N * per_point + per_cloud)


BF16_MFMA_PEAK = 2500.0  # TFLOP/s, dense bf16 MFMA (MI355X_MICROARCH.md; the headline 5 PF figure includes 2:1 sparsity)


def loop_floor(B, N, classes=40, matrix_mode='bf16x3', C=192):
    """What ONE HiT-ADV iteration on the PointNet engine costs at the chip's three ceilings, counted from what the kernels
    EXECUTE (not from the dense model): the three 128 -> 1024 layers as six bf16 products per useful one (or on the f32
    matrix cores in `f32` mode), the shared-layer chains, the nine FC layers and their nine backward layers on the f32
    matrix cores, and the activations / weights that cross HBM once per iteration.  The input-gradient pass behind the
    max-pools touches ~10 of 64 points per tile: its chain is counted on 32-row blocks (one per tile), its gather as one
    128-wide row per (cloud, channel).  floor = the three times ADDED (no overlap assumed): a bound from below on the
    iteration's duration that no schedule of these kernels can beat by more than the overlap it leaves out."""
    R, tiles = B * N, B * ((N + 63) // 64)
    v1_useful = 3 * 2.0 * R * 128 * 1024
    fwd_chain = 2.0 * R * (64 * 128 + (64 * 64 + 64 * 128) + (64 * 64 + 64 * 128))        # V2: s2 | t1, t2 | h1 @ T64, e2
    fc = 2.0 * B * (3 * (1024 * 512 + 512 * 256) + 256 * (9 + 4096 + classes))             # V4 forward
    bwd_gather = 3 * 2.0 * B * 1024 * 128                                                  # V3: one W3r row per (cloud, channel)
    bwd_chain = 2.0 * tiles * 32 * ((128 * 64) + (128 * 64 + 64 * 64) + (128 * 64 + 64 * 64 + 64 * 64))  # V3 on 32-row blocks
    chains = fwd_chain + bwd_gather + bwd_chain
    f32_flop = 2 * fc + (0.0 if matrix_mode == 'fp16x2' else chains)  # fp16x2 mode: the chains run on the fp16 cores too
    bf16_flop = {'bf16x3': 6, 'fp16x2': 3}.get(matrix_mode, 0) * v1_useful  # 16-bit MFMA flop (bf16 and fp16 share the peak)
    if matrix_mode == 'fp16x2':
        bf16_flop += 3 * chains
    if matrix_mode not in ('bf16x3', 'fp16x2'):
        f32_flop += v1_useful
    act = 4.0 * R * (64 + 128 + 64 + 64 + 128 + 128)          # a1s a2s h1 a1t a2t a2e: written once by V2 ...
    hbm = act + 4.0 * R * 3 * 128                               # ... and the three 128-wide ones read once by V1
    hbm += 2 * 4.0 * (3 * (1024 * 512 + 512 * 256) + 256 * (9 + 4096 + classes))   # FC weights, forward and backward
    hbm += 3 * 2 * 4.0 * 1024 * 128 + 4.0 * B * N * 3 * 4      # 128 -> 1024 weights (as pieces: the same bytes), clouds in / out
    hbm += 4.0 * B * (C * 4 * 4 + tiles // B * 4096 * 2)        # attack parameters + Adam moments, transform-gradient partials
    us = dict(bf16_mfma=bf16_flop / BF16_MFMA_PEAK / 1e6, f32_mfma=f32_flop / F32_MFMA_PEAK / 1e6, hbm=hbm / HBM_PEAK_GBS / 1e3)
    # two readings of the same three times: ADDED (no overlap at all: what a serial schedule of perfect kernels takes) and the
    # MAX of matrix time and HBM time (perfect overlap: below this no schedule can go)
    return dict(executed_bf16_flop=bf16_flop, executed_f32_mfma_flop=f32_flop, hbm_bytes=hbm,
                us=dict({k: round(v, 2) for k, v in us.items()}), loop_floor_us=round(sum(us.values()), 2),
                loop_floor_max_us=round(max(us['bf16_mfma'] + us['f32_mfma'], us['hbm']), 2))


LOOP_TRAFFIC_PROFILE = os.path.join("profiles", "r05_loop_traffic.json")  # named, not globbed: a later file must not change the line silently


def loop_traffic_profiled_offline():
    """HBM bytes of ONE B=32 iteration as the counters saw them in an OFFLINE profile kept under profiles/ (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes over tools/loop_pmc_probe.py -- three stacks of eight attacks, every loop kernel --, corrected per
    MI355X_MICROARCH.md: writes exact, fetches doubled) next to `hbm_bytes`, the model loop_floor() prices.  NOT a measurement of this
    run and not of this build: the counters need the profiler, and the file named here was taken on round 5's morning build, before
    rowmlp_stream_k became the default (profiles/README_r05.md).  Hence the field names, and hence nothing of it in `summary`."""
    path = os.path.join(ROOT, LOOP_TRAFFIC_PROFILE)
    if not os.path.exists(path):
        return dict(hbm_bytes_profiled_offline=None)
    with open(path) as f:
        d = json.load(f)
    per = d.get("per_b32_iteration", {})
    return dict(hbm_bytes_profiled_offline=per.get("hbm_bytes"), hbm_bytes_profiled_offline_fetch_not_doubled=per.get("hbm_bytes_fetch_not_doubled"),
                hbm_profiled_offline_source=LOOP_TRAFFIC_PROFILE,
                hbm_profiled_offline_note="an earlier build's profile (round 5, before the streaming V2), kept for orientation; not this run")


# --------------------------------------------------------------------------------------------- CPU baseline
# The port's inner iteration against the imported reference's, measured in the build container (the reference cannot travel):
# tests/golden/time_reference.py, B = 8, 8 threads, median of 22 iterations each, both warmed up, runs alternated; the two
# execute the same torch ops in the same number (op-for-op profile in DESIGN.md section 7), the ratio is host noise around 1.
PORT_OVER_REFERENCE = dict(ratio=1.01, measured_in="build container, NOT this run (the reference cannot travel to the GPU box)",
                           measured="8 cores: three runs of tests/golden/time_reference.py 8 12 gave 1.12, 0.94, 0.96 (outputs "
                                    "equal to 0.0); the judge's own run in round 4: 0.944", reference_runs_on_gpu_box=False)
def _cores():
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, 32)), avail  # more intra-op threads than this only adds contention for these op sizes


def cpu_baseline_hit_adv(cfg, timed_iters):
    """The CPU oracle (op-for-op restatement of the reference; for PointNet++ the victim samples / groups through
    oracle/victim_geometry.py) on this box's host cores, bounded: setup once + 1 warm-up + `timed_iters` timed inner
    iterations at the config's batch, extrapolated to 10 x 500."""
    from oracle import hitadv_oracle as O
    from oracle import victim_geometry as VG
    cores, avail = _cores()
    torch.set_num_threads(cores)
    B, N = cfg['B'], cfg['N']
    data, _ = synth(0, B, N)
    model = build_victim(cfg)
    if cfg['victim'] in ('pointnet++', 'pct'):
        model = VG.CpuVictim(model)
    with torch.no_grad():
        label = logits_of(model, data[:, :, :3].transpose(1, 2).contiguous()).argmax(1)
    att = O.HiTADVOracle(model, lambda l, t: O.untargeted_logits_adv_loss(l, t, 30.),
                         binary_step=BINARY_STEP, num_iter=NUM_ITER, **HP)
    torch.manual_seed(1)
    t0 = time.perf_counter()
    st = att.prepare(data, label)
    att.begin_step(st)
    t_setup = time.perf_counter() - t0
    att.inner_iteration(st)
    t0 = time.perf_counter()
    for _ in range(timed_iters):
        att.inner_iteration(st)
    t_iter = (time.perf_counter() - t0) / timed_iters
    total = t_setup + t_iter * NUM_ITER * BINARY_STEP
    return dict(value=B / total, unit="clouds/s", cores=cores, host_cpus=avail, kind="port",
                sample="setup (%.1f s) + 1 warm-up + %d timed inner iterations at B=%d (%.2f s/iter), extrapolated to "
                       "%d x %d iterations" % (t_setup, timed_iters, B, t_iter, BINARY_STEP, NUM_ITER),
                s_per_iteration=round(t_iter, 3), port_over_reference=PORT_OVER_REFERENCE)


def cpu_baseline_cw_sweep(cfg):
    """The three oracle attacks of the sweep on the CPU, a few iterations each, extrapolated to their full lengths."""
    from oracle import hitadv_oracle as O
    from oracle import victim_geometry as VG
    cores, avail = _cores()
    torch.set_num_threads(cores)
    B, N = cfg['B'], cfg['N']
    data, _ = synth(0, B, N)
    xyz = data[:, :, :3].contiguous()
    model = VG.CpuVictim(build_victim(cfg))
    torch.manual_seed(2)
    ae = ToyAE().eval()
    with torch.no_grad():
        label = logits_of(model, xyz.transpose(1, 2).contiguous()).argmax(1)
    target = (label + 1) % cfg['classes']
    clip = lambda pc, ori: O.clip_points_linf(pc, ori, 0.18)  # noqa: E731
    n, reps = 3, 3

    def per_iteration(run):
        """(seconds per inner iteration, seconds of a one-iteration call): calls of 1 and 1 + n iterations timed in
        ALTERNATION, `reps` times each, medians taken -- the difference of two single noisy timings can come out negative."""
        short, long_ = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            run(1)
            t1 = time.perf_counter()
            run(1 + n)
            t2 = time.perf_counter()
            short.append(t1 - t0)
            long_.append(t2 - t1)
        one, many = sorted(short)[reps // 2], sorted(long_)[reps // 2]
        if many <= one:
            raise RuntimeError("cpu_baseline: %d iterations (%.3f s) timed no longer than one (%.3f s): the host is too noisy "
                               "to extrapolate from; rerun" % (1 + n, many, one))
        return (many - one) / n, one

    it_adv, f_adv = per_iteration(lambda k: O.cw_family_attack(model, lambda l, t: O.logits_adv_loss(l, t, 0.), clip, xyz, target,
                                                               y_truth=label, ae_model=ae, targeted=True, fresh=True,
                                                               binary_step=1, num_iter=k))
    it_knn, f_knn = per_iteration(lambda k: O.cw_knn_attack(model, lambda l, t: O.logits_adv_loss(l, t, 15.), O.chamfer_knn_dist,
                                                            clip, xyz, target, num_iter=k))
    it_aof, f_aof = per_iteration(lambda k: O.cw_aof_attack(model, lambda l, t: O.untargeted_logits_adv_loss(l, t, 30.), clip,
                                                            xyz, label, binary_step=1, num_iter=k))
    # f_* = one call with a single iteration (setup + 1 iteration + final forward); AdvPC and AOF run two binary steps
    total = (2 * f_adv + 398 * it_adv) + (f_knn + 2499 * it_knn) + (2 * f_aof + 398 * it_aof)
    return dict(value=B / total, unit="clouds/s", cores=cores, host_cpus=avail, kind="port",
                sample="per attack: calls of 1 and %d iterations alternated %d times, medians, at B=%d (AdvPC %.2f, kNN %.2f, AOF "
                       "%.2f s/iter), extrapolated to 2x200 + 2500 + 2x200 iterations" % (1 + n, reps, B, it_adv, it_knn, it_aof),
                s_per_iteration=dict(advpc=round(it_adv, 3), knn=round(it_knn, 3), aof=round(it_aof, 3)))


# --------------------------------------------------------------------------------------------- the timed job
def make_runner(cfg, model, dev, concurrent):
    """Returns (run(batches) -> successes, prewarm(batch), info() -> dict, inner iterations per step)."""
    from hit_adv_amd.util.adv_utils import LogitsAdvLoss, UntargetedLogitsAdvLoss
    if cfg['attack'] == 'hit_adv':
        from hit_adv_amd.ShapeAttack.HiT_ADV import HiT_ADV
        att = HiT_ADV(model, adv_func=UntargetedLogitsAdvLoss(kappa=30.), binary_step=BINARY_STEP, num_iter=NUM_ITER,
                      verbose=False, **HP)

        ran = dict(groups=[])

        def run(todo):
            from hit_adv_amd import groups_in_flight
            ok, i = 0, 0
            ran['groups'] = groups_in_flight(len(todo), att.in_flight(concurrent), stacked=att.stacks())  # shared with eval_ASR
            for n in ran['groups']:
                group = todo[i:i + n]
                res = att.attack_many(group) if len(group) > 1 else [att.attack(*group[0])]
                ok += sum(int(k) for _, k in res)
                i += n
            return ok

        def prewarm_groups(todo):
            """The timed region's OWN group / stack shapes, run once over 1 x 10 iterations on the attacker that is timed: its
            workspaces are keyed by the stack's size, so a warm-up of another size leaves the timed sizes' buffers to be
            allocated and first touched inside the timed region."""
            keep = att.binary_step, att.num_iter
            att.binary_step, att.num_iter = 1, 10
            try:
                run(todo)
            finally:
                att.binary_step, att.num_iter = keep

        def prewarm(batch):  # library handles, lazy initialisation: a 4-iteration attack that is not a step
            short = HiT_ADV(model, adv_func=UntargetedLogitsAdvLoss(kappa=30.), binary_step=1, num_iter=4, verbose=False, **HP)
            short.attack(*batch)

        def profiled(batch):  # a slice of the real job for `top_kernels`: setup + 100 captured iterations
            part = HiT_ADV(model, adv_func=UntargetedLogitsAdvLoss(kappa=30.), binary_step=1, num_iter=100, verbose=False, **HP)
            part.attack(*batch)

        def hit_info():
            stacked = any(isinstance(k[3], str) for k in att._ws)  # attack_many merged the victim passes of its attacks
            return dict(hip_graph=att.last_graph_used, num_iter=NUM_ITER, binary_step=BINARY_STEP, central_num=HP["central_num"],
                        attacks_per_stack=att.attacks_per_stack if stacked else 1, victim_passes_stacked=stacked,
                        groups_of_the_last_run=list(ran['groups']), in_flight=max(ran['groups']) if ran['groups'] else 0)
        prewarm.profiled = profiled
        prewarm.groups = prewarm_groups
        return run, prewarm, hit_info, NUM_ITER * BINARY_STEP

    from hit_adv_amd import CW
    from hit_adv_amd.util.clip_utils import ClipPointsLinf
    from hit_adv_amd.util.dist_utils import ChamferkNNDist, L2Dist
    torch.manual_seed(2)
    ae = ToyAE().eval().to(dev)
    clip = ClipPointsLinf(budget=0.18)
    made = {}

    def attacks(short):
        kw = dict(verbose=False)
        n_adv, n_knn = (2, 20) if short is True else (short if short else (None, None))  # (AdvPC / AOF, kNN) iterations
        a = CW.CWAdvPC(model, ae, LogitsAdvLoss(kappa=0.), L2Dist(), clip_func=clip, **kw,
                       **(dict(binary_step=1, num_iter=n_adv) if short else {}))
        k = CW.CWKNN(model, LogitsAdvLoss(kappa=15.), ChamferkNNDist(), clip, **kw, **(dict(num_iter=n_knn) if short else {}))
        f = CW.CWAOF(model, UntargetedLogitsAdvLoss(kappa=30.), L2Dist(), clip_func=clip, **kw,
                     **(dict(binary_step=1, num_iter=n_adv) if short else {}))
        return a, k, f

    def sweep(batch, short=False):
        data, label = batch
        xyz = data[:, :, :3].contiguous()
        target = (label + 1) % cfg['classes']
        a, k, f = attacks(short)
        calls = [(a, (xyz, target, label)), (k, (xyz, target)), (f, (xyz, label))]
        t = [time.perf_counter()]
        if SEQUENTIAL_SWEEP:  # one attack after the other (round 3's form; A/B switch --sequential-sweep)
            res = []
            for att, args in calls:
                res.append(att.attack(*args))
                torch.cuda.synchronize()
                t.append(time.perf_counter())
        else:  # the three attacks in flight at once, results those of the sequence (CW.attack_concurrently)
            res = CW.attack_concurrently(calls)
            torch.cuda.synchronize()
            t.append(time.perf_counter())
        (_, _, s1), (_, s2), (_, s3) = res
        if not short:
            made['graph'] = k.last_graph_used
            made['graph_advpc'], made['graph_aof'] = a.last_graph_used, f.last_graph_used
            made.setdefault('seconds', []).append([round(t[i + 1] - t[i], 3) for i in range(len(t) - 1)])
        return int(s1) + int(s2) + int(s3)

    def run(todo):
        return sum(sweep(b) for b in todo)

    def info():
        sec = made.get('seconds', [[0, 0, 0]])[-1]
        out = dict(hip_graph_knn=made.get('graph'), hip_graph_advpc=made.get('graph_advpc'), hip_graph_aof=made.get('graph_aof'),
                   attacks=["CWAdvPC 2x200", "CWKNN 2500", "CWAOF 2x200"], in_flight=1 if SEQUENTIAL_SWEEP else 3)
        if SEQUENTIAL_SWEEP:
            out.update(sweep="one attack after the other", seconds_per_attack_last_step=dict(advpc=sec[0], knn=sec[1], aof=sec[2]))
        else:
            out.update(sweep="the three attacks in flight at once on three streams (CW.attack_concurrently: results of the "
                             "sequence)", seconds_last_step=sec[0])
        return out

    def prewarm(batch):
        return sweep(batch, short=True)
    prewarm.profiled = lambda batch: sweep(batch, short=(16, 48))  # a slice of the real sweep for `top_kernels` (captured loops)
    return run, prewarm, info, 2 * 200 + 2500 + 2 * 200


def reduce_over_ranks(elapsed_s, succeeded, attacked, dev, world, collectives):
    """MAX of the elapsed time and SUM of the success counters over the ranks (RCCL; gloo in the CPU test of this
    function).  Returns (elapsed, succeeded, attacked); `collectives` counts the calls a rank makes."""
    elapsed = torch.tensor([elapsed_s], device=dev, dtype=torch.float64)
    counters = torch.tensor([float(succeeded), float(attacked)], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)  # the path's one real collective (ASR aggregation)
        collectives['all_reduce_max'] += 1
        collectives['all_reduce_sum'] += 1
    return elapsed.item(), counters[0].item(), counters[1].item()


def kernel_knobs():
    """Which of the library's either-way kernels this process runs (results never depend on them): so that an A/B line says which side it is."""
    try:
        from hit_adv_amd import _lib
        lib = _lib.load()
        return {"fps_form": int(lib.hitadv_debug_fps_form(-1)), "v1_deferred_search": int(lib.hitadv_debug_v1_defer(-1))}
    except (OSError, AttributeError, RuntimeError):  # (the mock-CPU self-test of the launcher runs without the library)
        return {}


def headline(cfg, steps, warmup, world, elapsed, succeeded, attacked, iters_per_step, in_flight, info, collectives,
             matrix_mode=None, host=None, backend="nccl (RCCL)"):
    """The fields every configuration's line carries (throughput is whole-job: clouds of all ranks / max-over-ranks time)."""
    B, N = cfg['B'], cfg['N']
    clouds = steps * B * world
    return {
        "metric": cfg['metric'], "value": clouds / elapsed, "unit": "clouds/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": {"bf16x3": "f32 (128->1024 layers: 3 bf16 pieces per operand, six exact products, f32 accumulate -- f32-accurate)",
                  "fp16x2": "f32 (128->1024 layers: 2 fp16 pieces per operand, three exact products, f32 accumulate -- error vs "
                            "float64 no larger than the f32-MFMA kernel's: hot_loop_kernels.accuracy_vs_float64)"}.get(matrix_mode, "f32"),
        "data": "synthetic",
        "config": {"workload": cfg['workload'], "batch_per_gpu": B, "num_point": N,
                   **({"matrix_mode": matrix_mode} if matrix_mode else {}), **(host or {}),
                   "parallelism": "independent batch shards, 1 process per GPU",
                   "attacks_in_flight_per_gpu": in_flight,
                   "hip_hardware_queues": HW_QUEUES, "kernel_knobs": kernel_knobs(), **info},
        "cloud_iterations_per_s": clouds * iters_per_step / elapsed,
        "attack_success": {"succeeded": succeeded, "attacked": attacked},
        # what RCCL saw: the calls each rank made in this run (all zero in a single-process run)
        "collectives_per_rank": dict(collectives, world=world, backend=backend if world > 1 else None),
    }


def top_kernels(job, k=3):
    """The k kernels with the most device time while ``job()`` runs (torch.profiler's device activity records; hipGraph
    replays are traced kernel by kernel).  Informational: the committed rocprofv3 summaries under profiles/ are the record."""
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        job()
        torch.cuda.synchronize()
    rows = [(e.key, getattr(e, 'self_device_time_total', None) or getattr(e, 'self_cuda_time_total', 0.), e.count)
            for e in prof.key_averages()]
    rows = [r for r in rows if r[1] > 0]
    total = sum(r[1] for r in rows) or 1.
    rows.sort(key=lambda r: -r[1])
    return [dict(kernel=name.split('(')[0][:96], share=round(t / total, 4), calls=int(c), us_avg=round(t / max(c, 1), 2))
            for name, t, c in rows[:k]]


# (cfg5 inside the default run stays at ONE untimed-warm-up-free sweep: two more would add ~75 s to the driver's bench; `--config cfg5`
# itself now defaults to warmup 1 + 2 timed sweeps -- VERDICT r05 #7 ii)
OTHER_CONFIGS = dict(cfg3=dict(steps=4, warmup=1), cfg4=dict(steps=4, warmup=1), cfg5=dict(steps=1, warmup=0))


def other_configs(timeout_s=240):
    """cfg3 / cfg4 / cfg5 in front of the driver: each as a CHILD process of this same file (a fault in one of them cannot
    take the headline line with it), a short warm-up attack and then the timed steps of its own defaults; what comes back is
    the child's own line, cut down to the measurement, the graph flags, the roofline kernel and the top kernels."""
    import subprocess
    out = {}
    for name, o in OTHER_CONFIGS.items():
        cmd = [sys.executable, os.path.abspath(__file__), "--config", name, "--steps", str(o['steps']), "--warmup", str(o['warmup']),
               "--no-cpu-baseline", "--top-kernels"]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
            lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
            if r.returncode != 0 or not lines:
                out[name] = dict(error="exit code %d" % r.returncode, stderr_tail=r.stderr[-400:])
                continue
            d = json.loads(lines[-1])
            cfgd = d["config"]
            out[name] = dict(metric=d["metric"], value=d["value"], unit=d["unit"], steps=d["steps"], warmup=d["warmup"],
                             ms_per_step=d["ms_per_step"], attacks_in_flight_per_gpu=cfgd.get("attacks_in_flight_per_gpu"),
                             **{k: v for k, v in cfgd.items() if k.startswith("hip_graph") or k in ("seconds_per_attack_last_step", "seconds_last_step", "sweep")},
                             attack_success=d["attack_success"],
                             roofline={k: d["roofline"].get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac")},
                             top_kernels=d.get("top_kernels"), wall_s=round(time.perf_counter() - t0, 1))
        except subprocess.TimeoutExpired:
            out[name] = dict(error="no line within %d s" % timeout_s)
    return out


def visible_gpus():
    """GPUs this process would see, counted WITHOUT starting the HIP runtime (torch.cuda.device_count() goes through
    hipGetDeviceCount on ROCm builds without amdsmi, and the launcher parent must not touch the GPU): KFD topology nodes
    with SIMDs, cut to the list in HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES if one is set.  None if
    the topology cannot be read (then the ranks themselves fail loudly on a missing device)."""
    import glob
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    count = 0
    for path in nodes:
        try:
            with open(path) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            return None
        count += int(props.get("simd_count", "0")) > 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        listed = os.environ.get(var)
        if listed is not None:
            count = min(count, len([v for v in listed.split(",") if v.strip() != ""]))
    return count


def launch_ranks(n, argv, share_gpu=False):
    """``bench.py --gpus N`` without a launcher: start the N ranks ourselves.  Runs BEFORE this process has made any GPU
    call (importing torch is not one): N children of this same file, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment (what ``torch.distributed.run`` would have set); every child's output passes through,
    so rank 0's JSON line is the one line on stdout.  Returns the exit code: non-zero if any rank failed."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(0 if share_gpu else r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HITADV_BENCH_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL needs dmabuf IPC on this pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE,
                                      text=True, bufsize=1))

    def relay(p):  # stdout carries the JSON line and nothing else (gloo, for one, logs its connections to stdout)
        for text in p.stdout:
            out = sys.stdout if text.startswith('{"metric"') else sys.stderr
            out.write(text)
            out.flush()
    import threading
    readers = [threading.Thread(target=relay, args=(p,), daemon=True) for p in procs]
    for t in readers:
        t.start()
    rc = 0
    try:
        for r, p in enumerate(procs):
            code = p.wait()
            if code != 0:
                print("bench.py: rank %d exited with code %d" % (r, code), file=sys.stderr)
                rc = rc or code or 1
                for q in procs:  # the other ranks would wait for it in a collective forever
                    if q.poll() is None:
                        q.terminate()
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
        for t in readers:
            t.join(5)
    return rc


def mock_job(cfg, steps, warmup, world, rank, collectives):
    """``--mock-cpu``: the launcher / barrier / reduction self-test that runs without a GPU (gloo).  NO attack runs: a
    step is a short sleep and the line says so in `data` -- it can never be read as a measurement."""
    def sync():
        if world > 1:
            dist.barrier()
            collectives['barrier'] += 1
    time.sleep(0.01 * warmup)
    sync()
    t0 = time.perf_counter()
    time.sleep(0.01 * steps * (1 + rank))  # ranks differ: the line must carry the slowest one's time
    sync()
    elapsed, succeeded, attacked = reduce_over_ranks(time.perf_counter() - t0, steps, steps * cfg['B'], "cpu", world, collectives)
    if rank == 0:
        line = headline(cfg, steps, warmup, world, elapsed, succeeded, attacked, NUM_ITER * BINARY_STEP, 1, dict(mock=True),
                        collectives, backend="gloo")
        line["data"] = "MOCK: launcher self-test on CPU, no attack ran, value is not a measurement"
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--mock-cpu", action="store_true",
                    help="self-test of the N-rank launcher and the reduction on CPU (gloo); no attack runs, nothing is measured")
    ap.add_argument("--ranks-share-gpu", action="store_true",
                    help="DIAGNOSTIC (no N-GPU node at hand): the N ranks all run on cuda:0 and the two reductions go through gloo; "
                         "everything but RCCL itself is the real N-rank path (launcher, rendezvous, attacks from N host "
                         "processes, barrier, MAX / SUM); the line's `data` says so and `value` is NOT an N-GPU throughput")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="cfg2: skip the short driver-visible runs of cfg3 / cfg4 / cfg5 appended as `other_configs`")
    ap.add_argument("--top-kernels", action="store_true",
                    help="append `top_kernels`: the three kernels with the most device time in one short profiled pass")
    ap.add_argument("--sequential-sweep", action="store_true", help="cfg5: one attack after the other instead of three in flight")
    ap.add_argument("--no-single", action="store_true", help="cfg2: skip the extra one-attack-in-flight measurement")
    ap.add_argument("--no-f32", action="store_true", help="cfg2: skip the extra f32-matrix-mode measurement")
    ap.add_argument("--matrix-mode", choices=["bf16x3", "fp16x2", "f32"], default=None,
                    help="cfg2: how the PointNet engine runs its 128->1024 layers (default: the engine's own default)")
    ap.add_argument("--iters-per-graph", type=int, default=None, help="HiT-ADV iterations recorded per hipGraph")
    ap.add_argument("--concurrent", type=int, default=None,
                    help="independent attack() batches in flight per GPU (separate HIP streams; 1 = strictly serial)")
    args = ap.parse_args()
    global SEQUENTIAL_SWEEP
    SEQUENTIAL_SWEEP = args.sequential_sweep
    cfg = CONFIGS[args.config]
    steps = cfg['steps'] if args.steps is None else args.steps
    warmup = cfg['warmup'] if args.warmup is None else args.warmup
    concurrent = max(1, cfg['concurrent'] if args.concurrent is None else args.concurrent)
    B, N = cfg['B'], cfg['N']

    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # no launcher around us: be the launcher (no GPU call so far)
        seen = None if (args.mock_cpu or args.ranks_share_gpu) else visible_gpus()  # sysfs, not HIP: the parent never starts the runtime
        if seen is not None and seen < args.gpus:
            sys.exit("bench.py: --gpus %d but only %d GPU(s) visible" % (args.gpus, seen))
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], share_gpu=args.ranks_share_gpu))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # the line's n_gpus is the world that ran: never let the two disagree silently
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (python -m torch.distributed.run "
                 "--nproc-per-node %d ... bench.py --gpus %d), or leave WORLD_SIZE unset and bench.py starts them"
                 % (args.gpus, world, args.gpus, args.gpus))
    collectives = dict(barrier=0, all_reduce_max=0, all_reduce_sum=0)
    if args.mock_cpu:
        dev = torch.device("cpu")
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
        return mock_job(cfg, steps, warmup, world, rank, collectives)
    else:
        if args.ranks_share_gpu:
            local = 0
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if args.ranks_share_gpu:  # RCCL refuses two ranks on one device: the reductions' 24 bytes go through the host
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)
        from hit_adv_amd import _lib
        _lib.load()  # fail loudly if the HIP library is missing

    # N ranks x (attacks in flight) Python-driven graph replays share one host: every rank keeps to its share of the cores
    # (intra-op threads only matter for the CPU baseline leg, which runs at N = 1)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    threads = max(1, min(32, avail // max(1, local_world)))
    torch.set_num_threads(threads)
    host = {"host_threads_per_rank": threads, "host_cpus_visible": avail}

    model = build_victim(cfg).to(dev)
    matrix_mode = None
    if cfg['victim'] == 'pointnet':
        from hit_adv_amd.model.pointnet import FoldedPointNet
        if args.matrix_mode is not None:
            FoldedPointNet.matrix_mode = args.matrix_mode
        matrix_mode = FoldedPointNet.matrix_mode
    if args.iters_per_graph is not None:
        from hit_adv_amd.ShapeAttack import HiT_ADV as _H
        _H.HiT_ADV._chunk = lambda self, n=args.iters_per_graph: max(1, min(n, self.num_iter))
    run, prewarm, info, iters_per_step = make_runner(cfg, model, dev, concurrent)
    nbatch = warmup + steps
    extra = 2 if (args.config == 'cfg2' and not args.no_single and world == 1) else 0
    extra_f32 = min(concurrent, 12) if (extra and matrix_mode in ('bf16x3', 'fp16x2') and not args.no_f32) else 0  # one group of twelve
    batches = []
    for s in range(nbatch + extra + extra_f32 + 1):  # every (rank, step) attacks distinct clouds; all resident in HBM up front
        data, _ = synth((rank * (nbatch + extra + extra_f32 + 1) + s) * B, B, N)
        data = data.to(dev)
        with torch.no_grad():  # labels = clean predictions, so every cloud starts correctly classified
            label = logits_of(model, data[:, :, :3].transpose(1, 2).contiguous()).argmax(1)
        batches.append((data, label))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            collectives['barrier'] += 1
        torch.cuda.synchronize()

    torch.manual_seed(1234 + rank)
    single, other_modes, sphere = None, {}, None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        prewarm(batches[-1])
        run(batches[:warmup])
        if hasattr(prewarm, 'groups') and steps > 0:  # the timed region's own group / stack sizes, once, short (not a step)
            prewarm.groups(batches[warmup:nbatch])
        sync()
        t0 = time.perf_counter()
        succ = run(batches[warmup:nbatch])
        sync()
        elapsed = time.perf_counter() - t0
        timed_info = info()  # what the TIMED run did (the informational runs below go through the same runner)
        if extra:  # informational: the same attack with ONE batch in flight (latency of the dependent kernel chain)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for b in batches[nbatch:nbatch + extra]:
                run([b])
            torch.cuda.synchronize()
            single = (time.perf_counter() - t1) / extra
        if extra_f32:  # informational: the SAME group of batches, the same draws, in each form of the 128 -> 1024 layers
            group = batches[nbatch + extra:nbatch + extra + extra_f32]
            prewarm.groups(group)
            for mode in ('fp16x2', 'bf16x3', 'f32'):
                FoldedPointNet.matrix_mode = mode
                torch.manual_seed(4321)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                ok = run(group)
                torch.cuda.synchronize()
                other_modes[mode] = (time.perf_counter() - t1, ok)
            FoldedPointNet.matrix_mode = matrix_mode
        if extra_f32:  # informational: the same job on SURFACE-LIKE clouds (points on a sphere, 1 % noise: a scan's kNN statistics)
            from hit_adv_amd.Dataset.synthetic import sphere_batch
            group = []
            for s_ in range(extra_f32):
                data, _ = sphere_batch(B, N, first=(10 ** 6) + s_ * B)
                data = data.to(dev)
                with torch.no_grad():
                    label = logits_of(model, data[:, :, :3].transpose(1, 2).contiguous()).argmax(1)
                group.append((data, label))
            torch.manual_seed(4321)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ok = run(group)
            torch.cuda.synchronize()
            sphere = (time.perf_counter() - t1, ok)
    per_attack = 3 if cfg['attack'] == 'cw_sweep' else 1
    share = args.ranks_share_gpu and world > 1
    elapsed, succeeded, attacked = reduce_over_ranks(elapsed, succ, steps * B * per_attack, "cpu" if share else dev, world, collectives)

    if rank == 0:
        in_flight = timed_info.pop('in_flight')  # the largest group the runner really kept in flight (not the request)
        line = headline(cfg, steps, warmup, world, elapsed, succeeded, attacked, iters_per_step, in_flight, timed_info,
                        collectives, matrix_mode, host, backend="gloo (--ranks-share-gpu)" if share else "nccl (RCCL)")
        if share:
            line["data"] = ("synthetic; DIAGNOSTIC --ranks-share-gpu: the %d ranks ran on ONE GPU (cuda:0) and reduced through gloo -- "
                            "launcher, rendezvous, per-rank attacks, barrier and MAX / SUM reductions are the N-rank path, RCCL is "
                            "not; `value` is the throughput of one shared GPU, not of %d GPUs" % (world, world))
            line["n_gpus_physical"] = 1
        if args.config == 'cfg2':
            line["roofline"] = roofline_pairwise(dev)
            line["hot_loop_kernels"] = hot_loop_kernels(dev)
            floor = loop_floor(B, N, cfg['classes'], matrix_mode, HP['central_num'])
            eff = elapsed / steps / iters_per_step * 1e6  # wall time per B=32 iteration, all attacks in flight counted
            measured = loop_traffic_profiled_offline()
            line["end_to_end"] = dict(
                floor, us_per_iteration=round(eff, 2), attacks_in_flight=in_flight, **measured,
                frac=round(floor['loop_floor_us'] / eff, 4), frac_of_max_floor=round(floor['loop_floor_max_us'] / eff, 4),
                dense_forward_flops=pointnet_forward_flops(B, N),
                note="loop_floor_us = executed bf16 flop / 2.5 PF + executed f32-MFMA flop / 157.3 TF + HBM bytes / 8 TB/s of "
                     "ONE B=32 iteration (bench.py::loop_floor); us_per_iteration = ms_per_step / 5000 with "
                     "`attacks_in_flight` attacks sharing the GPU; frac = floor / measured; loop_floor_max_us = max(matrix time, HBM "
                     "time) of the same iteration (perfect overlap), frac_of_max_floor = that / measured")
            if measured.get("hbm_bytes_profiled_offline"):
                line["end_to_end"]["hbm_profiled_offline_over_model"] = round(measured["hbm_bytes_profiled_offline"] / floor["hbm_bytes"], 3)
            if single is not None:
                us1 = single / iters_per_step * 1e6
                line["single_attack"] = {"value": B / single, "unit": "clouds/s", "ms_per_step": single * 1e3, "steps": extra,
                                         "attacks_in_flight_per_gpu": 1, "us_per_iteration": round(us1, 2),
                                         "frac_of_loop_floor": round(floor['loop_floor_us'] / us1, 4)}
            for mode, (secs, ok) in other_modes.items():
                us_f = secs / extra_f32 / iters_per_step * 1e6
                line[mode + "_mode"] = {"value": extra_f32 * B / secs, "unit": "clouds/s", "steps": extra_f32,
                                        "attacks_in_flight_per_gpu": extra_f32, "us_per_iteration": round(us_f, 2),
                                        "attack_success": {"succeeded": ok, "attacked": extra_f32 * B},
                                        "loop_floor_us": loop_floor(B, N, cfg['classes'], mode, HP['central_num'])['loop_floor_us'],
                                        "note": "the same job with view.matrix_mode = %r for the three 128 -> 1024 layers: one "
                                                "group of attacks, informational" % mode}
            if sphere is not None:
                line["sphere_inputs"] = {"value": extra_f32 * B / sphere[0], "unit": "clouds/s", "steps": extra_f32, "attacks_in_flight_per_gpu": extra_f32,
                                         "attack_success": {"succeeded": sphere[1], "attacked": extra_f32 * B},
                                         "note": "the same job on surface-like clouds (unit sphere, 1 % radial noise, outward normals)"}
        elif args.config == 'cfg3':
            line["roofline"] = roofline_knn_features(dev)
        elif args.config == 'cfg4':
            line["roofline"] = roofline_group_add_relu_linear(dev, B, N, 512, 32, 64, "PointNet++ sa1")
        else:
            line["roofline"] = roofline_group_add_relu(dev, B, 512, 256, 32, 256, "PCT gather_local_1")
        if args.top_kernels:
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    line["top_kernels"] = top_kernels(lambda: prewarm.profiled(batches[-1]))
                line["top_kernels_note"] = ("torch.profiler over a slice of the job: setup + 100 iterations of one HiT-ADV attack "
                                            "(cfg5: 16 + 48 + 16 iterations of the three attacks), after the timed region")
            except Exception as e:  # noqa: BLE001  (informational: never costs the line)
                line["top_kernels"] = None
                line["top_kernels_note"] = "profiler unavailable: %r" % (e,)
        if args.config == 'cfg2' and world == 1 and not args.no_other_configs:
            line["other_configs"] = other_configs()
        if world == 1 and not args.no_cpu_baseline:
            if cfg['attack'] == 'hit_adv':
                line["cpu_baseline"] = cpu_baseline_hit_adv(cfg, 10 if args.config == 'cfg2' else 2)
            else:
                line["cpu_baseline"] = cpu_baseline_cw_sweep(cfg)
            line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
        # LAST, so that it survives in whatever keeps only the end of the line: the numbers the long fields above carry, compact
        brief = {"value": round(line["value"], 3), "attack_success": "%d/%d" % (succeeded, attacked), "roofline_frac": line["roofline"]["frac"]}
        if "end_to_end" in line:
            brief["loop_frac"] = line["end_to_end"]["frac"]
        if "single_attack" in line:
            brief["single_attack"] = round(line["single_attack"]["value"], 3)
        for mode in ('fp16x2', 'bf16x3', 'f32'):
            m = line.get(mode + "_mode")
            if m:
                brief["%s@%d" % (mode, m["attacks_in_flight_per_gpu"])] = [round(m["value"], 3), "%d/%d" % (
                    m["attack_success"]["succeeded"], m["attack_success"]["attacked"])]
        if "sphere_inputs" in line:
            brief["sphere"] = [round(line["sphere_inputs"]["value"], 3), "%d/%d" % (line["sphere_inputs"]["attack_success"]["succeeded"],
                                                                                    line["sphere_inputs"]["attack_success"]["attacked"])]
        for name, o in (line.get("other_configs") or {}).items():
            brief[name] = ([round(o["value"], 3), "%d/%d" % (o["attack_success"]["succeeded"], o["attack_success"]["attacked"]),
                            o["roofline"]["frac"]] if "value" in o else o.get("error"))
        if "cpu_baseline" in line:
            brief["cpu_port"] = line["cpu_baseline"]["value"]
        line["summary"] = brief
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
