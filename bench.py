#!/usr/bin/env python3
"""Headline benchmark: attacked point-clouds/sec of HiT-ADV (PointNet victim, N=1024, 500 iters).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full ``HiT_ADV.attack()`` over one batch of 32 synthetic clouds (cfg2 of
BASELINE.json: the reference's eval.py hyper-parameters with num_iter=500, binary_step=10, i.e.
5000 inner iterations).  Inputs are resident in HBM before the timed region.  With N > 1 every rank
attacks its own 32 clouds (independent shards, weak scaling, no data-path collective); the only
collectives are the barrier/MAX for timing and one SUM all-reduce of the success counters.

Rank 0 prints ONE JSON line; see DESIGN.md section "Measurement" for the field definitions.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU, NPOINT, NUM_ITER, BINARY_STEP = 32, 1024, 500, 10
HP = dict(attack_lr=1e-2, central_num=192, total_central_num=256, init_weight=10., max_weight=80.,
          cd_weight=1e-4, ker_weight=1., hide_weight=1., curv_loss_knn=16, max_sigm=1.2, min_sigm=0.1,
          budget=0.55)  # eval.py:126-133 / :48-62 defaults
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md, HBM3E spec peak


def synth(first, count):
    from hit_adv_amd.Dataset.synthetic import synth_batch
    return synth_batch(count, NPOINT, first=first)


def victim():
    from hit_adv_amd.model.pointnet import PointNetFeatureModel
    torch.manual_seed(0)
    return PointNetFeatureModel(40, normal_channel=False).eval()


def graph_timed(launch, per_graph=20, reps=50):
    """Average duration (us) of one launch: `per_graph` back-to-back launches captured into a hipGraph (the host's ctypes
    call rate must not open gaps between launches), replayed `reps` times between two events recorded on the stream the
    replays run on, after an untimed pass that lets the clocks settle under this kernel's load."""
    import ctypes
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


def pairwise_roofline(dev):
    """K1: materialising 1024x1024 pairwise kernel at B=32, timed with events on the launch stream."""
    from hit_adv_amd import ops
    x = torch.randn(B_PER_GPU, NPOINT, 3, device=dev)
    y = torch.randn(B_PER_GPU, NPOINT, 3, device=dev)
    for _ in range(20):
        P = ops.pairwise_sqdist(x, y, ops.FORM_GRAM)
    P = torch.empty(B_PER_GPU, NPOINT, NPOINT, device=dev)
    import ctypes
    from hit_adv_amd import _lib
    lib = _lib.load()
    ptrs = (ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(P.data_ptr()))
    us = graph_timed(lambda stream: lib.hitadv_pairwise_sqdist(*ptrs, B_PER_GPU, NPOINT, NPOINT, 3, ops.FORM_GRAM, stream))
    alg_bytes = (4 * NPOINT * NPOINT + 12 * (NPOINT + NPOINT)) * B_PER_GPU  # 4,218,880 B / cloud pair
    achieved = alg_bytes / (us * 1e-6) / 1e9
    # HBM bytes per launch from the committed PMC passes (separate rocprofv3 --pmc runs of tools/kbench.py at
    # the same B/N, corrected per MI355X_MICROARCH.md: WRITE_SIZE exact, FETCH_SIZE x2); null if absent.
    traffic = None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*kbench_traffic.json"))):
        with open(path) as f:
            traffic = json.load(f).get("hitadv::pairwise3_vec4<1>", {}).get("hbm_bytes_per_launch", traffic)
    return dict(kernel="pairwise3_vec4<gram> (hitadv_pairwise_sqdist, B=32, 1024x1024)", bound="hbm",
                achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                traffic=traffic, us_per_launch=round(us, 2), algorithmic_bytes=alg_bytes)


def hot_loop_kernels(dev):
    """Informational: the attack loop's own kernels (deformation fwd/bwd) at cfg2 sizes, C-ABI calls
    timed with events on the launch stream."""
    import ctypes
    from hit_adv_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(1)
    B, N, C = B_PER_GPU, NPOINT, HP['central_num']
    ori = torch.randn(B, 3, N, generator=g).to(dev)
    central = ori[:, :, :C].contiguous()
    P = (torch.rand(B, C, 3, generator=g) * 0.55).to(dev)
    sig = (0.1 + torch.rand(B, C, generator=g) * 1.1).to(dev)
    up = torch.randn(B, 3, N, generator=g).to(dev)
    adv, inv = torch.empty_like(ori), torch.empty(B, N, device=dev)
    part = torch.empty(lib.hitadv_deform_bwd_scratch_floats(B, N, C), device=dev)
    gp, gs = torch.empty_like(P), torch.empty_like(sig)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731

    def timed(fn, reps=200):
        for _ in range(10):
            fn()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0.record()
        for _ in range(reps):
            fn()
        t1.record()
        torch.cuda.synchronize()
        return round(t0.elapsed_time(t1) * 1e3 / reps, 2)

    out = {"deform_fwd_us": timed(lambda: lib.hitadv_deform_fwd(p(ori), p(central), p(P), p(sig), B, N, C, p(adv), p(inv), s)),
           "deform_bwd_us": timed(lambda: lib.hitadv_deform_bwd(p(ori), p(central), p(P), p(sig), p(adv), p(inv), p(up),
                                                                B, N, C, p(part), p(gp), p(gs), s)),
           "pairs_per_launch": B * N * C}
    # the kernel that dominates the loop (37 % of its device time): the victim's 128->1024 shared layer fused with the
    # max over points, on the f32 matrix cores (157.3 TFLOP/s dense f32 MFMA peak, MI355X_MICROARCH.md)
    h2 = torch.randn(B * N, 128, generator=g).to(dev)
    Wt = (torch.randn(128, 1024, generator=g) * 0.1).to(dev)
    bias = torch.randn(1024, generator=g).to(dev)
    n = lib.hitadv_linear_max_fwd_scratch(B, N, 1024)
    pv, pi = torch.empty(n, device=dev), torch.empty(n, device=dev, dtype=torch.int32)
    mo, mi = torch.empty(B, 1024, device=dev), torch.empty(B, 1024, device=dev, dtype=torch.int64)
    tk = torch.zeros(4096, device=dev, dtype=torch.int32)  # split tickets (self-resetting)
    us = round(graph_timed(lambda st: lib.hitadv_linear_max_fwd(p(h2), p(Wt), p(bias), B, N, 128, 1024, 1, p(pv), p(pi), p(mo),
                                                                p(mi), p(tk), st)), 2)
    flops = 2.0 * B * N * 128 * 1024
    out["linear_max_fwd"] = {"bound": "mfma", "us_per_launch": us, "achieved": round(flops / us / 1e6, 1), "peak": 157.3,
                             "unit": "TFLOP/s", "frac": round(flops / us / 1e6 / 157.3, 4), "dtype": "f32",
                             "flops_per_launch": flops}
    return out


def cpu_baseline():
    """The CPU oracle (op-for-op restatement of the reference) on this box's host cores, bounded:
    setup once + 1 warm-up + 2 timed inner iterations at B=32, extrapolated to 10 x 500."""
    from oracle import hitadv_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 32))  # more intra-op threads than this only adds contention for these op sizes
    torch.set_num_threads(cores)
    data, label = synth(0, B_PER_GPU)
    model = victim()
    att = O.HiTADVOracle(model, lambda l, t: O.untargeted_logits_adv_loss(l, t, 30.),
                         binary_step=BINARY_STEP, num_iter=NUM_ITER, **HP)
    torch.manual_seed(1)
    t0 = time.perf_counter()
    st = att.prepare(data, label)
    att.begin_step(st)
    t_setup = time.perf_counter() - t0
    att.inner_iteration(st)
    t0 = time.perf_counter()
    n_timed = 2
    for _ in range(n_timed):
        att.inner_iteration(st)
    t_iter = (time.perf_counter() - t0) / n_timed
    total = t_setup + t_iter * NUM_ITER * BINARY_STEP
    return dict(value=B_PER_GPU / total, unit="clouds/s", cores=cores, host_cpus=avail, kind="port",
                sample="setup (%.1f s) + 1 warm-up + %d timed inner iterations at B=32 (%.2f s/iter), "
                       "extrapolated to %d x %d iterations" % (t_setup, n_timed, t_iter, BINARY_STEP, NUM_ITER),
                s_per_iteration=round(t_iter, 3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--concurrent", type=int, default=3,
                    help="independent attack() batches in flight per GPU (separate HIP streams; 1 = strictly serial)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from hit_adv_amd import _lib
    from hit_adv_amd.ShapeAttack.HiT_ADV import HiT_ADV
    from hit_adv_amd.util.adv_utils import UntargetedLogitsAdvLoss
    _lib.load()  # fail loudly if the HIP library is missing

    model = victim().to(dev)
    att = HiT_ADV(model, adv_func=UntargetedLogitsAdvLoss(kappa=30.), binary_step=BINARY_STEP,
                  num_iter=NUM_ITER, verbose=False, **HP)
    nbatch = args.warmup + args.steps
    batches = []
    for s in range(nbatch):  # every (rank, step) attacks distinct clouds; all resident in HBM up front
        data, _ = synth((rank * nbatch + s) * B_PER_GPU, B_PER_GPU)
        data = data.to(dev)
        with torch.no_grad():  # labels = clean predictions, so every cloud starts correctly classified
            label = model(data[:, :, :3].transpose(1, 2).contiguous())[0].argmax(1)
        batches.append((data, label))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(todo):
        """Attack the given steps, `--concurrent` at a time (each step keeps single-call semantics)."""
        ok = 0
        for i in range(0, len(todo), max(1, args.concurrent)):
            group = todo[i:i + max(1, args.concurrent)]
            res = att.attack_many(group) if len(group) > 1 else [att.attack(*group[0])]
            ok += sum(int(n) for _, n in res)
        return ok

    torch.manual_seed(1234 + rank)
    run(batches[:args.warmup])
    sync()
    t0 = time.perf_counter()
    succ = run(batches[args.warmup:])
    sync()
    elapsed = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    counters = torch.tensor([float(succ), float(args.steps * B_PER_GPU)], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)  # the path's one real collective (ASR aggregation)
    elapsed = elapsed.item()

    if rank == 0:
        clouds = args.steps * B_PER_GPU * world
        line = {
            "metric": "attacked point-clouds/sec (HiT-ADV, PointNet, N=1024, 500 iters)",
            "value": clouds / elapsed, "unit": "clouds/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg2: synthetic ModelNet40-shaped clouds, 1024 pts, batch 32 per GPU, "
                                   "PointNet victim (random init, eval mode), HiT-ADV eval.py hyper-parameters, "
                                   "num_iter=500 x binary_step=10 = 5000 inner iterations per attack()",
                       "batch_per_gpu": B_PER_GPU, "num_point": NPOINT, "num_iter": NUM_ITER,
                       "binary_step": BINARY_STEP, "central_num": HP["central_num"],
                       "parallelism": "independent batch shards, 1 process per GPU", "hip_graph": att.last_graph_used,
                       "attacks_in_flight_per_gpu": min(max(1, args.concurrent), args.steps)},
            "cloud_iterations_per_s": clouds * NUM_ITER * BINARY_STEP / elapsed,
            "attack_success": {"succeeded": counters[0].item(), "attacked": counters[1].item()},
            "roofline": pairwise_roofline(dev),
            "hot_loop_kernels": hot_loop_kernels(dev),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
