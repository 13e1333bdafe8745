#!/usr/bin/env python3
"""The victim's 128 -> 1024 layer + max over the points: the full fp16x2 evaluation (V1) against the filtered form (V1F,
csrc/experimental/victim_filter.hip) at the stacked launch size, on the ENGINE'S OWN activations (a seeded PointNet on synthetic clouds, the
three layers s3 / t3 / e3) and with last call's winners as seeds (the steady state of the attack loop) or none (its first
iteration).  Prints JSON: us per call, candidates per (cloud, channel).     python tools/v1_filter_probe.py [B] [blocks]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hit_adv_amd import _lib, ops  # noqa: E402

lib = _lib.load()
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import v1_filter  # noqa: E402  (libhitadv_experimental.so: make -C hit_adv_amd/csrc experimental)
xlib = v1_filter.load()
p = bench._p
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 128
N, dev = 1024, torch.device('cuda', 0)
cfg = bench.CONFIGS['cfg2']
model = bench.build_victim(cfg).to(dev)
view = model.attack_view()
data, _ = bench.synth(0, B, N)
x = data[:, :, :3].transpose(1, 2).contiguous().to(dev)
R = B * N
out = {"B": B, "blocks": blocks}
# the three 128-wide activations as the engine produces them (packed pieces)
a1, a2 = torch.empty(R, 64, device=dev), torch.empty(R, 128, device=dev)
ops.pointnet_rowmlp_fwd(0, B, N, view.s2_w, view.s2_b, a2, x=x, W0=view.s1_w, b0=view.s1_b, o0=a1, mode=2, range_flag=view.range_flag)
layers = {"s3": (a2, view.pieces('s3', 2), view.s3_b)}
for name, (act, W2, bias) in layers.items():
    wn = v1_filter.weight_row_norms(getattr(view, name + '_wr'))  # [Cout, Cin]: the fp32 weights behind the pieces
    n = lib.hitadv_linear_max_fwd_bf16x3_scratch(B, N, 1024, blocks)
    pv, pi = torch.empty(n, device=dev), torch.empty(n, device=dev, dtype=torch.int32)
    mo, mi = torch.empty(B, 1024, device=dev), torch.empty(B, 1024, device=dev, dtype=torch.int64)
    tk = torch.zeros(4096, device=dev, dtype=torch.int32)
    full = bench.graph_timed(lambda s: lib.hitadv_linear_max_fwd_f16x2_packed(p(act), p(W2), p(bias), B, N, 128, 1024, 1, blocks, p(pv), p(pi), p(mo), p(mi), p(tk), s))
    words = xlib.hitadv_linear_max_filter_scratch_words(B, 1024)
    scratch = torch.zeros(words, device=dev, dtype=torch.int32)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    fo, fi = torch.empty(B, 1024, device=dev), torch.empty(B, 1024, device=dev, dtype=torch.int64)
    seed = torch.zeros(B, 1024, device=dev, dtype=torch.int64)
    call = lambda s: xlib.hitadv_linear_max_fwd_f16x2_filtered(p(act), p(W2), p(wn), p(bias), B, N, 128, 1024, 1, blocks, p(seed), p(scratch), p(fo), p(fi), p(flag), s)  # noqa: E731
    call(None)
    torch.cuda.synchronize()
    first = scratch[B * 1024:B * 1024 + B * 32].float()
    steady_us = bench.graph_timed(call)  # seeds = the winners from now on
    steady = scratch[B * 1024:B * 1024 + B * 32].float()
    seed.zero_()
    cold_us = bench.graph_timed(lambda s: (seed.zero_(), call(s)), per_graph=4, reps=10)
    out[name] = dict(full_us=round(full, 1), filtered_steady_us=round(steady_us, 1), filtered_no_seed_us=round(cold_us, 1),
                     candidates_per_channel_first_call=round(float(first.mean()) / 32, 2), candidates_per_channel_steady=round(float(steady.mean()) / 32, 2),
                     longest_list_steady=int(steady.max()), longest_list_first=int(first.max()), overflow=int(flag.item()),
                     same_maxima=bool((fo - mo).abs().max() <= 2e-6 * mo.abs().max()), argmax_differs=int((fi != mi).sum()))
print(json.dumps(out, indent=1))
