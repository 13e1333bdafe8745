#!/usr/bin/env python3
"""Ablation timing of the filtered V1 (csrc/experimental/victim_filter.hip) on synthetic rows: the kernel with one cost removed
(hitadv_debug_vf_ablate: 1 no candidate bookkeeping, 2 no MFMAs, 3 no row norms, 4 no detection); results of 1-4 are garbage."""
import sys, os, torch
sys.path.insert(0, '/root/repo')
import bench
from hit_adv_amd import _lib, ops
lib = _lib.load()
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import v1_filter  # noqa: E402  (libhitadv_experimental.so: make -C hit_adv_amd/csrc experimental)
xlib = v1_filter.load(); p = bench._p
B, N, blocks, dev = 256, 1024, 128, torch.device('cuda', 0)
g = torch.Generator().manual_seed(0)
h = (torch.randn(B * N, 128, generator=g).relu() * torch.rand(B * N, 1, generator=g)).to(dev)
hi = h.half(); lo = ((h - hi.float()) * 2048.).half()
xp = ((hi.view(torch.int16).int() & 0xffff) | (lo.view(torch.int16).int() << 16)).contiguous()
W = (torch.randn(1024, 128, generator=g) * 0.1).to(dev)
W2, wn = ops.split_weights_f16x2(W), v1_filter.weight_row_norms(W)
bias = torch.zeros(1024, device=dev)
scratch = torch.zeros(xlib.hitadv_linear_max_filter_scratch_words(B, 1024), device=dev, dtype=torch.int32)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
fo, fi = torch.empty(B, 1024, device=dev), torch.empty(B, 1024, device=dev, dtype=torch.int64)
seed = torch.zeros(B, 1024, device=dev, dtype=torch.int64)
call = lambda s: xlib.hitadv_linear_max_fwd_f16x2_filtered(p(xp), p(W2), p(wn), p(bias), B, N, 128, 1024, 1, blocks, p(seed), p(scratch), p(fo), p(fi), p(flag), s)
call(None); call(None); torch.cuda.synchronize()
good = seed.clone()
for abl in (0, 5, 1, 4):
    xlib.hitadv_debug_vf_ablate(abl)
    seed.copy_(good)
    us = bench.graph_timed(lambda s: (call(s), lib.hitadv_copy(p(good), p(seed), B * 1024 * 8, s) if hasattr(lib, 'hitadv_copy') and False else None))
    print("ablate", abl, "seed+stream+refine us", round(us, 1))
xlib.hitadv_debug_vf_ablate(0)
