#!/usr/bin/env python3
"""Only the materialising pairwise kernel K1 (and the fused minima K2) at the bench sizes -- for the two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE) behind `roofline.traffic` (tools/r02_measure.sh, tools/pmc_summary.py)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hit_adv_amd import _lib  # noqa: E402

lib = _lib.load()
B, N = 32, 1024
g = torch.Generator().manual_seed(0)
x, y = torch.randn(B, N, 3, generator=g).cuda(), torch.randn(B, N, 3, generator=g).cuda()
P = torch.empty(B, N, N, device='cuda')
mx, my = torch.empty(B, N, device='cuda'), torch.empty(B, N, device='cuda')
ax, ay = torch.empty(B, N, device='cuda', dtype=torch.int32), torch.empty(B, N, device='cuda', dtype=torch.int32)
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
for _ in range(10):
    for form in (0, 1):
        lib.hitadv_pairwise_sqdist(p(x), p(y), p(P), B, N, N, 3, form, s)
        lib.hitadv_nn_min(p(x), p(y), B, N, N, 3, form, p(mx), p(ax), p(my), p(ay), None, s)
torch.cuda.synchronize()
