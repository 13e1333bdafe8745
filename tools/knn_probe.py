#!/usr/bin/env python3
"""Only the kNN kernel at the bench sizes, a few launches per K -- meant for rocprofv3 --pmc passes
(gpurun -- rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES ... --output-format csv -d gpurun_out/knn_pmc -- python3 tools/knn_probe.py)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hit_adv_amd import _lib  # noqa: E402

lib = _lib.load()
B, N = 32, 1024
x = torch.randn(B, N, 3, generator=torch.Generator().manual_seed(0)).cuda()
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
for K in (1, 6, 17):
    d = torch.empty(B, N, K, device='cuda')
    ix = torch.empty(B, N, K, device='cuda', dtype=torch.int64)
    for _ in range(5):
        lib.hitadv_knn_points(p(x), p(x), B, N, N, K, 0, p(d), p(ix), 1, s)
torch.cuda.synchronize()
