#!/usr/bin/env python3
"""Only `group_add_relu_fwd_k` at the shape one configuration's roofline names (cfg4: PointNet++ sa1, cfg5: PCT gather_local_1)
-- for the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) behind that configuration's `roofline.traffic`
(tools/r04_measure.sh pmc_gar, tools/pmc_summary.py).   python tools/gar_probe.py cfg4|cfg5"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hit_adv_amd import _lib  # noqa: E402

SHAPES = dict(cfg4=(64, 2048, 512, 32, 64), cfg5=(32, 512, 256, 32, 256))  # B, N, S, nsample, C (bench.py: roofline_group_add_relu)
B, N, S, ns, C = SHAPES[sys.argv[1]]
lib = _lib.load()
g = torch.Generator().manual_seed(0)
U, V = torch.randn(B, N, C, generator=g).cuda(), torch.randn(B, S, C, generator=g).cuda()
idx = torch.randint(0, N, (B, S, ns), generator=g).cuda()
H = torch.empty(B, S, ns, C, device='cuda')
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
for _ in range(10):
    lib.hitadv_group_add_relu_fwd(p(U), p(V), p(idx), B, N, S, ns, C, p(H), s)
torch.cuda.synchronize()
