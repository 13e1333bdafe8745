#!/usr/bin/env python3
"""A few launches of the plain fp16x2 GEMM (32768 x 512 x 1024), ring kernel then staged kernel, for `rocprofv3 --pmc` passes
(tools/r05_measure.sh pmc_g16)."""
import ctypes
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hit_adv_amd import _lib, ops

lib = _lib.load()
P = ctypes.c_void_p
M, K, N = 32768, 512, 1024
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g).cuda()
W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
flag = torch.zeros(1, dtype=torch.int32, device='cuda')
Wp = ops.split_rows_f16x2(W, flag)
C = torch.empty(M, N, device='cuda')
s = P(torch.cuda.current_stream().cuda_stream)
for ring in (1, 0):
    lib.hitadv_debug_g16_ring(ring)
    for _ in range(6):
        lib.hitadv_gemm_f16x2(P(x.data_ptr()), None, P(Wp.data_ptr()), None, M, N, K, 0, P(C.data_ptr()), P(flag.data_ptr()), s)
    torch.cuda.synchronize()
print('done')
