#!/usr/bin/env python3
"""Where the time of gemm_f16x2_k goes: the kernel rebuilt with -DHITADV_G16_TUNE, timed with one cost removed at a time
(0 as shipped, 1 no fp16 conversions in the stash, 2 no global loads after the first K step, 3 no LDS reads / MFMAs, 4 one
product instead of three; 10-15: gemm_f16x2_ring_k as shipped, no MFMAs, no split, the DMA ring alone, no DMA after the
prologue, one product).  Builds /tmp/libg16.so with hipcc on the box; prints one JSON line."""
import ctypes
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(ROOT, 'hit_adv_amd', 'csrc', 'gemm16.hip')
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-slp-vectorize',
                       '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.dirname(src),
                       '-DHITADV_G16_TUNE', '-shared', src, '-o', '/tmp/libg16.so'])
lib = ctypes.CDLL('/tmp/libg16.so')
P = ctypes.c_void_p
lib.hitadv_gemm_f16x2_ablate.argtypes = [ctypes.c_int, P, P, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, P, P]
lib.hitadv_split_rows_f16x2.argtypes = [P, ctypes.c_int, ctypes.c_int, P, P, P]
out = {}
for M, K, N in ((32768, 512, 1024), (32768, 1024, 512)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    Wp = torch.empty(2, N, K, dtype=torch.int16, device='cuda')
    C = torch.empty(M, N, device='cuda')
    s = P(torch.cuda.current_stream().cuda_stream)
    lib.hitadv_split_rows_f16x2(P(W.data_ptr()), N, K, P(Wp.data_ptr()), None, s)
    row = {}
    for abl in (0, 1, 2, 3, 4, 0, 10, 11, 12, 13, 14, 15, 10):
        for _ in range(3):
            lib.hitadv_gemm_f16x2_ablate(abl, P(x.data_ptr()), P(Wp.data_ptr()), M, N, K, P(C.data_ptr()), s)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(20):
            lib.hitadv_gemm_f16x2_ablate(abl, P(x.data_ptr()), P(Wp.data_ptr()), M, N, K, P(C.data_ptr()), s)
        t1.record()
        torch.cuda.synchronize()
        row['abl%d%s' % (abl, '_again' if 'abl%d' % abl in row else '')] = round(t0.elapsed_time(t1) * 50, 1)
    out['%dx%dx%d' % (M, K, N)] = row
print(json.dumps(out))
