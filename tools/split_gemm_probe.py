#!/usr/bin/env python3
"""Probe: an fp32-accurate GEMM out of fp16 library GEMMs (two fp16 pieces per operand, three exact products), against the
f32 GEMM torch dispatches to hipBLASLt.  Prints one JSON line: times and errors against float64 per shape."""
import json
import sys
import time

import torch


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def split(a):
    a1 = a.half()
    a2 = ((a - a1.float()) * 2048.0).half()
    return a1, a2


def main():
    out = {"torch": torch.__version__}
    g = torch.Generator(device='cuda').manual_seed(0)
    for M, K, N in ((32768, 512, 1024), (32768, 1024, 512), (1048576, 64, 64), (524288, 128, 128), (32768, 256, 256), (131072, 64, 128)):
        a = torch.randn(M, K, device='cuda', generator=g)
        b = torch.randn(K, N, device='cuda', generator=g) / K ** 0.5
        row = {}
        row['f32_us'] = round(timed(lambda: torch.mm(a, b)), 1)
        try:
            a1, a2 = split(a)
            b1, b2 = split(b)
            row['f16_single_us'] = round(timed(lambda: torch.mm(a1, b1, out_dtype=torch.float32)), 1)
            a12 = torch.cat([a1, a2], 1).contiguous()
            b21 = torch.cat([b2, b1], 0).contiguous()

            def three():
                hi = torch.mm(a1, b1, out_dtype=torch.float32)
                return torch.addmm(hi, a12, b21, beta=1.0, alpha=1.0 / 2048.0, out_dtype=torch.float32)
            row['split_gemms_us'] = round(timed(three), 1)

            def full():
                x1 = a.half()
                x12 = torch.cat([x1, ((a - x1.float()) * 2048.0).half()], 1)
                hi = torch.mm(x12[:, :K], b1, out_dtype=torch.float32)
                return torch.addmm(hi, x12, b21, beta=1.0, alpha=1.0 / 2048.0, out_dtype=torch.float32)
            row['split_with_input_split_us'] = round(timed(full), 1)
            ms = min(M, 4096)
            ref = a[:ms].double() @ b.double()
            scale = float(ref.abs().max())
            row['err_f32_over_scale'] = float((torch.mm(a, b)[:ms].double() - ref).abs().max()) / scale
            row['err_split_over_scale'] = float((three()[:ms].double() - ref).abs().max()) / scale
            row['err_f16_over_scale'] = float((torch.mm(a1, b1, out_dtype=torch.float32)[:ms].double() - ref).abs().max()) / scale
        except Exception as e:  # noqa: BLE001
            row['error'] = repr(e)[:300]
        row['f32_tflops'] = round(2.0 * M * K * N / row['f32_us'] / 1e6, 1)
        if 'split_gemms_us' in row:
            row['split_useful_tflops'] = round(2.0 * M * K * N / row['split_gemms_us'] / 1e6, 1)
        out['%dx%dx%d' % (M, K, N)] = row
        del a, b
    print(json.dumps(out))


if __name__ == '__main__':
    sys.exit(main())
