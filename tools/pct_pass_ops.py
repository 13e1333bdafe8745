#!/usr/bin/env python3
"""Which aten ops (and how many kernels) one PCT forward + input-gradient pass at B = 32 consists of, grouped by op name, and the
same for the four offset-attention layers alone.   gpurun -- python tools/pct_pass_ops.py"""
import argparse
import json
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hit_adv_amd.Dataset.synthetic import synth_batch  # noqa: E402
from hit_adv_amd.model.pct import Pct  # noqa: E402


def table(prof, top=28):
    rows = []
    for e in prof.key_averages():  # with CPU activity on: the aten / autograd op a kernel was launched by (self device time)
        t = getattr(e, 'self_device_time_total', None) or getattr(e, 'self_cuda_time_total', 0.)
        if t > 0 and not e.key.startswith(('void ', 'Cijk', 'hitadv::', 'Memset', 'Memcpy')):
            rows.append((e.key[:60], e.count, round(t, 1)))
    rows.sort(key=lambda r: -r[2])
    return dict(total_us=round(sum(r[2] for r in rows), 1), kernels=sum(r[1] for r in rows), top=rows[:top])


def main():
    torch.manual_seed(0)
    m = Pct(argparse.Namespace(dropout=0.2), output_channels=40).eval().cuda()
    data, _ = synth_batch(32, 1024)
    x = data[:, :, :3].transpose(1, 2).contiguous().cuda()

    def whole():
        xi = x.detach().requires_grad_()
        torch.autograd.grad(m(xi).logsumexp(1).sum(), xi)

    h = torch.randn(32, 256, 256, device='cuda')

    def attention_only():
        hi = h.detach().requires_grad_()
        out = m.pt_last.forward_pm(hi)
        torch.autograd.grad(out.sum(), hi)
    res = {}
    for name, job in (('whole pass', whole), ('pt_last (conv1, conv2, four offset-attention layers, cat)', attention_only)):
        job()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            job()
            torch.cuda.synchronize()
        res[name] = table(prof)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
