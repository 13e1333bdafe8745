#!/usr/bin/env python3
"""Does farthest point sampling give the same table when other launches share its CUs?  Four streams, each sampling its own
batch over and over (64 clouds of 2048 -> 512, then the sampled 512 -> 128: PointNet++'s two levels; 32 clouds of 1024 -> 512
-> 256 with PCT's sampler), with and without a GEMM between the launches; every table against the one computed alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hit_adv_amd import ops
torch.manual_seed(0)
streams = [torch.cuda.Stream() for _ in range(4)]
flag = torch.zeros(1, dtype=torch.int32, device='cuda')
a = torch.randn(8192, 512).cuda(); Wp = ops.split_rows_f16x2(torch.randn(512, 512).cuda(), flag)
res = {}
for name, B, n1, m1, m2, pct in (('pointnet++', 64, 2048, 512, 128, False), ('pct', 32, 1024, 512, 256, True)):
    xs = [torch.randn(B, n1, 3).cuda() for _ in streams]
    st1 = [torch.randint(0, n1, (B,)).cuda() for _ in streams]
    st2 = [torch.randint(0, m1, (B,)).cuda() for _ in streams]
    def levels(i):
        f = (lambda x, m, s: ops.fps_pct(x, m, s, reference=True)) if pct else ops.fps_from_start
        i1 = f(xs[i], m1, st1[i])
        x2 = torch.gather(xs[i], 1, i1.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        return i1, f(x2, m2, st2[i])
    refs = [tuple(t.clone() for t in levels(i)) for i in range(4)]
    torch.cuda.synchronize()
    for gemm in (False, True):
        bad = 0
        for rep in range(30):
            outs = []
            for i, s in enumerate(streams):
                with torch.cuda.stream(s):
                    for _ in range(3):
                        if gemm:
                            ops.gemm_f16x2(a, Wp, range_flag=flag)
                        outs.append((i, levels(i)))
            torch.cuda.synchronize()
            bad += sum(int(not (torch.equal(o[0], refs[i][0]) and torch.equal(o[1], refs[i][1]))) for i, o in outs)
        res['%s%s' % (name, '+gemm' if gemm else '')] = '%d of %d' % (bad, 30 * 12)
print(res)
