#!/usr/bin/env python3
"""Time of every fully connected layer shape of the PointNet engine (B = 32), 20 launches per graph, for the split-K
setting given by HITADV_FC_CHUNKS (128-deep chunks per block; unset = the library's choice)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hit_adv_amd import ops  # noqa: E402

SHAPES = [(1024, 512), (512, 256), (256, 4096), (256, 40), (256, 512), (512, 1024), (4096, 256)]


def main():
    g = torch.Generator().manual_seed(0)
    out = {'chunks': os.environ.get('HITADV_FC_CHUNKS', 'default')}
    s = torch.cuda.Stream()
    for K, NOUT in SHAPES:
        x, W, b, m = (torch.randn(*sh, generator=g).cuda() for sh in ((32, K), (K, NOUT), (NOUT,), (32, K)))
        fn = lambda: ops.fc_layer(x, W, b, relu=True, mask=m)  # noqa: E731
        with torch.cuda.stream(s):
            fn()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for _ in range(20):
                    fn()
            gr.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                gr.replay()
            torch.cuda.synchronize()
            out['%dx%d' % (K, NOUT)] = round((time.perf_counter() - t0) / 400 * 1e6, 2)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
