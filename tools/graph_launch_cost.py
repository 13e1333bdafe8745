#!/usr/bin/env python3
"""What a hipGraph launch of a captured attack iteration costs the HOST against what the iteration costs the GPU (cfg5's
CWKNN on PCT at B = 32): if the two are of the same size, one host thread cannot keep several attacks' streams fed.

    gpurun -- python tools/graph_launch_cost.py
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hit_adv_amd import CW  # noqa: E402
from hit_adv_amd.Dataset.synthetic import synth_batch  # noqa: E402
from hit_adv_amd.model.pct import Pct  # noqa: E402
from hit_adv_amd.util import graph_loop  # noqa: E402
from hit_adv_amd.util.adv_utils import LogitsAdvLoss  # noqa: E402
from hit_adv_amd.util.clip_utils import ClipPointsLinf  # noqa: E402
from hit_adv_amd.util.dist_utils import ChamferkNNDist  # noqa: E402


def main():
    torch.manual_seed(0)
    m = Pct(argparse.Namespace(dropout=0.2), output_channels=40).eval().cuda()
    data, _ = synth_batch(32, 1024)
    xyz = data[:, :, :3].contiguous().cuda()
    with torch.no_grad():
        label = m(xyz.transpose(1, 2).contiguous()).argmax(1)
    host = []
    real = graph_loop.IterationGraph.step

    def timed(self):
        t0 = time.perf_counter()
        real(self)
        host.append(time.perf_counter() - t0)
    graph_loop.IterationGraph.step = timed
    att = CW.CWKNN(m, LogitsAdvLoss(kappa=15.), ChamferkNNDist(), ClipPointsLinf(budget=0.18), num_iter=200, verbose=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    att.attack(xyz, (label + 1) % 40)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    host.sort()
    print(json.dumps(dict(graph=att.last_graph_used, iterations=len(host), host_ms_per_launch_median=round(host[len(host) // 2] * 1e3, 3),
                          host_ms_per_launch_p10=round(host[len(host) // 10] * 1e3, 3), host_s_all_launches=round(sum(host), 3),
                          wall_s_attack=round(wall, 3), wall_ms_per_iteration_upper_bound=round(wall / len(host) * 1e3, 3))))


if __name__ == '__main__':
    main()
