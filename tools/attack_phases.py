#!/usr/bin/env python3
"""Wall time of the phases of one HiT_ADV.attack() call at the bench configuration (host-synchronised)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from hit_adv_amd.ShapeAttack.HiT_ADV import HiT_ADV  # noqa: E402
from hit_adv_amd.util.adv_utils import UntargetedLogitsAdvLoss  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    cfg = bench.CONFIGS['cfg2']
    model = bench.build_victim(cfg).to(dev)
    att = HiT_ADV(model, adv_func=UntargetedLogitsAdvLoss(kappa=30.), binary_step=bench.BINARY_STEP,
                  num_iter=bench.NUM_ITER, verbose=False, **bench.HP)
    data, _ = bench.synth(0, cfg['B'], cfg['N'])
    data = data.to(dev)
    with torch.no_grad():
        label = model(data[:, :, :3].transpose(1, 2).contiguous())[0].argmax(1)
    att.attack(data, label)  # first call: library handles, caches
    for rep in range(2):
        def tick():
            torch.cuda.synchronize()
            return time.perf_counter()
        t0 = tick()
        att._view.refresh(att.model)
        ws = att._setup(data, label)
        t1 = tick()
        att._prepare_graphs([ws])
        t2 = tick()
        att._reset_search(ws)
        ws.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ws.stream):
            for b in range(att.binary_step):
                att._run_step(ws, b, False)
        torch.cuda.current_stream().wait_stream(ws.stream)
        t3 = tick()
        att._finish(ws, False)
        t4 = tick()
        print("setup %.1f ms | warm-up + capture %.1f ms | %d replays %.1f ms (%.3f ms/iter) | finish %.1f ms" %
              ((t1 - t0) * 1e3, (t2 - t1) * 1e3, att.binary_step * att.num_iter, (t3 - t2) * 1e3,
               (t3 - t2) * 1e3 / (att.binary_step * att.num_iter), (t4 - t3) * 1e3))


if __name__ == '__main__':
    main()
