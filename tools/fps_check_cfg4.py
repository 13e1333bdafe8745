#!/usr/bin/env python3
"""python tools/fps_check_cfg4.py : cfg4's attack (PointNet++, two attacks in flight) with every FPS table
computed by both sampling kernels; prints how many clouds' tables differed and the success counts."""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fps_check
fps_check.install()
sys.argv = ["bench.py", "--config", "cfg4", "--no-cpu-baseline", "--steps", "2", "--warmup", "0"]
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
except SystemExit:
    pass
import torch
torch.cuda.synchronize()
print("FPS_CHECK", fps_check.counts())
caps = fps_check.captures()
os.makedirs("gpurun_out", exist_ok=True)
torch.save(caps, "gpurun_out/fps_mismatch.pt")
for k, v in caps.items():
    if bool(v['have']):
        d = (v['first'] != v['key64']).nonzero().flatten()
        x = v['xyz']
        print("CAPTURE", k, "first differing sample", int(d[0]), "of", len(v['first']), "lean", v['first'][int(d[0])].item(), "key64", v['key64'][int(d[0])].item(),
              "finite", bool(torch.isfinite(x).all()), "absmax", float(x.abs().max()), "unique points", len(torch.unique(x, dim=0)))
