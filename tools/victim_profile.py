import argparse, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import synth_batch
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from victim_breakdown import build
name, k = sys.argv[1], int(sys.argv[2])
m = build(name, k).cuda().eval()
data, _ = synth_batch(32, 1024)
x = data[:, :, :3].transpose(1, 2).contiguous().cuda().requires_grad_()
for _ in range(40):
    o = m(x); lo = o[0] if isinstance(o, tuple) else o
    torch.autograd.grad(lo.sum(), x)
torch.cuda.synchronize()
