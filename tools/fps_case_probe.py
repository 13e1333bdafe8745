import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hit_adv_amd import ops
from oracle import c_oracle as N
v = torch.load('tests/golden/_tmp_fps_case.pt')
x = v['xyz'].unsqueeze(0).repeat(64, 1, 1); st = v['start'].reshape(1).repeat(64)
ref = N.fps_from_start(x[:1], 512, st[:1])[0]
bad = 0
for rep in range(50):
    got = ops.fps_from_start(x.cuda(), 512, st.cuda()).cpu()
    bad += int((got != ref).any(dim=1).sum())
print('alone: wrong tables', bad, 'of', 50 * 64)
# beside other work on a second stream
s2 = torch.cuda.Stream(); y = torch.randn(64, 2048, 3).cuda(); bad = 0
for rep in range(50):
    with torch.cuda.stream(s2):
        for _ in range(4):
            ops.fps_from_start(y, 512, st.cuda())
    got = ops.fps_from_start(x.cuda(), 512, st.cuda()).cpu()
    bad += int((got != ref).any(dim=1).sum())
print('beside FPS on another stream: wrong tables', bad, 'of', 50 * 64)
