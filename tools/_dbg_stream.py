import sys, torch
sys.path.insert(0, '/root/repo')
from hit_adv_amd import ops as A
B, Np, mode = 3, 1000, 1
g = torch.Generator().manual_seed(5)
R = B * Np
cu = lambda t: t.cuda()
W2, b2 = cu(torch.randn(64, 128, generator=g) * 0.2), cu(torch.randn(128, generator=g))
T64 = cu(torch.eye(64).repeat(B, 1, 1) + 0.05 * torch.randn(B, 64, 64, generator=g)).contiguous()
hin = cu(torch.randn(R, 64, generator=g).relu())
res = {}
for form in (1, 0):
    A.pointnet_rowmlp_form(form)
    o0, o2 = torch.zeros(R, 64, device='cuda'), torch.zeros(R, 128, device='cuda')
    A.pointnet_rowmlp_fwd(2, B, Np, W2, b2, o2, T=T64, hin=hin, o0=o0, mode=mode)
    torch.cuda.synchronize()
    res[form] = (o0, o2)
d = (res[0][1] != res[1][1])
print("o0 equal", torch.equal(res[0][0], res[1][0]), "o2 mismatches", int(d.sum()), "of", d.numel())
idx = d.nonzero()
print(idx[:20].tolist())
rows = idx[:, 0].unique()
print("rows", rows[:40].tolist(), "rows mod 64", (rows % 64).unique().tolist()[:70], "cols", idx[:, 1].unique().tolist()[:140])
i, j = idx[0].tolist()
print(res[0][1][i, j].item(), res[1][1][i, j].item())
o0 = res[0][0]
rows = idx[:, 0].unique()
print("min |o0| in bad rows:", [float(o0[r].abs().min()) for r in rows[:12]])
print("min |o0| in some good rows:", [float(o0[r].abs().min()) for r in range(0, 12)])
small = (o0.abs() < 6.2e-5).any(dim=1)
print("rows with an fp16-subnormal hi piece:", int(small.sum()), "bad rows among them:", int(small[rows].sum()), "of", len(rows))
tiny = ((o0 - o0.half().float()).abs() * 2048 < 6.2e-5) & (o0 != o0.half().float())
print("rows with an fp16-subnormal LO piece:", int(tiny.any(dim=1).sum()), "bad rows among them:", int(tiny.any(dim=1)[rows].sum()))
# determinism of each form, and which one agrees with the exact arithmetic of the pieces
def run(form):
    A.pointnet_rowmlp_form(form)
    o0, o2 = torch.zeros(R, 64, device='cuda'), torch.zeros(R, 128, device='cuda')
    A.pointnet_rowmlp_fwd(2, B, Np, W2, b2, o2, T=T64, hin=hin, o0=o0, mode=mode)
    torch.cuda.synchronize()
    return o0, o2
a1, a2 = run(1)[1], run(1)[1]
b1, b2_ = run(0)[1], run(0)[1]
print("tile form deterministic:", torch.equal(a1, a2), " stream form deterministic:", torch.equal(b1, b2_))
x = o0.double()
hi = o0.half().double()
lo = ((o0 - o0.half().float()) * 2048).half().double()
w = W2.double()
whi = W2.half().double()
wlo = ((W2 - W2.half().float()) * 2048).half().double()
exact = hi @ whi + (lo @ whi + hi @ wlo) / 2048. + b2.double()
exact = exact.clamp_min(0)
r = int(rows[0])
cols = idx[idx[:, 0] == r][:, 1]
print("row", r, "cols", cols.tolist())
print("exact  ", exact[r, cols][:8].tolist())
print("tile   ", a1[r, cols][:8].double().tolist())
print("stream ", b1[r, cols][:8].double().tolist())
print("max |tile - exact|", float((a1.double() - exact).abs().max()), "max |stream - exact|", float((b1.double() - exact).abs().max()))
print("sum |tile - exact| on mismatches", float((a1.double() - exact)[d].abs().sum()), "stream", float((b1.double() - exact)[d].abs().sum()))
