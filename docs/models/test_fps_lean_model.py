"""A numpy model of fps_lean's step (hit_adv_amd/csrc/sampling.hip) against the oracle: the point -> (wave, lane, u) layout
k = thread + threads * u, running distances compared as BIT PATTERNS, the wave's winner found as "lowest u that holds the wave maximum,
lowest lane of that u", the waves' (bits, ~k) keys joined by an unsigned maximum, points past N holding 0.  What the GPU tests check on
the device, checked here on the host for the selection logic alone: lowest index on every tie, whatever the layout."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as N


def lean_model(x, m, start, NW, PT):
    n = x.shape[0]
    TH = 64 * NW
    assert n <= TH * PT
    k = (np.arange(TH)[:, None] + TH * np.arange(PT)[None, :])            # [thread, u]
    inside = k < n
    kk = np.where(inside, k, 0)
    p = x[kk]                                                              # [thread, u, 3]
    run = np.where(inside, np.float32(1e10).view(np.uint32), np.uint32(0)).astype(np.uint32)
    far, out = int(start), []
    for _ in range(m):
        out.append(far)
        c = x[far]
        d = p - c
        dd = ((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]).astype(np.float32)
        run = np.minimum(run, dd.view(np.uint32))                          # unsigned order = float order for values >= +0
        best = 0
        for w in range(NW):
            r = run[64 * w:64 * w + 64]                                    # [lane, u]
            M = r.max()
            u = int(np.nonzero((r >= M).any(axis=0))[0][0])                # ballots from u = PT - 1 down to 0 leave the lowest u
            lane = int(np.nonzero(r[:, u] >= M)[0][0])                     # ... and the lowest lane that holds it
            key = (int(M) << 32) | (0xFFFFFFFF - (TH * u + 64 * w + lane))
            best = max(best, key)                                          # ds_max_u64
        far = 0xFFFFFFFF - (best & 0xFFFFFFFF)
    return np.array(out)


@pytest.mark.parametrize("n,NW,PT", [(300, 4, 2), (512, 4, 2), (700, 8, 2), (1024, 8, 2), (1500, 8, 4), (2048, 8, 4), (3000, 8, 8)])
def test_model_of_the_lean_step_gives_the_oracle_table(n, NW, PT):
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, 3, generator=g)
    lattice = (torch.randint(-3, 4, (n, 3), generator=g).float() * 0.25)   # exact ties, repeated points
    for cloud, m in ((x, 40), (lattice, min(n, 120))):
        start = n // 3
        want = N.fps_from_start(cloud.unsqueeze(0), m, torch.tensor([start]))[0].numpy()
        got = lean_model(cloud.numpy(), m, start, NW, PT)
        assert np.array_equal(got, want)
