"""Every index kernel north_star's bit-exact bar is about, under CONCURRENCY (VERDICT r05 next-round #1c).

Round 5 showed that single-stream bit-exact tests are not enough: a build of fps_lean passed every one of them, and 1,400 launches on
four streams, and still returned wrong tables while other kernels shared the GPU (docs/kernels/round5.md section 8).  Only FPS got a
cross-check then.  Here each selection kernel -- knn_points (direct form, K = 17 and 6: what HiT_ADV.py:78-80,320-336 calls),
nn_min (Chamfer / Hausdorff), the CUDA extension's ball_query and furthest_point_sampling, the victims' query_ball_point, both
fps_from_start kernels and PCT's sampler -- is launched >= 1,000 times round-robin on three streams while two more streams keep a GEMM
(matrix pipe + HBM) and an LDS-heavy kernel (the Gram-form kNN over feature rows) in flight, and EVERY result is compared with the C
oracle's table ON THE DEVICE (torch.equal semantics; the mismatch counters are read once, at the end).  Different inputs alternate, so a
result that is another launch's result counts as a mismatch.

STATUS: written in round 6, which had no GPU access (gpurun refused every call: docs/kernels/round6.md section 1) -- this file has NOT
RUN ON HARDWARE YET.  It sorts last on purpose: the driver runs `pytest -x`, and a fault in a test that has never seen the GPU must not
hide the suite that has."""
import os

import pytest
import torch

from oracle import c_oracle as N

pytestmark = pytest.mark.gpu

# (the two knobs exist for the CPU wave emulator -- tests/native/emu_plugin.py --, where "beside other streams" means nothing and a launch
# takes a second: it runs this file with 24 launches and no noise to check the TEST's own logic: shapes, dtypes, the oracle calls)
LAUNCHES = int(os.environ.get("HITADV_CONCURRENCY_LAUNCHES", "1008"))  # per kernel; a multiple of 3 streams x 8 inputs
NOISE = os.environ.get("HITADV_CONCURRENCY_NOISE", "1") != "0"
INPUTS = 8


@pytest.fixture(scope="module")
def A():
    import hit_adv_amd.ops as ops
    return ops


def _cloud(n, seed, kind):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 3, generator=g)
    if kind == 'sphere':  # surface-like: points on the unit sphere + 1 % noise
        x = x / x.norm(dim=1, keepdim=True) + 0.01 * torch.randn(n, 3, generator=g)
    else:
        x = x - x.mean(0)
        x = x / x.norm(dim=1).max()
    return x


def _batch(b, n, seed):
    return torch.stack([_cloud(n, seed * 100 + i, 'sphere' if i % 2 else 'gaussian') for i in range(b)])


class Noise:
    """Two side streams that always have work queued: a 2048^3 fp32 GEMM, and the feature-space kNN (LDS-staged Gram tiles +
    selection) over 64-wide rows.  `pump()` tops both queues up; nothing of theirs is read."""

    def __init__(self, A):
        self.A = A
        self.s_gemm, self.s_lds = torch.cuda.Stream(), torch.cuda.Stream()
        g = torch.Generator().manual_seed(1)
        self.a = torch.randn(2048, 2048, generator=g).cuda()
        self.b = torch.randn(2048, 2048, generator=g).cuda()
        self.c = torch.empty(2048, 2048, device='cuda')
        self.feat = torch.randn(8, 1024, 64, generator=g).cuda()
        self.big_x, self.big_y = _batch(8, 2048, 7).cuda(), _batch(8, 2048, 8).cuda()
        torch.cuda.synchronize()

    def pump(self):
        if not NOISE:
            return
        with torch.cuda.stream(self.s_gemm):
            torch.mm(self.a, self.b, out=self.c)
        with torch.cuda.stream(self.s_lds):
            if self.A.knn_features_supported(64, 20):
                self.A.knn_features(self.feat, 20)
            self.A.nn_min(self.big_x, self.big_y)


def _hammer(A, launch, wants):
    """`launch(i)` -> tuple of device tensors for input i % INPUTS, on the current stream; `wants[i]` the oracle's tuple (device)."""
    noise = Noise(A)
    streams = [torch.cuda.Stream() for _ in range(3)]
    bad = [torch.zeros((), dtype=torch.int64, device='cuda') for _ in streams]  # one counter per stream: no cross-stream race on it
    torch.cuda.synchronize()
    for it in range(LAUNCHES):
        if it % 4 == 0:
            noise.pump()
        s = streams[it % 3]
        i = (it // 3) % INPUTS
        with torch.cuda.stream(s):
            got = launch(i)
            for g_, w_ in zip(got, wants[i]):
                assert g_.shape == w_.shape and g_.dtype == w_.dtype, (g_.shape, w_.shape, g_.dtype, w_.dtype)
                bad[it % 3] += (g_ != w_).any().to(torch.int64)
    torch.cuda.synchronize()
    return int(sum(int(b.item()) for b in bad))


def _dev(*ts):
    return tuple(t.cuda() for t in ts)


@pytest.mark.parametrize("K", [17, 6])
def test_knn_points_direct_form_beside_other_streams(A, K):
    """HiT_ADV.py:78-80 (K = curv_loss_knn + 1 = 17, self-cloud) and CW/kNN's K = 5 + 1: knn_select<K, form 0> -- the kernel that
    carries 16 packed f32 instructions per instantiation (none of them in the suspect pair: tests/test_isa_guards.py)."""
    from hit_adv_amd.pytorch3d_ops import knn_points
    xs = [_batch(4, 1024, 10 + i) for i in range(INPUTS)]
    wants = [_dev(*N.knn_points(x, x, K)) for x in xs]
    xd = [x.cuda() for x in xs]

    def launch(i):
        r = knn_points(xd[i], xd[i], K=K)
        return r.dists, r.idx
    assert _hammer(A, launch, wants) == 0


def test_knn_points_cross_cloud_beside_other_streams(A):
    """HiT_ADV.py:329-336: the 256 sampled centres against the cloud, K = 17."""
    from hit_adv_amd.pytorch3d_ops import knn_points
    ps = [_batch(4, 1024, 30 + i) for i in range(INPUTS)]
    qs = [p[:, ::4].contiguous() for p in ps]
    wants = [_dev(*N.knn_points(q, p, 17)) for q, p in zip(qs, ps)]
    pd, qd = [p.cuda() for p in ps], [q.cuda() for q in qs]

    def launch(i):
        r = knn_points(qd[i], pd[i], K=17)
        return r.dists, r.idx
    assert _hammer(A, launch, wants) == 0


def test_nn_min_beside_other_streams(A):
    """util/set_distance.py:40-70: both directions' minima and arg-minima of Chamfer / Hausdorff, direct form."""
    xs = [_batch(4, 1024, 50 + i) for i in range(INPUTS)]
    ys = [_batch(4, 1024, 70 + i) for i in range(INPUTS)]
    wants = []
    for x, y in zip(xs, ys):
        mx, ax = N.nn_min(x, y)
        my, ay = N.nn_min(y, x)
        wants.append(_dev(mx, ax, my, ay))
    xd, yd = [x.cuda() for x in xs], [y.cuda() for y in ys]
    assert _hammer(A, lambda i: A.nn_min(xd[i], yd[i]), wants) == 0


def test_extension_ball_query_and_fps_beside_other_streams(A):
    """_ext-src/src/ball_query_gpu.cu:9-44 and sampling_gpu.cu:69-173 semantics (uniform_loss's calls, GeoA3_args.py:258-302):
    51 samples, then the ball of radius 0.22 with 49 neighbours around them."""
    from hit_adv_amd.pointnet2_ops import _ext
    xs = [_batch(4, 1024, 90 + i) for i in range(INPUTS)]
    wants, centres = [], []
    for x in xs:
        f = N.furthest_point_sampling(x, 51)
        new_xyz = torch.gather(x, 1, f.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        centres.append(new_xyz.cuda())
        wants.append(_dev(f, N.ball_query(new_xyz, x, 0.22, 49)))
    xd = [x.cuda() for x in xs]

    def launch(i):
        return _ext.furthest_point_sampling(xd[i], 51), _ext.ball_query(centres[i], xd[i], 0.22, 49)
    assert _hammer(A, launch, wants) == 0


def test_victim_ball_query_beside_other_streams(A):
    """model/pointnet2_utils.py:87-107 (PointNet++'s own query_ball_point: Gram-form square_distance, `>` radius^2, sort): cfg4's first level."""
    xs = [_batch(2, 2048, 110 + i) for i in range(INPUTS)]
    qs = [x[:, ::4].contiguous() for x in xs]
    wants = [_dev(N.query_ball_point(0.2, 32, x, q)) for x, q in zip(xs, qs)]
    xd, qd = [x.cuda() for x in xs], [q.cuda() for q in qs]
    assert _hammer(A, lambda i: (A.query_ball_point(0.2, 32, xd[i], qd[i], reference=True),), wants) == 0


@pytest.mark.parametrize("form", [0, 1])
def test_fps_from_start_and_pct_sampler_beside_other_streams(A, form):
    """HiT_ADV.py:489-510 (N = 1024 -> 256, random start) and util/other_utils.py:254-272 (PCT: 1024 -> 512), by the 64-bit-key kernel
    (form 0, the default) and by fps_lean (form 1, HITADV_FPS_FORM=1): the second is the kernel whose packed-f32 build failed under
    exactly this kind of load."""
    from hit_adv_amd import _lib
    L = _lib.load()
    xs = [_batch(4, 1024, 130 + i) for i in range(INPUTS)]
    starts = [torch.tensor([3, 1000, 511, 0]) + i for i in range(INPUTS)]
    wants = [_dev(N.fps_from_start(x, 256, s), N.fps_pct(x, 512, s)) for x, s in zip(xs, starts)]
    xd, sd = [x.cuda() for x in xs], [s.cuda() for s in starts]
    shipped = L.hitadv_debug_fps_form(-1)
    try:
        L.hitadv_debug_fps_form(form)
        n_bad = _hammer(A, lambda i: (A.fps_from_start(xd[i], 256, sd[i]), A.fps_pct(xd[i], 512, sd[i], reference=True)), wants)
    finally:
        L.hitadv_debug_fps_form(shipped)
    assert n_bad == 0
