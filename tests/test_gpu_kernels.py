"""GPU parity tests: every HIP kernel, called through the C ABI (hit_adv_amd.ops / _ext -> ctypes ->
libhitadv_hip.so), against the CPU oracle and the golden vectors captured from the reference.

Bar: bit-exact for every index output and for direct-form squared distances; for floating point
results the tolerance is written next to each assertion.
"""
import copy
import os

import numpy as np
import pytest
import torch

from helpers import close as _close
from helpers import T, golden, note, synth_batch
from oracle import c_oracle as N
from oracle import hitadv_oracle as O

pytestmark = pytest.mark.gpu

EPS = float(np.finfo(np.float32).eps)


@pytest.fixture(scope="module")
def A():
    import hit_adv_amd.ops as ops
    from hit_adv_amd import _lib
    _lib.load()
    return ops


def cu(t):
    return t.cuda()


def close(a, b, rtol=1e-05, atol=1e-6, what=None):
    _close(a, b, rtol=rtol, atol=atol, what=what)


def clouds(b, n, first=0, kind='gaussian'):
    """``kind='sphere'``: the surface-like distribution (points on the unit sphere, 1 % radial noise): a scan's
    neighbour statistics, and many more near-tied distances than an iid Gaussian cloud has."""
    if kind == 'sphere':
        from hit_adv_amd.Dataset.synthetic import sphere_batch
        data, _ = sphere_batch(b, n, first=first)
    else:
        data, _ = synth_batch(b, n, first=first)
    return data[:, :, :3].contiguous(), data[:, :, 3:].contiguous()


KINDS = pytest.mark.parametrize("kind", ["gaussian", "sphere"])


# ------------------------------------------------------------------ K1 pairwise
@pytest.mark.parametrize("n,m", [(1024, 1024), (256, 1024), (100, 1001), (7, 3), (2, 2)])
def test_pairwise_direct_bit_exact(A, n, m):
    x, _ = clouds(2, n, 100)
    y, _ = clouds(2, m, 110)
    P = A.pairwise_sqdist(cu(x), cu(y), A.FORM_DIRECT).cpu()
    assert torch.equal(P, O.pairwise_sqdist_direct(x, y))


@pytest.mark.parametrize("n,m", [(1024, 1024), (100, 1001), (7, 3)])
@pytest.mark.parametrize("form", ["gram", "gram_knn"])
def test_pairwise_gram_forms_bit_exact(A, n, m, form):
    """Both Gram forms reproduce the oracle's restatement of the reference's arithmetic bit for bit -- which in turn is
    checked bit for bit against torch itself in tests/test_oracle_gram.py (set_distance.py:15-32, dist_utils.py:148-150)."""
    x, _ = clouds(2, n, 100)
    y, _ = clouds(2, m, 110)
    f_hip, f_c = (A.FORM_GRAM, N.FORM_GRAM) if form == "gram" else (A.FORM_GRAM_KNN, N.FORM_GRAM_KNN)
    P = A.pairwise_sqdist(cu(x), cu(y), f_hip).cpu()
    assert torch.equal(P, N.pairwise(x, y, f_c))
    truth = ((x.double()[:, :, None] - y.double()[:, None]) ** 2).sum(-1)
    scale = float((x ** 2).sum(-1).max() + (y ** 2).sum(-1).max())
    assert (P - truth).abs().max().item() <= 8 * EPS * scale  # the Gram form's own cancellation noise


def test_pairwise_generic_dim(A):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 5, 70, generator=g)
    y = torch.randn(2, 4, 70, generator=g)
    P = A.pairwise_sqdist(cu(x), cu(y)).cpu()
    truth = ((x.double()[:, :, None] - y.double()[:, None]) ** 2).sum(-1)
    close(P, truth.float(), rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------ K2 fused NN-min
@KINDS
@pytest.mark.parametrize("n,m", [(1024, 1024), (256, 1024), (1000, 37), (1, 5), (130, 2049)])
def test_nn_min_bit_exact_both_directions(A, n, m, kind):
    x, _ = clouds(3, n, 120, kind)
    y, _ = clouds(3, m, 130, kind)
    mx, ax, my, ay = (t.cpu() for t in A.nn_min(cu(x), cu(y)))
    rx, rax = N.nn_min(x, y)
    ry, ray = N.nn_min(y, x)
    assert torch.equal(mx, rx) and torch.equal(ax, rax)
    assert torch.equal(my, ry) and torch.equal(ay, ray)


@pytest.mark.parametrize("n,m", [(1024, 1024), (256, 1024), (1000, 37), (130, 2049)])
def test_nn_min_reference_arithmetic_bit_exact(A, n, m):
    """Gram-form minima (the reference's own values, set_distance.py:45-49) in both directions, bit for bit."""
    x, _ = clouds(3, n, 120)
    y = clouds(3, m, 130)[0] if n != m else x + 0.01 * torch.randn(x.shape, generator=torch.Generator().manual_seed(2))
    mx, ax, my, ay = (t.cpu() for t in A.nn_min(cu(x), cu(y), reference=True))
    rx, rax = N.nn_min(x, y, N.FORM_GRAM)
    ry, ray = N.nn_min(y, x, N.FORM_GRAM)
    assert torch.equal(mx, rx) and torch.equal(ax, rax)
    assert torch.equal(my, ry) and torch.equal(ay, ray)
    with A.reference_arithmetic(True):  # the package-wide switch selects the same kernels
        mx2, _, my2, _ = A.nn_min(cu(x), cu(y))
    assert torch.equal(mx2.cpu(), rx) and torch.equal(my2.cpu(), ry)
    assert not A.reference_arithmetic.get()


def test_nn_min_ties_take_lowest_index(A):
    x, _ = clouds(1, 64, 140)
    y = torch.cat([x, x, x], 1)  # every query has three exact nearest neighbours at distance 0
    mx, ax, my, ay = (t.cpu() for t in A.nn_min(cu(x), cu(y)))
    assert (mx == 0).all() and torch.equal(ax[0], torch.arange(64, dtype=torch.int32))
    assert (my == 0).all() and torch.equal(ay[0], torch.arange(64, dtype=torch.int32).repeat(3))


def test_nn_min_generic_dim_q1_shape(A):
    # quirk Q1: [B,3,N] tensors, i.e. 3 "points" of dimension N
    x, _ = clouds(2, 1024, 150)
    y = x + 0.01 * torch.randn(x.shape, generator=torch.Generator().manual_seed(3))
    xt, yt = x.transpose(1, 2).contiguous(), y.transpose(1, 2).contiguous()
    mx, ax, my, ay = (t.cpu() for t in A.nn_min(cu(xt), cu(yt)))
    P = ((xt.double()[:, :, None] - yt.double()[:, None]) ** 2).sum(-1)
    close(mx, P.min(2).values.float(), rtol=1e-5)
    close(my, P.min(1).values.float(), rtol=1e-5)
    assert torch.equal(ax.long(), P.argmin(2)) and torch.equal(ay.long(), P.argmin(1))


def test_set_distance_modules_vs_reference_vectors(A):
    from hit_adv_amd.util import dist_utils, set_distance
    fx = golden('g1_set_distance.npz')
    adv, ori, small, w = (cu(T(fx[k])) for k in ('adv', 'ori', 'small', 'weights'))
    # the reference evaluates the Gram form whose own fp32 noise is ~4*eps*(|x|^2+|y|^2) per entry;
    # min-values here are ~1e-3, so means agree to ~1e-5 relative and maxima to the noise floor.
    noise = 8 * EPS * 2.0
    for name, mod, a, b in (('chamfer', set_distance.chamfer, adv, ori),
                            ('hausdorff', set_distance.hausdorff, adv, ori),
                            ('chamfer_small', set_distance.chamfer, small, ori),
                            ('hausdorff_small', set_distance.hausdorff, small, ori)):
        l1, l2 = mod(a, b)
        close(l1, fx[name + '_l1'], rtol=1e-5, atol=noise)
        close(l2, fx[name + '_l2'], rtol=1e-5, atol=noise)
    for m in ('adv2ori', 'ori2adv', 'both'):
        close(dist_utils.ChamferDist(m)(adv, ori, w, batch_avg=False), fx['ChamferDist_%s' % m], atol=noise)
        close(dist_utils.HausdorffDist(m)(adv, ori, w, batch_avg=False), fx['HausdorffDist_%s' % m], atol=noise)
    close(dist_utils.ChamferDist()(adv, ori), fx['ChamferDist_avg'], atol=noise)
    close(dist_utils.HausdorffDist()(adv, ori), fx['HausdorffDist_avg'], atol=noise)
    q1 = dist_utils.ChamferDist()(adv.transpose(1, 2).contiguous(), ori.transpose(1, 2).contiguous(),
                                  torch.from_numpy(np.ones(2) * 1e-4), batch_avg=False)
    close(q1, fx['ChamferDist_q1'], rtol=1e-4)
    a = adv.clone().requires_grad_()
    dist_utils.ChamferDist('both')(a, ori, w).backward()
    close(a.grad, fx['ChamferDist_both_grad'], rtol=1e-5, atol=1e-7)
    a = adv.clone().requires_grad_()
    dist_utils.HausdorffDist('both')(a, ori, w).backward()
    close(a.grad, fx['HausdorffDist_both_grad'], rtol=1e-5, atol=1e-7)


def test_set_distance_modules_reference_arithmetic_equal_reference_vectors(A):
    """north_star's bar -- Chamfer / Hausdorff within 1e-5 relative of the reference -- WITHOUT an absolute floor: with
    ``reference_arithmetic`` on, the minima are the reference's bit for bit, so Hausdorff values are EQUAL and Chamfer
    values differ by the summation order of a 1024-term mean."""
    from hit_adv_amd.util import dist_utils, set_distance
    fx = golden('g1_set_distance.npz')
    adv, ori, small, w = (cu(T(fx[k])) for k in ('adv', 'ori', 'small', 'weights'))
    cham, haus = set_distance.ChamferDistance(True), set_distance.HausdorffDistance(True)
    for name, mod, a, b in (('chamfer', cham, adv, ori), ('chamfer_small', cham, small, ori)):
        l1, l2 = mod(a, b)
        close(l1, fx[name + '_l1'], rtol=1e-6, atol=0, what=name + ' l1 (reference arithmetic)')
        close(l2, fx[name + '_l2'], rtol=1e-6, atol=0, what=name + ' l2 (reference arithmetic)')
    for name, a in (('hausdorff', adv), ('hausdorff_small', small)):
        l1, l2 = haus(a, ori)
        assert np.array_equal(l1.cpu().numpy(), fx[name + '_l1']) and np.array_equal(l2.cpu().numpy(), fx[name + '_l2'])
    for m in ('adv2ori', 'ori2adv', 'both'):
        close(dist_utils.ChamferDist(m, reference_arithmetic=True)(adv, ori, w, batch_avg=False),
              fx['ChamferDist_%s' % m], rtol=1e-6, atol=0, what='ChamferDist %s (reference arithmetic)' % m)
        close(dist_utils.HausdorffDist(m, reference_arithmetic=True)(adv, ori, w, batch_avg=False),
              fx['HausdorffDist_%s' % m], rtol=1e-6, atol=0, what='HausdorffDist %s (reference arithmetic)' % m)
    fx2 = golden('g2_knn_dist.npz')
    adv2, ori2 = cu(T(fx2['adv'])), cu(T(fx2['ori']))
    for k in (4, 5):
        close(dist_utils.KNNDist(k=k, reference_arithmetic=True)(adv2, batch_avg=False), fx2['KNNDist_k%d' % k],
              rtol=1e-5, atol=0, what='KNNDist k=%d (reference arithmetic)' % k)
    close(dist_utils.ChamferkNNDist(reference_arithmetic=True)(adv2, ori2, batch_avg=False), fx2['ChamferkNNDist'],
          rtol=1e-5, atol=0, what='ChamferkNNDist (reference arithmetic)')
    # gradients still flow (through the same arg-minima), evaluated as 2 g (x - y)
    a = adv.clone().requires_grad_()
    dist_utils.ChamferDist('both', reference_arithmetic=True)(a, ori, w).backward()
    close(a.grad, fx['ChamferDist_both_grad'], rtol=1e-4, atol=1e-7, what='ChamferDist grad (reference arithmetic)')


def test_nn_min_backward_matches_autograd_of_direct_matrix(A):
    x, _ = clouds(2, 300, 160)
    y, _ = clouds(2, 257, 170)
    g = torch.Generator().manual_seed(5)
    wx, wy = torch.randn(2, 300, generator=g), torch.randn(2, 257, generator=g)
    xr, yr = x.clone().requires_grad_(), y.clone().requires_grad_()
    P = O.pairwise_sqdist_direct(xr, yr)
    ((P.min(2).values * wx).sum() + (P.min(1).values * wy).sum()).backward()
    xg, yg = cu(x).requires_grad_(), cu(y).requires_grad_()
    mx, _, my, _ = A.nn_min(xg, yg)
    ((mx * cu(wx)).sum() + (my * cu(wy)).sum()).backward()
    close(xg.grad, xr.grad, rtol=1e-5, atol=1e-6)
    close(yg.grad, yr.grad, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ K4 kNN
@KINDS
@pytest.mark.parametrize("K", [1, 4, 5, 6, 8, 16, 17, 20, 30, 32, 33, 64])
def test_knn_points_bit_exact(A, K, kind):
    from hit_adv_amd.pytorch3d_ops import knn_points
    x, _ = clouds(2, 1024, 180, kind)
    q = x[:, :300].contiguous()
    r = knn_points(cu(q), cu(x), K=K)
    d, ix = N.knn_points(q, x, K)
    assert r.idx.dtype == torch.int64
    assert torch.equal(r.idx.cpu(), ix) and torch.equal(r.dists.cpu(), d)


@pytest.mark.parametrize("m", [2048, 1500, 51, 7, 3000])
@pytest.mark.parametrize("K", [3, 6, 17, 32])
def test_knn_points_sizes_and_both_selection_kernels(A, m, K):
    """Reference counts on both sides of the select kernel's range (M <= 2048; beyond it the insertion-list kernel runs),
    fewer references than a lane's list (M = 7, 51), queries != references."""
    if K > m:
        pytest.skip("K > M")
    x, _ = clouds(2, m, 181)
    q, _ = clouds(2, 130, 182)
    d, ix = A.KnnPoints.apply(cu(q), cu(x), K)
    rd, rix = N.knn_points(q, x, K)
    assert torch.equal(ix.cpu(), rix) and torch.equal(d.cpu(), rd)


@pytest.mark.parametrize("K", [5, 6, 17, 33])
def test_knn_points_gram_knn_form_bit_exact(A, K):
    """KNNDist's own distance matrix (dist_utils.py:148-150) inside the fused selection: the k+1 smallest values the
    reference's topk(-dist) returns, bit for bit (rank 0 need not be the point itself in this form)."""
    x, _ = clouds(2, 1024, 183)
    d, ix = A.KnnPoints.apply(cu(x), cu(x), K, A.FORM_GRAM_KNN)
    rd, rix = N.knn_points(x, x, K, N.FORM_GRAM_KNN)
    assert torch.equal(d.cpu(), rd) and torch.equal(ix.cpu(), rix)


@pytest.mark.parametrize("K,m,form", [(2, 1024, 0), (5, 1024, 2), (6, 1024, 0), (6, 2048, 2), (8, 512, 0), (12, 1024, 0),
                                      (17, 1024, 0), (17, 2048, 0), (18, 1000, 2)])
@KINDS
def test_knn_points_full_batch_with_ties_and_falling_distances(A, K, m, form, kind):
    """A batch as large as the attack's own calls (16 clouds x 1061 queries, ragged last block), with one cloud made of six
    distinct points (exact ties everywhere) and one whose references come ever closer (the logs overflow and compact)."""
    B, n = 16, 1024 + 37  # ragged last block
    x, _ = clouds(B, m, 184, kind)
    q, _ = clouds(B, n, 185, kind)
    g = torch.Generator().manual_seed(K)
    x[1] = torch.randn(1, 6, 3, generator=g)[:, torch.randint(0, 6, (m,), generator=g)]  # six distinct points: ties
    x[2, :, 0] = torch.linspace(5.0, 0.5, m)  # every next reference is closer: the logs overflow and compact
    d, ix = A.KnnPoints.apply(cu(q), cu(x), K, A.FORM_GRAM_KNN if form else A.FORM_DIRECT)
    rd, rix = N.knn_points(q, x, K, N.FORM_GRAM_KNN if form else N.FORM_DIRECT)
    assert torch.equal(ix.cpu(), rix) and torch.equal(d.cpu(), rd)


def test_knn_points_vs_independent_float64_top_k(A):
    """An independent statement of the selection rule: brute force in float64 -- squared differences summed in
    float64, ``torch.topk`` -- on clouds whose neighbour distances are well separated, so that fp32 rounding cannot change
    the ranking: same indices.  And the explicit tie policy on exact duplicates: equal distances rank by ascending index."""
    x, _ = clouds(2, 777, 186)
    q, _ = clouds(2, 200, 187)
    d64 = ((q.double()[:, :, None, :] - x.double()[:, None, :, :]) ** 2).sum(-1)
    K = 17
    ref = d64.topk(K + 1, dim=-1, largest=False, sorted=True)
    gaps = (ref.values[..., 1:] - ref.values[..., :-1]) / ref.values[..., 1:].clamp_min(1e-30)
    clear = (gaps > 1e-5).all(-1)  # every one of the K+1 nearest is separated from the next by more than fp32 noise
    assert clear.float().mean() > 0.8
    d, ix = A.KnnPoints.apply(cu(q), cu(x), K)
    assert torch.equal(ix.cpu()[clear], ref.indices[..., :K][clear])
    close(d.cpu()[clear], ref.values[..., :K][clear].float(), rtol=2e-6, atol=1e-9, what='kNN distances vs float64 brute force')
    dup = torch.cat([x[:, :50], x[:, :50].flip(1), x[:, :50]], 1)  # every point three times, at indices i, 99-i, 100+i
    d, ix = A.KnnPoints.apply(cu(x[:, :50].contiguous()), cu(dup), 3)
    i = torch.arange(50)
    want = torch.stack([torch.minimum(i, 99 - i), torch.maximum(i, 99 - i), 100 + i], -1)
    assert (d == 0).all() and torch.equal(ix.cpu(), want.expand(2, 50, 3))


def test_knn_points_heavy_ties_and_log_compaction(A):
    """Adversarial orders for the select kernel: distances that fall monotonically along the scan (every reference is
    accepted: the per-lane log overflows and is compacted again and again) and clouds made of a few distinct points
    (many exact ties at the threshold)."""
    g = torch.Generator().manual_seed(9)
    q = torch.zeros(1, 64, 3)
    q[0, :, 0] = torch.linspace(-0.01, 0.01, 64)
    far_to_near = torch.zeros(1, 1024, 3)
    far_to_near[0, :, 0] = torch.linspace(5.0, 0.5, 1024)   # reference j+1 is closer than reference j for every query
    far_to_near[0, :, 1] = 0.001 * torch.randn(1024, generator=g)
    few = torch.randn(1, 6, 3, generator=g)[:, torch.randint(0, 6, (1024,), generator=g)]
    for refs in (far_to_near, few, torch.cat([few[:, :500], far_to_near[:, :524]], 1)):
        for K in (6, 17, 32):
            d, ix = A.KnnPoints.apply(cu(q), cu(refs), K)
            rd, rix = N.knn_points(q, refs, K)
            assert torch.equal(ix.cpu(), rix) and torch.equal(d.cpu(), rd), K


def test_knn_points_duplicates_ragged_and_gather(A):
    from hit_adv_amd.pytorch3d_ops import knn_gather, knn_points
    x, _ = clouds(2, 77, 190)
    dup = torch.cat([x[:, :9], x[:, :9], x], 1)  # M = 95, exact ties
    r = knn_points(cu(dup), cu(dup), K=4, return_nn=True)
    d, ix = N.knn_points(dup, dup, 4)
    assert torch.equal(r.idx.cpu(), ix) and torch.equal(r.dists.cpu(), d)
    assert torch.equal(r.knn.cpu(), O.knn_gather(dup, ix))
    assert torch.equal(knn_gather(cu(dup), r.idx).cpu(), O.knn_gather(dup, ix))
    with pytest.raises(RuntimeError):
        knn_points(cu(x), cu(x), K=78)


@pytest.mark.parametrize("n,m,K", [(200, 210, 5), (1024, 1024, 6), (700, 33, 17)])
def test_knn_points_backward(A, n, m, K):
    """(1024, 1024, 6): the reference-side scatter streams its (index, gradient) entries through LDS in two chunks."""
    from hit_adv_amd.pytorch3d_ops import knn_points
    x, _ = clouds(2, n, 200)
    y, _ = clouds(2, m, 150)
    w = torch.randn(2, n, K, generator=torch.Generator().manual_seed(6))
    xr, yr = x.clone().requires_grad_(), y.clone().requires_grad_()
    P = O.pairwise_sqdist_direct(xr, yr)
    (torch.sort(P, dim=-1, stable=True).values[..., :K] * w).sum().backward()
    xg, yg = cu(x).requires_grad_(), cu(y).requires_grad_()
    (knn_points(xg, yg, K=K).dists * cu(w)).sum().backward()
    if m < 100:  # few references: hundreds of terms per reference gradient, the tolerance follows their magnitude
        close(xg.grad, xr.grad, rtol=1e-5, atol=1e-6)
        close(yg.grad, yr.grad, rtol=1e-5, atol=2e-7 * float(yr.grad.abs().max()))
        return
    close(xg.grad, xr.grad, rtol=1e-5, atol=1e-6)
    close(yg.grad, yr.grad, rtol=1e-5, atol=1e-6)


def test_knn_dist_operators_vs_reference_vectors(A):
    from hit_adv_amd.util import dist_utils
    fx = golden('g2_knn_dist.npz')
    ori, nrm, adv = (cu(T(fx[k])) for k in ('ori', 'normal', 'adv'))
    # achieved on MI355X: 7.7e-8 / 2.4e-7 relative (profiles/r03_parity_report.json); north_star's bar is 1e-5
    for k in (4, 5):
        close(dist_utils.KNNDist(k=k)(adv, batch_avg=False), fx['KNNDist_k%d' % k], rtol=1e-6)
        close(dist_utils.KNNDist(k=k)(adv.transpose(1, 2).contiguous(), batch_avg=False),
              fx['KNNDist_k%d_chfirst' % k], rtol=1e-6)
    close(dist_utils.ChamferkNNDist()(adv, ori, batch_avg=False), fx['ChamferkNNDist'], rtol=1e-6)
    ori_t, adv_t, nrm_t = (t.transpose(1, 2).contiguous() for t in (ori, adv, nrm))
    close(dist_utils.CurvStdDist(k=4)(ori_t, adv_t, nrm_t), fx['CurvStdDist_k4'], rtol=1e-5)
    kstd, kappa, _ = dist_utils.curvature_std(ori_t, nrm_t, 16)
    close(kappa, fx['kappa_k16'], rtol=1e-5)
    close(kstd, fx['kappa_std_k16'], rtol=1e-5, atol=1e-7)


def test_knn_dist_grad_vs_direct_form_oracle(A):
    """Gradient parity against autograd through the same (direct-form) distances; the reference's
    Gram-form gradient differs only through which near-tied point is selected/thresholded."""
    from hit_adv_amd.util import dist_utils
    x, _ = clouds(2, 400, 220)
    w = torch.tensor([0.5, 2.0])

    def oracle_direct(pc):
        d = torch.sort(O.pairwise_sqdist_direct(pc, pc), dim=-1, stable=True).values[..., 1:6].mean(-1)
        with torch.no_grad():
            mask = (d > (d.mean(-1) + 1.05 * d.std(-1))[:, None]).float()
        return ((d * mask).mean(1) * w).mean()

    xr = x.clone().requires_grad_()
    oracle_direct(xr).backward()
    xg = cu(x).requires_grad_()
    v = dist_utils.KNNDist(k=5)(xg, cu(w))
    v.backward()
    close(v, oracle_direct(x), rtol=1e-5)
    close(xg.grad, xr.grad, rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------ K3 deformation
@pytest.mark.parametrize("C", [16, 192])
def test_deform_forward_backward_vs_reference_vectors(A, C):
    fx = golden('g3_deform.npz')
    p = 'c%d_' % C
    ori, central, up = cu(T(fx['ori'])), cu(T(fx[p + 'central'])), cu(T(fx[p + 'upstream']))
    P = cu(T(fx[p + 'P'])).requires_grad_()
    sig = cu(T(fx[p + 'sigma'])).requires_grad_()
    adv = A.deform(ori, central, P, sig)
    close(adv, fx[p + 'adv'], rtol=1e-5, atol=2e-6)  # fused sum vs 192-step fp32 accumulation
    (adv * up).sum().backward()
    scale_p = float(np.abs(fx[p + 'grad_P']).max())
    scale_s = float(np.abs(fx[p + 'grad_sigma']).max())
    close(P.grad, fx[p + 'grad_P'], rtol=1e-4, atol=1e-5 * scale_p)
    close(sig.grad, fx[p + 'grad_sigma'], rtol=1e-4, atol=1e-5 * scale_s)


def test_deform_identity_and_ragged_sizes(A):
    x, _ = clouds(3, 1000, 230)
    ori = cu(x.transpose(1, 2).contiguous())
    central = ori[:, :, :37].contiguous()
    sig = torch.full((3, 37), 0.4, device='cuda')
    adv = A.deform(ori, central, torch.zeros(3, 37, 3, device='cuda'), sig)
    assert torch.equal(adv, ori)  # zero translations leave every point where it was, exactly
    P = torch.rand(3, 37, 3, generator=torch.Generator().manual_seed(2)) - 0.5
    adv = A.deform(ori, central, cu(P), sig).cpu()
    ref = O.deform_loop(x.transpose(1, 2).contiguous(), P,
                        O.kernel_density(central.cpu(), x.transpose(1, 2).contiguous(), sig.cpu()))
    close(adv, ref, rtol=1e-5, atol=2e-6)


# ------------------------------------------------------------------ K5 FPS + natives
def test_fps_from_start_vs_reference_vector(A):
    fx = golden('g4_fps.npz')
    idx = A.fps_from_start(cu(T(fx['xyz'])), 256, cu(T(fx['start'])))
    assert idx.dtype == torch.int64 and (idx.cpu().numpy() == fx['idx']).all()


@KINDS
@pytest.mark.parametrize("n,m", [(1024, 256), (2048, 512), (300, 300), (64, 5), (5000, 64)])
def test_fps_from_start_bit_exact_sizes(A, n, m, kind):
    x, _ = clouds(3, n, 240, kind)
    start = torch.tensor([0, n - 1, n // 2])
    assert torch.equal(A.fps_from_start(cu(x), m, cu(start)).cpu(), N.fps_from_start(x, m, start))


@pytest.mark.parametrize("n,m", [(1024, 512), (512, 256), (2048, 512), (300, 300), (64, 5), (5000, 64)])
def test_fps_pct_bit_exact_sizes(A, n, m):
    """PCT's sampler (util/other_utils.py:254-272): running distances = sqrt of the clamped Gram form in torch's own fp32
    arithmetic (oracle form 4, pinned against torch and fixture g12 in tests/test_oracle_gram.py, test_victims_cpu.py)."""
    x, _ = clouds(4, n, 245)
    x[3, 7] = x[3, 3]  # a duplicate: the clamped branch (d < 0 -> 1e-7) and exact ties
    start = torch.tensor([0, n - 1, n // 2, 3])
    got = A.fps_pct(cu(x), m, cu(start), reference=True).cpu()
    assert torch.equal(got, N.fps_pct(x, m, start))
    assert torch.equal(A.fps_pct(cu(x), m, cu(start), reference=False).cpu(), N.fps_from_start(x, m, start))


@pytest.mark.parametrize("n,m", [(600, 600), (1024, 400), (2048, 300), (4500, 40)])
def test_fps_both_samplers_on_a_lattice_of_exact_ties(A, n, m):
    """Coordinates on a coarse lattice with repeated points: most steps have several points at exactly the maximal running
    distance (and, for PCT's sampler, distinct squared distances whose sqrt rounds to one float), and m = n drives every
    running distance to zero -- the lowest index must win each time (model/pointnet2_utils.py:63-84 through torch.max;
    util/other_utils.py:254-272).  The 64-bit-key kernel and the lean one (sampling.hip) must both give the oracle's table."""
    g = torch.Generator().manual_seed(77 + n)
    x = torch.randint(-3, 4, (3, n, 3), generator=g).float() * 0.25
    x[1] += 0.001 * torch.randn(n, 3, generator=g).round(decimals=3)  # near-lattice: values one or two ulps apart
    start = torch.tensor([0, n - 1, 5])
    want0, want2 = N.fps_from_start(x, m, start), N.fps_pct(x, m, start)
    from hit_adv_amd import _lib
    L = _lib.load()
    shipped = L.hitadv_debug_fps_form(-1)  # (an invalid value changes nothing and returns the current form)
    try:
        for form in (0, 1):
            L.hitadv_debug_fps_form(form)
            assert torch.equal(A.fps_from_start(cu(x), m, cu(start)).cpu(), want0), form
            assert torch.equal(A.fps_pct(cu(x), m, cu(start), reference=True).cpu(), want2), form
    finally:
        L.hitadv_debug_fps_form(shipped)


def test_fps_pct_reproduces_the_reference_table(A):
    fx = golden('g12_pct.npz')
    pts = T(fx['x']).transpose(1, 2).contiguous()
    ref = T(fx['fps1'])
    assert torch.equal(A.fps_pct(cu(pts), 512, cu(ref[:, 0].contiguous())).cpu(), ref)


@pytest.mark.parametrize("n,s,radius,nsample", [(1024, 512, 0.2, 32), (512, 128, 0.4, 64), (2048, 512, 0.2, 32), (300, 77, 0.3, 16),
                                                (64, 48, 0.05, 8)])
@KINDS
def test_query_ball_point_victim_bit_exact(A, n, s, radius, nsample, kind):
    """The victims' ball query (model/pointnet2_utils.py:87-107) on the reference's Gram-form square_distance (oracle form
    3): threshold = the fp32 value of the double radius ** 2, ``>`` excluded, index order, padded with the first hit, an
    empty ball = n."""
    x, _ = clouds(3, n, 246, kind)
    q = x[:, :s].contiguous().clone()
    q[0, 1] = 50.  # an empty ball
    for k in range(2, min(s, 40)):  # points ON the sphere up to rounding: where the two forms and the two thresholds part
        d = torch.nn.functional.normalize(torch.randn(3, generator=torch.Generator().manual_seed(k)), dim=0)
        x[1, k + 100 if k + 100 < n else k] = q[1, k] + d * radius
    for reference, form in ((True, N.FORM_SQUARE_DISTANCE), (False, N.FORM_DIRECT)):
        got = A.query_ball_point(radius, nsample, cu(x), cu(q), reference=reference).cpu()
        assert got.dtype == torch.int64 and torch.equal(got, N.query_ball_point(radius, nsample, x, q, form))
    assert int(got[0, 1, 0]) == n


def test_query_ball_point_victim_reproduces_the_reference_table(A):
    fx = golden('g11_pointnet2.npz')
    pts = T(fx['x']).transpose(1, 2).contiguous()
    new = torch.gather(pts, 1, T(fx['fps1']).unsqueeze(-1).expand(-1, -1, 3))
    assert torch.equal(A.query_ball_point(0.2, 32, cu(pts), cu(new)).cpu(), T(fx['ball1']))


@pytest.mark.parametrize("n,s,K", [(1024, 512, 32), (512, 256, 32), (2048, 300, 32), (3000, 64, 32), (300, 77, 8), (64, 48, 33)])
def test_knn_points_square_distance_form_bit_exact(A, n, s, K):
    """PCT's knn_point (model/pct_utils.py:98-109): the K smallest entries of square_distance(new_xyz, xyz), oracle form 3."""
    x, _ = clouds(3, n, 247)
    q = x[:, :s].contiguous() + 0.
    x[2, 11] = x[2, 5]
    d, ix = A.KnnPoints.apply(cu(q), cu(x), K, A.FORM_SQUARE_DISTANCE)
    rd, rix = N.knn_points(q, x, K, N.FORM_SQUARE_DISTANCE)
    assert torch.equal(d.cpu(), rd) and torch.equal(ix.cpu(), rix)


@pytest.mark.parametrize("n,m", [(1024, 51), (2048, 102), (300, 64), (700, 700), (5000, 33)])
def test_fps_ext_bit_exact(A, n, m):
    from hit_adv_amd.pointnet2_ops import _ext
    x, _ = clouds(3, n, 250)
    x[1, : n // 3] *= 0.01  # a block of near-origin points that the kernel must skip
    x[2, 5] = x[2, 9]
    out = _ext.furthest_point_sampling(cu(x), m)
    assert out.dtype == torch.int32 and torch.equal(out.cpu(), N.furthest_point_sampling(x, m))


def test_fps_ext_known_answers(A):
    from hit_adv_amd.pointnet2_ops import _ext
    line = torch.zeros(1, 8, 3)
    line[0, :, 0] = torch.arange(1, 9).float()
    assert _ext.furthest_point_sampling(cu(line), 4)[0].tolist() == [0, 7, 4, 2]
    skip = torch.tensor([[[1., 0, 0], [0.01, 0, 0], [0, 0.02, 0], [-1., 0, 0], [0, 1., 0]]])
    assert _ext.furthest_point_sampling(cu(skip), 3)[0].tolist() == [0, 3, 4]
    assert _ext.furthest_point_sampling(cu(torch.zeros(1, 4, 3)), 3)[0].tolist() == [0, 0, 0]
    tie = torch.zeros(1, 1024, 3)
    tie[0, :, 0] = 0.1
    tie[0, 1] = torch.tensor([0.1, 0.5, 0.0])
    tie[0, 514] = torch.tensor([0.1, -0.5, 0.0])
    assert _ext.furthest_point_sampling(cu(tie), 2)[0].tolist() == [0, 514]


def test_ball_query_group_gather_bit_exact(A):
    from hit_adv_amd.pointnet2_ops import _ext
    x, _ = clouds(3, 1024, 260)
    fidx = N.furthest_point_sampling(x, 51)
    flipped = x.transpose(1, 2).contiguous()
    new_xyz = N.gather_points(flipped, fidx).transpose(1, 2).contiguous()
    g_new = _ext.gather_points(cu(flipped), cu(fidx))
    assert torch.equal(g_new.cpu(), N.gather_points(flipped, fidx))
    for r, ns in ((0.126, 16), (0.2, 49), (0.01, 8), (5.0, 70)):
        ref = N.ball_query(new_xyz, x, r, ns)
        out = _ext.ball_query(cu(new_xyz), cu(x), r, ns)
        assert out.dtype == torch.int32 and torch.equal(out.cpu(), ref)
        assert torch.equal(_ext.group_points(cu(flipped), out).cpu(), N.group_points(flipped, ref))
    kx = torch.tensor([[[0., 0, 0], [0.5, 0, 0], [1.0, 0, 0], [0.2, 0, 0], [5, 5, 5]]])
    kq = torch.tensor([[[0., 0, 0], [9., 9, 9], [1.0, 0, 0]]])
    assert _ext.ball_query(cu(kq), cu(kx), 0.5, 3)[0].tolist() == [[0, 3, 0], [0, 0, 0], [2, 2, 2]]
    assert _ext.ball_query(cu(kq), cu(kx), 0.6, 2)[0].tolist() == [[0, 1], [0, 0], [1, 2]]


def test_native_gradients_and_interpolation(A):
    from hit_adv_amd.pointnet2_ops import _ext
    g = torch.Generator().manual_seed(0)
    pts = torch.randn(2, 6, 300, generator=g)
    idx = torch.randint(0, 300, (2, 20, 9), generator=g).int()
    go = torch.randn(2, 6, 20, 9, generator=g)
    close(_ext.group_points_grad(cu(go), cu(idx), 300), N.group_points_grad(go, idx, 300), rtol=1e-5, atol=1e-6)
    gi = torch.randint(0, 300, (2, 40), generator=g).int()
    gg = torch.randn(2, 6, 40, generator=g)
    close(_ext.gather_points_grad(cu(gg), cu(gi), 300), N.gather_points_grad(gg, gi, 300), rtol=1e-5, atol=1e-6)
    unknown, known = torch.randn(2, 500, 3, generator=g), torch.randn(2, 90, 3, generator=g)
    d2, ti = _ext.three_nn(cu(unknown), cu(known))
    rd2, rti = N.three_nn(unknown, known)
    assert torch.equal(ti.cpu(), rti) and torch.equal(d2.cpu(), rd2)
    tie = _ext.three_nn(cu(torch.zeros(1, 1, 3)),
                        cu(torch.tensor([[[1., 0, 0], [-1., 0, 0], [0, 1., 0], [0, 0, 2.]]])))
    assert tie[1][0, 0].tolist() == [0, 1, 2]
    feats = torch.randn(2, 6, 90, generator=g)
    w = torch.rand(2, 500, 3, generator=g)
    assert torch.equal(_ext.three_interpolate(cu(feats), ti, cu(w)).cpu(), N.three_interpolate(feats, rti, w))
    g3 = torch.randn(2, 6, 500, generator=g)
    close(_ext.three_interpolate_grad(cu(g3), ti, cu(w), 90), N.three_interpolate_grad(g3, rti, w, 90),
          rtol=1e-5, atol=1e-5)


def test_native_argument_checks(A):
    from hit_adv_amd.pointnet2_ops import _ext, pointnet2_utils as pu
    x, _ = clouds(1, 64, 270)
    with pytest.raises(RuntimeError):
        _ext.furthest_point_sampling(x, 4)  # CPU tensor
    with pytest.raises(RuntimeError):
        _ext.furthest_point_sampling(cu(x).transpose(1, 2), 4)  # non-contiguous
    with pytest.raises(RuntimeError):
        _ext.gather_points(cu(x), cu(torch.zeros(1, 4, dtype=torch.int64)))  # int64 idx
    idx = pu.furthest_point_sample(cu(x), 8)
    feats = cu(x.transpose(1, 2).contiguous()).requires_grad_()
    out = pu.gather_operation(feats, idx)
    out.sum().backward()
    assert feats.grad.sum().item() == pytest.approx(8 * 3)
    bq = pu.ball_query(0.5, 4, cu(x), out.transpose(1, 2).contiguous())
    grouped = pu.grouping_operation(feats, bq)
    assert grouped.shape == (1, 3, 8, 4)


# ------------------------------------------------------------------ attack-state kernels
def test_best_update_and_adam_match_host_logic(A):
    g = torch.Generator().manual_seed(4)
    B, K, Nn, C = 5, 40, 128, 12
    st = dict(bestdist=torch.full((B,), 1e10), bestscore=torch.full((B,), -1, dtype=torch.int64),
              o_bestdist=torch.full((B,), 1e10), o_bestscore=torch.full((B,), -1, dtype=torch.int64),
              o_bestattack=torch.zeros(B, 3, Nn), pred=torch.zeros(B, dtype=torch.int64),
              dist_val=torch.zeros(B))
    dst = {k: v.cuda() for k, v in st.items()}
    label = torch.randint(0, K, (B,), generator=g)
    for it in range(6):
        logits = torch.randn(B, K, generator=g)
        logits[0, 3] = logits[0, 7] = 50.0  # tie -> lowest index
        if it % 2:
            logits[1, label[1]] = 100.0  # still classified correctly -> no update for sample 1
        P = torch.randn(B, C, 3, generator=g) * (1.0 / (it + 1))
        sig = torch.rand(B, C, generator=g)
        adv = torch.randn(B, 3, Nn, generator=g)
        A.best_update(cu(logits), cu(label), cu(P), cu(sig), cu(adv), dst)
        pred = logits.argmax(1)
        dist = O.transformation_loss(P, sig, C, batch_avg=False)
        for e in range(B):
            if pred[e] != label[e]:
                if dist[e] < st['bestdist'][e]:
                    st['bestdist'][e], st['bestscore'][e] = dist[e], pred[e]
                if dist[e] < st['o_bestdist'][e]:
                    st['o_bestdist'][e], st['o_bestscore'][e] = dist[e], pred[e]
                    st['o_bestattack'][e] = adv[e]
        assert torch.equal(dst['pred'].cpu(), pred)
        close(dst['dist_val'], dist, rtol=1e-6)
        assert torch.equal(dst['bestscore'].cpu(), st['bestscore'])
        assert torch.equal(dst['o_bestscore'].cpu(), st['o_bestscore'])
        assert torch.equal(dst['o_bestattack'].cpu(), st['o_bestattack'])
        close(dst['o_bestdist'], st['o_bestdist'], rtol=1e-6)

    p = torch.randn(3, 7, 3, generator=g).requires_grad_()
    s = torch.rand(3, 7, generator=g).requires_grad_()
    opt = torch.optim.Adam([{'params': p, 'lr': 0.05}, {'params': s, 'lr': 0.03}], weight_decay=0.)
    dp, ds = p.detach().clone().cuda(), s.detach().clone().cuda()
    m_p, v_p, m_s, v_s = (torch.zeros_like(t) for t in (dp, dp, ds, ds))
    step = torch.zeros(1, dtype=torch.int32, device='cuda')
    for it in range(25):
        gp, gs = torch.randn(3, 7, 3, generator=g), torch.randn(3, 7, generator=g) * 1e-3
        p.grad, s.grad = gp.clone(), gs.clone()
        opt.step()
        A.adam_step(dp, ds, cu(gp), cu(gs), m_p, v_p, m_s, v_s, step, 0.05, 0.03)
    assert step.item() == 25
    close(dp, p, rtol=1e-5, atol=1e-6)
    close(ds, s, rtol=1e-5, atol=1e-6)


def test_adam_step_sum_and_projection(A):
    """hitadv_adam_step_sum == torch.optim.Adam on the summed gradient followed by the reference's clamp."""
    g = torch.Generator().manual_seed(8)
    p = (torch.randn(3, 7, 3, generator=g) * 0.1).requires_grad_()
    s = (0.1 + torch.rand(3, 7, generator=g)).requires_grad_()
    opt = torch.optim.Adam([{'params': p, 'lr': 0.05}, {'params': s, 'lr': 0.03}], weight_decay=0.)
    dp, ds = p.detach().clone().cuda(), s.detach().clone().cuda()
    m_p, v_p, m_s, v_s = (torch.zeros_like(t) for t in (dp, dp, ds, ds))
    step = torch.zeros(1, dtype=torch.int32, device='cuda')
    for it in range(25):
        gp, gp2 = torch.randn(3, 7, 3, generator=g), torch.randn(3, 7, 3, generator=g)
        gs, gs2 = torch.randn(3, 7, generator=g), torch.randn(3, 7, generator=g)
        p.grad, s.grad = gp + gp2, gs + gs2
        opt.step()
        with torch.no_grad():
            p.clamp_(-0.3, 0.3)
            s.clamp_(0.1, 1.2)
        step += 1  # what hitadv_best_update does once per iteration
        A.adam_step_sum(dp, ds, cu(gp), cu(gp2), cu(gs), cu(gs2) if it % 2 else None, m_p, v_p, m_s, v_s, step, 0.05,
                        0.03, (-0.3, 0.3), (0.1, 1.2)) if it % 2 else \
            A.adam_step_sum(dp, ds, cu(gp + gp2), None, cu(gs), cu(gs2), m_p, v_p, m_s, v_s, step, 0.05, 0.03,
                            (-0.3, 0.3), (0.1, 1.2))
        close(dp, p, rtol=1e-5, atol=1e-6)
        close(ds, s, rtol=1e-5, atol=1e-6)
    assert step.item() == 25 and float(dp.abs().max()) <= float(np.float32(0.3))


def test_adam_single_matches_torch_adam(A):
    """The one-tensor Adam the captured CW loops use (hitadv_adam_step with an empty second group) against
    torch.optim.Adam's defaults over 30 steps."""
    g = torch.Generator().manual_seed(5)
    p0 = torch.randn(3, 3, 257, generator=g)
    grads = [torch.randn(3, 3, 257, generator=g) * (0.1 + i % 3) for i in range(30)]
    ref = p0.clone().requires_grad_()
    opt = torch.optim.Adam([ref], lr=1e-2, weight_decay=0.)
    p = p0.clone().cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    step = torch.zeros(1, device='cuda', dtype=torch.int32)
    for gr in grads:
        ref.grad = gr.clone()
        opt.step()
        A.adam_single(p, gr.cuda(), m, v, step, 1e-2)
    assert int(step.item()) == 30
    close(p, ref.detach(), rtol=2e-6, atol=2e-7)


@pytest.mark.parametrize("B,K", [(32, 40), (5, 16), (1, 40), (70, 130)])
def test_fused_adv_losses_match_reference_modules(A, B, K):
    """hitadv_adv_loss (value + gradient in one launch) vs the torch formulation of util/adv_utils.py under autograd,
    which the oracle tests pin to the reference (fixture g6)."""
    from hit_adv_amd.util.adv_utils import CrossEntropyAdvLoss, LogitsAdvLoss, UntargetedLogitsAdvLoss
    g = torch.Generator().manual_seed(B + K)
    logits = torch.randn(B, K, generator=g) * 8
    target = torch.randint(0, K, (B,), generator=g)
    logits[0, target[0]] = 100.  # confidently right: untargeted margin active, targeted inactive
    for mod in (UntargetedLogitsAdvLoss(30.), UntargetedLogitsAdvLoss(0.), LogitsAdvLoss(5.), CrossEntropyAdvLoss()):
        z = logits.clone().requires_grad_()
        ref = mod(z, target)
        gref, = torch.autograd.grad(ref, z)
        out = torch.zeros((), device='cuda')
        loss, d = mod.fused(cu(logits), cu(target), loss_out=out)
        assert loss.data_ptr() == out.data_ptr()
        close(loss, ref, rtol=2e-6, atol=1e-6)
        close(d, gref, rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------ full-size properties (cfg2 sizes)
def test_full_size_properties_b32_n1024(A):
    from hit_adv_amd.pytorch3d_ops import knn_points
    from hit_adv_amd.util.set_distance import chamfer, hausdorff
    x, _ = clouds(32, 1024, 300)
    y, _ = clouds(32, 1024, 400)
    xc, yc = cu(x), cu(y)
    # symmetry: swapping the arguments swaps the two outputs, bit for bit
    a1, a2 = chamfer(xc, yc)
    b1, b2 = chamfer(yc, xc)
    assert torch.equal(a1, b2) and torch.equal(a2, b1)
    h1, h2 = hausdorff(xc, yc)
    assert (h1 >= a1).all() and (h2 >= a2).all()  # max of the minima dominates their mean
    # a cloud against itself: zero distance, identity arg-min (no duplicate points in these clouds)
    mx, ax, _, _ = A.nn_min(xc, xc)
    assert (mx == 0).all() and torch.equal(ax, torch.arange(1024, device='cuda', dtype=torch.int32).expand(32, -1))
    # materialised matrix reductions == fused reductions, bit for bit (direct form)
    P = A.pairwise_sqdist(xc, yc, A.FORM_DIRECT)
    m1, _, m2, _ = A.nn_min(xc, yc)
    assert torch.equal(P.min(2).values, m1) and torch.equal(P.min(1).values, m2)
    # kNN: ascending, rank 0 is the point itself, and rank-1 distance equals the masked NN-min
    r = knn_points(xc, xc, K=17)
    assert (r.dists[..., 1:] >= r.dists[..., :-1]).all() and (r.dists[..., 0] == 0).all()
    assert torch.equal(r.idx[..., 0], torch.arange(1024, device='cuda').expand(32, -1))
    Pxx = A.pairwise_sqdist(xc, xc, A.FORM_DIRECT)
    Pxx.diagonal(dim1=1, dim2=2).fill_(float('inf'))
    assert torch.equal(Pxx.min(2).values, r.dists[..., 1])
    # FPS: no repeated index on clouds without duplicates; first index is the given start
    start = torch.arange(32, device='cuda') * 7
    f = A.fps_from_start(xc, 256, start)
    assert torch.equal(f[:, 0], start)
    assert all(len(set(row.tolist())) == 256 for row in f.cpu())


# ------------------------------------------------------------------ victim-side helpers
def test_max_over_points_and_linear_max_bwd(A):
    g = torch.Generator().manual_seed(7)
    B, Np, Cin, Cout = 3, 500, 128, 1024
    x = torch.randn(B * Np, Cin, generator=g)
    W = torch.randn(Cout, Cin, generator=g) * 0.1
    y = x @ W.t()
    y[5, 17] = y[300, 17] = 99.0  # tie inside cloud 0 -> lowest point index wins
    val, idx = A.max_over_points(cu(y), B, Np)
    ref_val, ref_idx = y.view(B, Np, Cout).max(dim=1)
    assert torch.equal(val.cpu(), ref_val)
    assert idx.dtype == torch.int64 and idx[0, 17].item() == 5
    same = torch.ones(B, Cout, dtype=torch.bool)
    same[0, 17] = False
    assert torch.equal(idx.cpu()[same], ref_idx[same])
    # backward: dX[b,n,:] = sum_{j: idx[b,j]==n} dg[b,j] W[j,:]  vs dense autograd through max
    dg = torch.randn(B, Cout, generator=g)
    xr = x.clone().requires_grad_()
    (xr @ W.t()).view(B, Np, Cout).gather(1, idx.cpu().unsqueeze(1)).squeeze(1).mul(dg).sum().backward()
    dx = A.linear_max_bwd(cu(dg), cu(W), idx, Np)
    close(dx, xr.grad, rtol=1e-5, atol=1e-5)
    assert torch.equal(dx, A.linear_max_bwd(cu(dg), cu(W), idx, Np))
    # bias + ReLU in the merge pass, activation mask in the backward
    bias = torch.randn(Cout, generator=g)
    val2, idx2 = A.max_over_points(cu(y), B, Np, bias=cu(bias), relu=True)
    assert torch.equal(idx2, idx) and torch.equal(val2.cpu(), (ref_val + bias).clamp_min(0.))
    dxm = A.linear_max_bwd(cu(dg), cu(W), idx, Np, act_out=val2)
    xr2 = x.clone().requires_grad_()
    # (channel (0,17) carries the planted tie value in y only, so take the ReLU mask from the kernel's own output)
    ((xr2 @ W.t()).view(B, Np, Cout).gather(1, idx.cpu().unsqueeze(1)).squeeze(1) * (dg * (val2.cpu() > 0))).sum().backward()
    close(dxm, xr2.grad, rtol=1e-5, atol=1e-5)
    # a single hot point owning every channel (long serial chain) and a small odd Cin
    hot = torch.zeros(B, Cout, dtype=torch.int64)
    W2 = torch.randn(Cout, 70, generator=g)
    dx = A.linear_max_bwd(cu(dg), cu(W2), cu(hot), Np).cpu().view(B, Np, 70)
    close(dx[:, 0], dg @ W2, rtol=1e-4, atol=1e-4)
    assert (dx[:, 1:] == 0).all()


@pytest.mark.parametrize("B,Np,Cin,Cout", [(3, 500, 128, 1024), (2, 1024, 128, 1024), (5, 77, 64, 192), (1, 64, 128, 64),
                                           (32, 1024, 128, 1024)])
def test_linear_max_fwd_mfma(A, B, Np, Cin, Cout):
    """Fused shared layer + max over points (f32 MFMA) vs  (x @ Wt).view(B,N,C).max(1)  evaluated in float64."""
    g = torch.Generator().manual_seed(B * 1000 + Np)
    x = torch.randn(B * Np, Cin, generator=g)
    Wt = torch.randn(Cin, Cout, generator=g) * 0.1
    bias = torch.randn(Cout, generator=g)
    y = (x.double() @ Wt.double()).view(B, Np, Cout)
    ref_val, ref_idx = y.max(dim=1)
    val, idx = A.linear_max_fwd(cu(x), cu(Wt), B, Np)
    assert idx.dtype == torch.int64 and val.shape == (B, Cout)
    close(val, ref_val.float(), rtol=2e-6, atol=2e-5)
    # the arg-max is the reference's wherever the runner-up is further away than f32 noise ...
    top2 = y.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4
    assert clear.float().mean() > 0.9
    assert torch.equal(idx.cpu()[clear], ref_idx[clear])
    # ... and everywhere it points at a point whose activation IS the reported maximum
    close(y.gather(1, idx.cpu().unsqueeze(1)).squeeze(1).float(), val, rtol=2e-6, atol=2e-5)
    # bias + ReLU in the merge pass; bitwise reproducible; agrees with the unfused HIP path's indices where clear
    val2, idx2 = A.linear_max_fwd(cu(x), cu(Wt), B, Np, bias=cu(bias), relu=True)
    assert torch.equal(idx2, idx) and torch.equal(val2, (val + cu(bias)).clamp_min(0.))
    v3, i3 = A.linear_max_fwd(cu(x), cu(Wt), B, Np)
    assert torch.equal(v3, val) and torch.equal(i3, idx)
    _, iu = A.max_over_points(cu(x) @ cu(Wt), B, Np)
    assert torch.equal(iu.cpu()[clear], idx.cpu()[clear])


@pytest.mark.parametrize("B,Np,Cin,Cout", [(32, 1024, 128, 1024), (3, 1000, 128, 1024), (2, 130, 64, 256), (5, 64, 128, 320)])
def test_linear_max_fwd_bf16x3_is_fp32_accurate(A, B, Np, Cin, Cout):
    """The same operator on the bf16 matrix cores with both operands split into three bf16 pieces (csrc/victim_bf3.hip):
    the pieces sum to the fp32 value EXACTLY, and the result is as close to float64 as the f32-MFMA kernel's -- nothing is
    rounded to bf16 precision.  Achieved errors of both forms are recorded side by side."""
    g = torch.Generator().manual_seed(B * 1000 + Np)
    x = torch.randn(B * Np, Cin, generator=g).relu()  # the layer's input is a ReLU output
    Wt = torch.randn(Cin, Cout, generator=g) * 0.1
    bias = torch.randn(Cout, generator=g)
    W3 = A.split_weights_bf16x3(cu(Wt.t().contiguous()))
    # fragment order [piece][c/16][k/32][(k%32)/8][c%16][k%8] -> [piece][c][k]
    pieces = W3.cpu().view(torch.bfloat16).float().view(3, Cout // 16, Cin // 32, 4, 16, 8).permute(0, 1, 4, 2, 3, 5)
    pieces = pieces.reshape(3, Cout, Cin)
    assert torch.equal(pieces[0] + pieces[1] + pieces[2], Wt.t())  # exact three-way split
    y = (x.double() @ Wt.double()).view(B, Np, Cout)
    ref_val, ref_idx = y.max(dim=1)
    val, idx = A.linear_max_fwd_bf16x3(cu(x), W3, B, Np)
    f32_val, f32_idx = A.linear_max_fwd(cu(x), cu(Wt), B, Np)
    close(val, ref_val.float(), rtol=2e-6, atol=2e-5, what='bf16x3 max vs float64')
    close(f32_val, ref_val.float(), rtol=2e-6, atol=2e-5, what='f32 MFMA max vs float64')
    e3, e1 = (val.cpu().double() - ref_val).abs().max().item(), (f32_val.cpu().double() - ref_val).abs().max().item()
    assert e3 <= 2 * e1 + 1e-6, (e3, e1)  # no worse than the exact-f32 kernel (both are bounded by the fp32 accumulation)
    top2 = y.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4
    assert torch.equal(idx.cpu()[clear], ref_idx[clear]) and torch.equal(idx.cpu()[clear], f32_idx.cpu()[clear])
    close(y.gather(1, idx.cpu().unsqueeze(1)).squeeze(1).float(), val, rtol=2e-6, atol=2e-5, what='bf16x3 arg-max row')
    val2, idx2 = A.linear_max_fwd_bf16x3(cu(x), W3, B, Np, bias=cu(bias), relu=True)
    assert torch.equal(idx2, idx) and torch.equal(val2, (val + cu(bias)).clamp_min(0.))
    v3, i3 = A.linear_max_fwd_bf16x3(cu(x), W3, B, Np)
    assert torch.equal(v3, val) and torch.equal(i3, idx)  # bitwise reproducible


@pytest.mark.parametrize("B,Np,Cin,Cout", [(32, 1024, 128, 1024), (3, 1000, 128, 1024), (2, 130, 64, 256), (5, 64, 128, 320)])
@pytest.mark.parametrize("scale", [1.0, 1e-3, 40.0])
def test_linear_max_fwd_f16x2_error_is_at_fp32_roundoff(A, B, Np, Cin, Cout, scale):
    """The same operator on the fp16 matrix cores: two fp16 pieces per operand (the second scaled by 2^11), three exact
    products into two fp32 accumulator sets.  The pieces reproduce every weight to 2^-23 relative (half an fp32 ulp), and the
    result is as close to float64 as the f32-MFMA kernel's up to a factor that is asserted here (achieved figures of both
    forms are recorded side by side), for activations of unit scale, of 1e-3 (fp16 subnormal first pieces) and of 40."""
    g = torch.Generator().manual_seed(B * 1000 + Np)
    x = torch.randn(B * Np, Cin, generator=g).relu() * scale
    Wt = torch.randn(Cin, Cout, generator=g) * 0.1
    bias = torch.randn(Cout, generator=g)
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    W2 = A.split_weights_f16x2(cu(Wt.t().contiguous()), range_flag=flag)
    pieces = W2.cpu().view(torch.float16).double().view(2, Cout // 16, Cin // 32, 4, 16, 8).permute(0, 1, 4, 2, 3, 5).reshape(2, Cout, Cin)
    back, want = pieces[0] + pieces[1] / 2048., Wt.t().double()
    normal = want.abs() >= 1e-4  # first piece a normal fp16 number: half an fp32 ulp; below that, 2^-36 absolute
    assert float(((back - want).abs() / want.abs())[normal].max()) <= 2.0 ** -22 and float((back - want).abs().max()) <= 2.0 ** -22 * 0.5
    y = (x.double() @ Wt.double()).view(B, Np, Cout)
    ref_val, ref_idx = y.max(dim=1)
    val, idx = A.linear_max_fwd_f16x2(cu(x), W2, B, Np, range_flag=flag)
    f32_val, f32_idx = A.linear_max_fwd(cu(x), cu(Wt), B, Np)
    s = float(ref_val.abs().max())
    close(val / s, ref_val.float() / s, rtol=0, atol=2e-6, what='fp16x2 max vs float64 (over the output scale)')
    close(f32_val / s, ref_val.float() / s, rtol=0, atol=2e-6, what='f32 MFMA max vs float64 (over the output scale)')
    e2, e1 = (val.cpu().double() - ref_val).abs().max().item(), (f32_val.cpu().double() - ref_val).abs().max().item()
    assert e2 <= 2 * e1 + 1e-7 * s, (e2, e1)
    top2 = y.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4 * s
    assert torch.equal(idx.cpu()[clear], ref_idx[clear]) and torch.equal(idx.cpu()[clear], f32_idx.cpu()[clear])
    val2, idx2 = A.linear_max_fwd_f16x2(cu(x), W2, B, Np, bias=cu(bias), relu=True, blocks=128)
    assert torch.equal(idx2, idx) and torch.equal(val2, (val + cu(bias)).clamp_min(0.))  # grid-independent, bias / ReLU in the merge
    assert int(flag.item()) == 0


@pytest.mark.parametrize("G,ns,Cin,Cout", [(70, 32, 64, 128), (33, 64, 128, 256), (5, 32, 128, 128), (1, 32, 64, 128),
                                          (2049, 32, 64, 128), (300, 64, 128, 128)])
def test_group_linear_max_forward_and_backward(A, G, ns, Cin, Cout):
    """The last shared layer of a sample-and-group block fused with the max over the neighbours (csrc/group_mlp.hip) against
    float64: values at fp32's roundoff (the fp16x2 scheme of V1), arg-max = the float64 winner wherever the runner-up is not
    within rounding, ties -> the lowest row; the backward (the routed gradient through W on the matrix cores) against the
    float64 gradient of the same function with the kernel's own winners."""
    g = torch.Generator().manual_seed(G * 7 + ns)
    x = torch.randn(G, ns, Cin, generator=g).relu()
    x[0, 5] = x[0, 2]  # duplicate rows: the lower one must win every channel it wins
    Wr = torch.randn(Cout, Cin, generator=g) * 0.1
    bias = torch.randn(Cout, generator=g) * 0.3
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    xg = cu(x).requires_grad_()
    out, arg = A.group_linear_max(xg, cu(Wr), cu(bias), flag, return_arg=True)
    y = x.double() @ Wr.double().t() + bias.double()          # [G,ns,Cout]
    ref = y.max(dim=1).values.clamp_min(0.)
    s = float(ref.abs().max())
    close(out / s, ref.float() / s, rtol=0, atol=2e-6, what='group_linear_max vs float64 (over the output scale)')
    top2 = y.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4 * s
    assert torch.equal(arg.cpu().long()[clear], y.argmax(dim=1)[clear])
    assert int((arg[0] == 5).sum()) == 0 and arg.dtype == torch.int32 and int(arg.min()) >= 0 and int(arg.max()) < ns
    # backward: d/dx of sum(out * w) with the kernel's winners
    wgt = torch.randn(G, Cout, generator=g)
    (out * cu(wgt)).sum().backward()
    gd = torch.zeros(G, ns, Cin, dtype=torch.float64)
    routed = (wgt.double() * (out.detach().cpu().double() > 0))  # [G,Cout]
    contrib = routed.unsqueeze(-1) * Wr.double().unsqueeze(0)     # [G,Cout,Cin]
    gd.scatter_add_(1, arg.cpu().long().unsqueeze(-1).expand(-1, -1, Cin), contrib)
    sg = float(gd.abs().max())
    close(xg.grad / sg, gd.float() / sg, rtol=0, atol=2e-6, what='group_linear_max backward vs float64 (over the gradient scale)')
    # the torch composition gives the same function
    xt = cu(x).requires_grad_()
    ot = torch.relu(torch.nn.functional.linear(xt, cu(Wr), cu(bias))).max(dim=1)[0]
    close(out, ot, rtol=0, atol=4e-6 * s, what='group_linear_max vs the fp32 torch composition')
    assert int(flag.item()) == 0
    out2, arg2 = A.group_linear_max(cu(x), cu(Wr), cu(bias), flag, return_arg=True)
    assert torch.equal(out2, out) and torch.equal(arg2, arg)  # bitwise reproducible
    # relu_input: the same gradient gated by (x > 0) on its way out of the kernel (x is a ReLU output: a third of it is zero)
    xm = cu(x).requires_grad_()
    om = A.group_linear_max(xm, cu(Wr), cu(bias), flag, relu_input=True)
    (om * cu(wgt)).sum().backward()
    assert torch.equal(om, out) and torch.equal(xm.grad, torch.where(xm.detach() > 0, xg.grad, torch.zeros_like(xg.grad)))
    assert float((x > 0).float().mean()) < 0.7


@pytest.mark.parametrize("B,N,S,ns,C,Cout", [(3, 300, 64, 32, 64, 64), (2, 128, 32, 64, 128, 128), (5, 77, 8, 16, 64, 128)])
def test_group_add_relu_linear_equals_the_two_calls(A, B, N, S, ns, C, Cout):
    """hitadv_group_add_relu_linear (the gather / add / ReLU of a block's first layer inside the middle layer's kernel) gives the bits
    of group_add_relu followed by rows_linear, and its autograd node the gradients of the two nodes chained; an index outside the
    cloud yields a zero row, as in group_add_relu."""
    g = torch.Generator().manual_seed(B * 100 + ns)
    U = torch.randn(B, N, C, generator=g)
    V = torch.randn(B, S, C, generator=g)
    idx = torch.randint(0, N, (B, S, ns), generator=g)
    idx[0, 0, 3] = -1
    idx[B - 1, S - 1, ns - 1] = N + 5
    Wr = torch.randn(Cout, C, generator=g) * 0.1
    bias = torch.randn(Cout, generator=g) * 0.3
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    W2, Wt2 = A.split_weights_f16x2(cu(Wr), range_flag=flag), A.split_weights_f16x2(cu(Wr.t().contiguous()), range_flag=flag)
    assert A.group_add_relu_linear_supported(C, Cout, S, ns)
    Ua, Va = cu(U).requires_grad_(), cu(V).requires_grad_()
    y = A.GroupAddReLULinear.apply(Ua, Va, cu(idx), W2, Wt2, cu(bias), flag)
    Ub, Vb = cu(U).requires_grad_(), cu(V).requires_grad_()
    H = A.group_add_relu(Ub, Vb, cu(idx))
    y2 = A.rows_linear(H.detach().reshape(-1, C), W2, cu(bias), True, flag).view(B, S, ns, Cout)
    assert torch.equal(y, y2)
    assert torch.equal(y.detach()[0, 0, 3], torch.relu(cu(bias))) and int(flag.item()) == 0  # an index outside the cloud: a zero row in
    gy = torch.randn(B, S, ns, Cout, generator=g)
    y.backward(cu(gy))
    dH = A.rows_linear(cu(gy).reshape(-1, Cout), Wt2, None, False, flag).view(B, S, ns, C)
    H.backward(dH)
    assert torch.equal(Ua.grad, Ub.grad) and torch.equal(Va.grad, Vb.grad)
    # against float64 (the composition itself)
    ok = (idx >= 0) & (idx < N)
    Hd = torch.relu(torch.gather(U.double(), 1, idx.clamp(0, N - 1).view(B, S * ns, 1).expand(-1, -1, C)).view(B, S, ns, C)
                    + V.double().unsqueeze(2)) * ok.unsqueeze(-1)
    ref = torch.relu(Hd @ Wr.double().t() + bias.double())
    s_ = float(ref.abs().max())
    close(y / s_, ref.float() / s_, rtol=0, atol=2e-6, what='group_add_relu_linear vs float64 over the output scale')


def test_points_major_is_the_permuted_copy(A):
    """ops.points_major = x.permute(0, 2, 1).contiguous() for a cloud [B,3,N] (hitadv_transpose_small), forward and backward."""
    g = torch.Generator().manual_seed(3)
    for B, C, N in ((5, 3, 1000), (2, 6, 77), (64, 3, 2048)):
        x = torch.randn(B, C, N, generator=g)
        xg = cu(x).requires_grad_()
        y = A.points_major(xg)
        assert y.is_contiguous() and torch.equal(y.cpu(), x.permute(0, 2, 1).contiguous())
        w = torch.randn(B, N, C, generator=g)
        (y * cu(w)).sum().backward()
        assert torch.equal(xg.grad.cpu(), w.permute(0, 2, 1).contiguous())


@pytest.mark.parametrize("rows,Cin,Cout", [(64 * 41, 64, 64), (64 * 300 + 17, 128, 128), (5, 64, 128), (64 * 9 + 63, 128, 64),
                                            (64 * 2500, 64, 64)])
def test_rows_linear_is_fp32_accurate(A, rows, Cin, Cout):
    """The middle shared layer of a sample-and-group block over very many rows (csrc/rows_linear.hip, fp16x2 arithmetic) against
    float64: errors at fp32's roundoff, no worse than torch's f32 GEMM; whole and ragged last tiles, one and many blocks; with the
    transposed pieces, no bias and no activation it is the layer's input gradient; the autograd node of model/_pointwise.py."""
    g = torch.Generator().manual_seed(rows + Cin)
    x = torch.randn(rows, Cin, generator=g).relu()
    Wr = torch.randn(Cout, Cin, generator=g) * 0.1
    bias = torch.randn(Cout, generator=g) * 0.3
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    W2, Wt2 = A.split_weights_f16x2(cu(Wr), range_flag=flag), A.split_weights_f16x2(cu(Wr.t().contiguous()), range_flag=flag)
    assert A.rows_linear_supported(Cin, Cout)
    y = A.rows_linear(cu(x), W2, cu(bias), True, flag)
    ref = (x.double() @ Wr.double().t() + bias.double()).clamp_min(0.)
    s = float(ref.abs().max())
    e = float((y.cpu().double() - ref).abs().max()) / s
    e32 = float((torch.relu(torch.nn.functional.linear(cu(x), cu(Wr), cu(bias))).cpu().double() - ref).abs().max()) / s
    note('rows_linear %dx%d->%d: max error over the output scale (torch f32 GEMM: %.2e)' % (rows, Cin, Cout, e32), e)
    assert e <= 2e-6 and e <= 2 * e32 + 1e-7
    assert torch.equal(y, A.rows_linear(cu(x), W2, cu(bias), True, flag))  # the same bits every run
    # input gradient of the layer: dX = dY W
    gy = torch.randn(rows, Cout, generator=g)
    dx = A.rows_linear(cu(gy), Wt2, None, False, flag)
    rd = gy.double() @ Wr.double()
    sd = float(rd.abs().max())
    assert float((dx.cpu().double() - rd).abs().max()) / sd <= 2e-6
    assert int(flag.item()) == 0
    # the autograd node (the ReLU backward of this layer is left to its consumer: the gradient arrives gated)
    from hit_adv_amd.model import _pointwise
    xg = cu(x).requires_grad_()
    yy = _pointwise._RowsLinearReLUGatedLater.apply(xg, (W2, Wt2), cu(bias), flag)
    yy.backward(cu(gy))
    assert torch.equal(yy, y) and torch.equal(xg.grad, dx)
    # an entry beyond the split's range raises the flag (and only then)
    xb = x.clone()
    xb[rows // 2, 3] = 7e4
    A.rows_linear(cu(xb), W2, cu(bias), True, flag)
    assert int(flag.item()) == 1


@pytest.mark.parametrize("G,ns,Cin,Cout", [(300, 32, 256, 256), (70, 64, 128, 384), (9, 32, 128, 128)])
def test_group_linear_max_on_the_gemm_core(A, G, ns, Cin, Cout):
    """The same fused layer on the tiled GEMM core (the widths the register-resident kernels do not cover): values, winners
    (lowest row on ties), the routed gradient, the gradient gated by the ReLU of the layer in front, reproducibility -- and,
    where both forms exist, the same winners and values to fp32 roundoff."""
    g = torch.Generator().manual_seed(G * 3 + ns + Cout)
    x = torch.randn(G, ns, Cin, generator=g).relu()
    x[0, 5] = x[0, 2]
    Wr = torch.randn(Cout, Cin, generator=g) * 0.1
    bias = torch.randn(Cout, generator=g) * 0.3
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    assert A.group_linear_max_g16_supported(Cin, Cout, ns)
    Wp, Wtp = A.split_rows_f16x2(cu(Wr), flag), A.split_rows_f16x2(cu(Wr.t().contiguous()), flag)
    xg = cu(x).requires_grad_()
    out, arg = A.group_linear_max_g16(xg, Wp, Wtp, cu(bias), flag, return_arg=True)
    y = x.double() @ Wr.double().t() + bias.double()
    ref = y.max(dim=1).values.clamp_min(0.)
    s = float(ref.abs().max())
    close(out / s, ref.float() / s, rtol=0, atol=2e-6, what='group_linear_max_g16 vs float64 (over the output scale)')
    top2 = y.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4 * s
    assert torch.equal(arg.cpu().long()[clear], y.argmax(dim=1)[clear])
    assert int((arg[0] == 5).sum()) == 0 and int(arg.min()) >= 0 and int(arg.max()) < ns
    wgt = torch.randn(G, Cout, generator=g)
    (out * cu(wgt)).sum().backward()
    gd = torch.zeros(G, ns, Cin, dtype=torch.float64)
    contrib = (wgt.double() * (out.detach().cpu().double() > 0)).unsqueeze(-1) * Wr.double().unsqueeze(0)
    gd.scatter_add_(1, arg.cpu().long().unsqueeze(-1).expand(-1, -1, Cin), contrib)
    sg = float(gd.abs().max())
    close(xg.grad / sg, gd.float() / sg, rtol=0, atol=2e-6, what='group_linear_max_g16 backward vs float64 (over the gradient scale)')
    xm = cu(x).requires_grad_()
    om, am = A.group_linear_max_g16(xm, Wp, Wtp, cu(bias), flag, return_arg=True, relu_input=True)
    (om * cu(wgt)).sum().backward()
    assert torch.equal(om, out) and torch.equal(am, arg)
    assert torch.equal(xm.grad, torch.where(xm.detach() > 0, xg.grad, torch.zeros_like(xg.grad)))
    if A.group_linear_max_supported(Cin, Cout, ns):
        o2, a2 = A.group_linear_max(cu(x), cu(Wr), cu(bias), flag, return_arg=True)
        close(o2, out, rtol=0, atol=2e-6 * s)
        assert torch.equal(a2.cpu()[clear], arg.cpu()[clear])
    assert int(flag.item()) == 0


def test_linear_max_fwd_f16x2_raises_its_range_flag(A):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2 * 64, 128, generator=g).relu()
    W2 = A.split_weights_f16x2(cu(torch.randn(256, 128, generator=g)))
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    A.linear_max_fwd_f16x2(cu(x), W2, 2, 64, range_flag=flag)
    assert int(flag.item()) == 0
    x[70, 3] = 1e5
    A.linear_max_fwd_f16x2(cu(x), W2, 2, 64, range_flag=flag)
    assert int(flag.item()) == 1
    flag.zero_()
    A.split_weights_f16x2(cu(torch.full((16, 32), 7e4)), range_flag=flag)
    assert int(flag.item()) == 1


@pytest.mark.parametrize("B,T,J,K,NOUT", [(32, 1, 40, 256, 512), (32, 16, 9, 256, 512), (5, 3, 64, 200, 70), (33, 2, 7, 130, 33)])
def test_fc_layer_with_its_input_evaluated_on_the_way_in(A, B, T, J, K, NOUT):
    """hitadv_fc_layer_pre == sum_partials -> fc_layer -> fc_layer(mask) to fp32 rounding (the small product is summed in a
    different order), against float64; bitwise reproducible."""
    g = torch.Generator().manual_seed(B * 100 + J)
    pre = torch.randn(B, T, J, generator=g)
    Wpre = torch.randn(J, K, generator=g) * 0.3
    Wt = torch.randn(K, NOUT, generator=g) * 0.1
    mask = torch.randn(B, K, generator=g)
    bias = torch.randn(NOUT, generator=g)
    x64 = (pre.double().sum(1) @ Wpre.double()) * (mask > 0)
    ref = (x64 @ Wt.double() + bias.double()).clamp_min(0.)
    out = A.fc_layer_pre(cu(pre), cu(Wpre), cu(Wt), bias=cu(bias), relu=True, mask=cu(mask))
    close(out, ref.float(), rtol=2e-5, atol=2e-5, what='fc_layer_pre vs float64')
    three = A.fc_layer(A.fc_layer(A.sum_partials(cu(pre)), cu(Wpre)), cu(Wt), bias=cu(bias), relu=True, mask=cu(mask))
    close(out, three, rtol=2e-5, atol=2e-5, what='fc_layer_pre vs the three launches')
    assert torch.equal(out, A.fc_layer_pre(cu(pre), cu(Wpre), cu(Wt), bias=cu(bias), relu=True, mask=cu(mask)))
    nomask = A.fc_layer_pre(cu(pre), cu(Wpre), cu(Wt))
    close(nomask, (pre.double().sum(1) @ Wpre.double() @ Wt.double()).float(), rtol=2e-5, atol=2e-5)


def test_linear_max_fwd_bf16x3_same_bits_for_every_grid(A):
    """The ``blocks`` argument only changes how the work is spread (point splits merged in point order, or several clouds
    per block): values and arg-max are the same bits, ties included."""
    g = torch.Generator().manual_seed(5)
    B, Np, Cin, Cout = 32, 1000, 128, 1024
    x = torch.randn(B * Np, Cin, generator=g).relu()
    x[Np + 7] = x[Np + 3]  # a duplicate point: equal activations, the lower index must win everywhere
    W3 = A.split_weights_bf16x3(cu(torch.randn(Cout, Cin, generator=g) * 0.1))
    bias = cu(torch.randn(Cout, generator=g))
    ref = None
    for blocks in (0, 256, 128, 64, 40, 8):
        out = A.linear_max_fwd_bf16x3(cu(x), W3, B, Np, bias=bias, relu=True, blocks=blocks)
        if ref is None:
            ref = out
        assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]), blocks
    with pytest.raises(RuntimeError):
        A.linear_max_fwd_bf16x3(cu(x), W3, B, Np, blocks=5)


@pytest.mark.parametrize("mode", ["bf16x3", "fp16x2", "packed"])
def test_linear_max_flat_stream_equals_the_split_form(A, mode):
    """N a multiple of 128 and no point split selects V1's FLAT instantiation (the clouds of a workgroup as one stream of tiles, the
    running maximum finished every N / 64 tiles); few clouds on many workgroups select the split form (merged in point order).
    Same bits either way, an odd number of clouds over workgroups that take 2, 4 and 7 of them, ties included."""
    g = torch.Generator().manual_seed(11)
    B, Np, Cin, Cout = 13, 1280, 128, 1024
    x = torch.randn(B * Np, Cin, generator=g).relu()
    x[3 * Np + 700] = x[3 * Np + 5]   # duplicate points in different tiles of one cloud: the lower index wins
    x[12 * Np + 64] = x[12 * Np + 63]  # ... and across a tile boundary of the last cloud
    Wr = cu(torch.randn(Cout, Cin, generator=g) * 0.1)
    bias = cu(torch.randn(Cout, generator=g))
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    if mode == "bf16x3":
        W = A.split_weights_bf16x3(Wr)
        run = lambda blocks: A.linear_max_fwd_bf16x3(cu(x), W, B, Np, bias=bias, relu=True, blocks=blocks)
    else:
        W = A.split_weights_f16x2(Wr, range_flag=flag)
        xin = cu(x)
        if mode == "packed":  # one word per value: fp16 hi | fp16 lo << 16 (what V2 hands over)
            hi = xin.half()
            lo = ((xin - hi.float()) * 2048.).half()
            xin = (hi.view(torch.int16).int() & 0xffff | (lo.view(torch.int16).int() << 16)).view(torch.float32)
        run = lambda blocks: A.linear_max_fwd_f16x2(xin, W, B, Np, bias=bias, relu=True, blocks=blocks, range_flag=flag,
                                                     packed=(mode == "packed"))
    ref = run(0)  # 52 (cloud, column group) pairs on 256 workgroups: five point splits per cloud, the split form
    for blocks in (256, 128, 40, 16, 8):  # 40 and fewer: several clouds per workgroup, the FLAT form
        out = run(blocks)
        assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]), blocks
    y = (x.double() @ Wr.cpu().double().t()).view(B, Np, Cout)
    s = float(y.abs().max())
    close(ref[0] / s, (y.max(dim=1).values + bias.cpu().double()).clamp_min(0.).float() / s, rtol=0, atol=2e-6,
          what='V1 (%s) vs float64 over the output scale' % mode)
    assert int(ref[1][3].eq(700).sum()) == 0 and int(ref[1][12].eq(64).sum()) == 0 and int(flag.item()) == 0


def test_folded_pointnet_pieces_follow_the_view(A):
    """The bf16 weight pieces are registered buffers: a view built on the CPU and moved afterwards is split on its new
    device before the first forward pass, and gives the bits of a view built on the GPU."""
    from hit_adv_amd.model.pointnet import PointNetFeatureModel
    torch.manual_seed(3)
    m = PointNetFeatureModel(40, normal_channel=False).eval()
    late = m.attack_view().cuda()          # folded on the CPU, moved afterwards
    early = m.cuda().attack_view()
    assert late.s3_w3.is_cuda and late._split_on is None
    x = cu(clouds(2, 256, 77)[0].transpose(1, 2).contiguous())
    a, b = late(x)[0], early(x)[0]
    assert torch.equal(a, b) and torch.equal(late.e3_w3, early.e3_w3)
    assert 's3_w3' not in late.state_dict()


def test_linear_max_fwd_bf16x3_ties_keep_the_first_point(A):
    g = torch.Generator().manual_seed(3)
    B, Np, Cin, Cout = 2, 1024, 128, 256
    base = torch.randn(B, 8, Cin, generator=g)
    x = base.repeat_interleave(Np // 8, dim=1).reshape(B * Np, Cin)
    W3 = A.split_weights_bf16x3(cu(torch.randn(Cout, Cin, generator=g)))
    _, idx = A.linear_max_fwd_bf16x3(cu(x), W3, B, Np)
    assert (idx.cpu() % (Np // 8) == 0).all()


def test_linear_max_fwd_ties_and_errors(A):
    """Duplicate points give exactly equal activations: the lowest point index wins, across MFMA tiles,
    lane halves and point splits.  Unsupported shapes are refused, not silently mis-computed."""
    g = torch.Generator().manual_seed(3)
    B, Np, Cin, Cout = 2, 1024, 128, 256
    base = torch.randn(B, 8, Cin, generator=g)
    x = base.repeat_interleave(Np // 8, dim=1).reshape(B * Np, Cin)   # every point is one of 8 rows, runs of 128
    Wt = torch.randn(Cin, Cout, generator=g)
    val, idx = A.linear_max_fwd(cu(x), cu(Wt), B, Np)
    assert (idx.cpu() % (Np // 8) == 0).all()                            # first point of the winning run
    xs = x.view(B, Np, Cin).flip(1).reshape(B * Np, Cin).contiguous()
    _, idx_f = A.linear_max_fwd(cu(xs), cu(Wt), B, Np)
    assert (idx_f.cpu() % (Np // 8) == 0).all()
    with pytest.raises(A._lib.HitAdvLibraryError):
        A.linear_max_fwd(cu(torch.randn(64, 96)), cu(torch.randn(96, 64)), 1, 64)
    with pytest.raises(A._lib.HitAdvLibraryError):
        A.linear_max_fwd(cu(torch.randn(64, 128)), cu(torch.randn(128, 100)), 1, 64)


@pytest.mark.parametrize("B,K,NOUT", [(32, 1024, 512), (32, 512, 256), (32, 256, 9), (4, 256, 4096), (33, 256, 40),
                                      (5, 9, 256), (32, 4096, 256), (7, 40, 256), (1, 70, 33)])
def test_fc_layer_mfma(A, B, K, NOUT):
    g = torch.Generator().manual_seed(K + NOUT)
    x = torch.randn(B, K, generator=g)
    Wt = torch.randn(K, NOUT, generator=g) / K ** 0.5
    bias = torch.randn(NOUT, generator=g)
    mask = torch.randn(B, K, generator=g)
    ref = x.double() @ Wt.double()
    close(A.fc_layer(cu(x), cu(Wt)), ref.float(), rtol=1e-5, atol=1e-5)
    close(A.fc_layer(cu(x), cu(Wt), cu(bias), relu=True), (ref + bias.double()).clamp_min(0.).float(), rtol=1e-5, atol=1e-5)
    refm = (x.double() * (mask > 0)) @ Wt.double()
    got = A.fc_layer(cu(x), cu(Wt), mask=cu(mask))
    close(got, refm.float(), rtol=1e-5, atol=1e-5)
    assert torch.equal(got, A.fc_layer(cu(x), cu(Wt), mask=cu(mask)))
    part = torch.randn(B, 5, NOUT, generator=g)
    close(A.sum_partials(cu(part)), part.sum(1), rtol=1e-6, atol=1e-6)
    close(A.sum_partials(cu(part), cu(bias.expand(B, NOUT).contiguous())), part.sum(1) + bias, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("B,Np,mode", [(4, 1024, 'fp16x2'), (3, 1000, 'fp16x2'), (2, 77, 'fp16x2'), (33, 256, 'fp16x2'),
                                       (3, 1000, 'bf16x3'), (2, 77, 'f32'), (4, 1024, 'f32')])
def test_pointnet_engine_matches_module(B, Np, mode):
    """The HIP PointNet engine (rowmlp / linear_max / fc_layer kernels) against the nn.Module evaluated in float64
    on the CPU: logits, trans_feat and the input gradient -- also for a gradient arriving through trans_feat.  In each of
    the engine's matrix modes: 'fp16x2' (default: shared layers and the 128 -> 1024 layers on the fp16 cores, packed pieces
    in between), 'bf16x3' and 'f32' (shared layers as exact f32 MFMA chains)."""
    from hit_adv_amd.model.pointnet import PointNetFeatureModel
    torch.manual_seed(B + Np)
    m = PointNetFeatureModel(40, normal_channel=False).eval()
    with torch.no_grad():  # non-trivial BatchNorm statistics so that the folding is exercised
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.normal_(0, 0.1)
                mod.running_var.uniform_(0.5, 1.5)
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.normal_(0, 0.1)
    data, _ = synth_batch(B, Np, first=40)
    x = data[:, :, :3].transpose(1, 2).contiguous()
    g = torch.Generator().manual_seed(1)
    wl, wt = torch.randn(B, 40, generator=g), torch.randn(B, 64, 64, generator=g) * 0.05
    md = PointNetFeatureModel(40, normal_channel=False).double().eval()
    md.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
    xd = x.double().requires_grad_()
    ld, td = md(xd)
    gl, = torch.autograd.grad((ld * wl.double()).sum(), xd, retain_graph=True)
    gboth, = torch.autograd.grad((ld * wl.double()).sum() + (td * wt.double()).sum(), xd)
    view = m.cuda().attack_view()
    assert view.hip_engine
    view.matrix_mode = mode
    xc = cu(x).requires_grad_()
    lh, th = view(xc)
    view.check_range()
    close(lh, ld.float(), rtol=1e-4, atol=2e-5)
    close(th, td.float(), rtol=1e-4, atol=2e-5)
    gh, = torch.autograd.grad((lh * cu(wl)).sum(), xc, retain_graph=True)
    scale = float(gl.abs().max())
    close(gh, gl.float(), rtol=1e-3, atol=2e-5 * scale)
    gh2, = torch.autograd.grad((lh * cu(wl)).sum() + (th * cu(wt)).sum(), xc)
    close(gh2, gboth.float(), rtol=1e-3, atol=2e-5 * float(gboth.abs().max()))
    # bitwise reproducible, and the same function as the PyTorch-op formulation of the view
    l2, _ = view(xc)
    assert torch.equal(l2, lh)
    view.hip_engine = False
    xt = cu(x).requires_grad_()
    lt, tt = view(xt)
    gt, = torch.autograd.grad((lt * cu(wl)).sum(), xt)
    close(lh, lt, rtol=1e-4, atol=2e-5)
    close(gh, gt, rtol=1e-3, atol=2e-5 * scale)


@pytest.mark.parametrize("B,Np", [(3, 1000), (2, 64), (5, 130)])
def test_packed_pieces_equal_the_split_in_v1(A, B, Np):
    """rowmlp_fwd mode 2 hands its 128-wide activation on as packed words (fp16 hi | fp16 lo << 16): the words are the two
    pieces V1 would split the mode-1 activation into, the 64-wide activation is the same bits in both modes, and V1 on the
    packed words returns the bits of V1 on the floats."""
    from hit_adv_amd.model.pointnet import PointNetFeatureModel
    torch.manual_seed(Np)
    view = PointNetFeatureModel(40, normal_channel=False).cuda().eval().attack_view()
    data, _ = synth_batch(B, Np, first=3)
    x = data[:, :, :3].transpose(1, 2).contiguous().cuda()
    R = B * Np
    a1f, a2f, a1p, a2p = (torch.empty(R, c, device='cuda') for c in (64, 128, 64, 128))
    A.pointnet_rowmlp_fwd(0, B, Np, view.s2_w, view.s2_b, a2f, x=x, W0=view.s1_w, b0=view.s1_b, o0=a1f, mode=1,
                          range_flag=view.range_flag)
    A.pointnet_rowmlp_fwd(0, B, Np, view.s2_w, view.s2_b, a2p, x=x, W0=view.s1_w, b0=view.s1_b, o0=a1p, mode=2,
                          range_flag=view.range_flag)
    assert torch.equal(a1f, a1p)
    words = a2p.view(torch.int32)
    hi = (words & 0xffff).to(torch.int16).view(torch.float16)
    lo = (words >> 16).to(torch.int16).view(torch.float16)
    h1 = a2f.half()
    h2 = ((a2f - h1.float()) * 2048.0).half()
    assert torch.equal(hi, h1) and torch.equal(lo, h2)
    assert torch.equal(words != 0, a2f != 0)  # what the backward pass uses the packed activation for: the ReLU mask
    # mode 1 against the exact f32 chain of mode 0: fp32 roundoff of a 64-deep product
    a1e, a2e = torch.empty(R, 64, device='cuda'), torch.empty(R, 128, device='cuda')
    A.pointnet_rowmlp_fwd(0, B, Np, view.s2_w, view.s2_b, a2e, x=x, W0=view.s1_w, b0=view.s1_b, o0=a1e, mode=0)
    assert torch.equal(a1e, a1f)  # the 3 -> 64 layer is exact f32 on the VALU in every mode
    close(a2f, a2e, rtol=0, atol=2e-6 * float(a2e.abs().max()))
    g1, j1 = A.linear_max_fwd_f16x2(a2f, view.pieces('s3', 2), B, Np, bias=view.s3_b, relu=True, range_flag=view.range_flag)
    g2, j2 = A.linear_max_fwd_f16x2(a2p, view.pieces('s3', 2), B, Np, bias=view.s3_b, relu=True, range_flag=view.range_flag,
                                    packed=True)
    assert torch.equal(g1, g2) and torch.equal(j1, j2)
    view.check_range()


@pytest.mark.parametrize("Bn,M,N,K,ta,tb", [(3, 256, 256, 64, False, True), (2, 256, 256, 256, True, False), (5, 64, 128, 192, False, False),
                                           (1, 128, 64, 64, True, True), (32, 256, 64, 256, False, False)])
def test_bmm_matches_float64_and_autograd(A, Bn, M, N, K, ta, tb):
    """The tiled batched product (PCT's attention blocks) against float64, all four storage combinations, and its backward --
    four more calls of the same kernel -- against autograd through torch.bmm; bitwise reproducible."""
    g = torch.Generator().manual_seed(M + N + K + Bn)
    a = torch.randn(Bn, *((K, M) if ta else (M, K)), generator=g)
    b = torch.randn(Bn, *((N, K) if tb else (K, N)), generator=g)
    w = torch.randn(Bn, M, N, generator=g)
    ad, bd = a.double().requires_grad_(), b.double().requires_grad_()
    ref = torch.bmm(ad.transpose(1, 2) if ta else ad, bd.transpose(1, 2) if tb else bd)
    (ref * w.double()).sum().backward()
    ag, bg = cu(a).requires_grad_(), cu(b).requires_grad_()
    out = A.bmm(ag, bg, ta, tb)
    assert out.shape == (Bn, M, N)
    close(out, ref.detach().float(), rtol=0, atol=2e-6 * float(ref.detach().abs().max()))
    (out * cu(w)).sum().backward()
    close(ag.grad, ad.grad.float(), rtol=0, atol=2e-6 * float(ad.grad.abs().max()))
    close(bg.grad, bd.grad.float(), rtol=0, atol=2e-6 * float(bd.grad.abs().max()))
    assert torch.equal(out, A.bmm(cu(a), cu(b), ta, tb))
    # shapes the tiles do not cover go to torch.bmm
    x, y = cu(torch.randn(2, 10, 7, generator=g)), cu(torch.randn(2, 7, 5, generator=g))
    assert torch.equal(A.bmm(x, y), torch.bmm(x, y))


@pytest.mark.parametrize("M,K,N", [(1000, 64, 128), (4096, 512, 1024), (777, 1024, 512), (3, 32, 128), (70000, 128, 256)])
def test_gemm_f16x2_is_fp32_accurate(A, M, K, N):
    """The fp16x2 GEMM (two fp16 pieces per operand, three exact products) against float64: no further from it than torch's
    f32 GEMM, with bias / ReLU / a ReLU-mask gate on the input; pieces reconstruct the weights; bitwise reproducible."""
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    mask = torch.randn(M, K, generator=g)
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    Wp = A.split_rows_f16x2(cu(W), range_flag=flag)
    pieces = Wp.view(torch.float16).float().cpu()
    rec = pieces[0].double() + pieces[1].double() / 2048.0
    assert float((rec - W.double()).abs().max()) <= 2.0 ** -22 * float(W.abs().max())
    ref = x.double() @ W.double().t()
    scale = float(ref.abs().max())
    got = A.gemm_f16x2(cu(x), Wp, range_flag=flag)
    lib = (cu(x) @ cu(W).t()).cpu()
    err, err_lib = float((got.cpu().double() - ref).abs().max()), float((lib.double() - ref).abs().max())
    assert err <= max(err_lib, 2e-7 * scale), (err, err_lib)
    close(got, ref.float(), rtol=0, atol=5e-7 * scale)
    assert torch.equal(got, A.gemm_f16x2(cu(x), Wp, range_flag=flag))
    got = A.gemm_f16x2(cu(x), Wp, bias=cu(bias), relu=True, range_flag=flag)
    close(got, (ref + bias.double()).clamp_min(0.).float(), rtol=0, atol=5e-7 * scale)
    refm = (x.double() * (mask > 0)) @ W.double().t()
    close(A.gemm_f16x2(cu(x), Wp, mask=cu(mask), range_flag=flag), refm.float(), rtol=0, atol=5e-7 * scale)
    assert int(flag.item()) == 0
    x[0, 0] = 7e4  # beyond fp16
    A.gemm_f16x2(cu(x), Wp, range_flag=flag)
    assert int(flag.item()) == 1


@pytest.mark.parametrize("M,K,N", [(256, 64, 128), (700, 96, 128), (2048 + 37, 512, 256), (4096, 1024, 512), (300, 32, 128)])
def test_gemm_f16x2_ring_kernel_equals_the_staged_kernel(A, M, K, N):
    """gemm_f16x2_ring_k (operands global -> LDS by DMA into a three-stage ring, the fp32 A rows split by the wave that reads them)
    computes gemm_f16x2_k's products in its order: the same bits, for whole and partial row blocks, 1 to 32 K steps (K = 32 stays
    on the staged kernel), with the bias / ReLU epilogue; and the fused DGCNN layer and the range flag through the same switch."""
    from hit_adv_amd import _lib
    L = _lib.load()
    g = torch.Generator().manual_seed(M * 3 + K + N)
    x = torch.randn(M, K, generator=g) * torch.logspace(-3, 2, K)  # columns of very different magnitudes
    W = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    Wp = A.split_rows_f16x2(cu(W), range_flag=flag)
    outs = {}
    try:
        for ring in (0, 2):  # 2: the ring kernel whatever K (1, the default, keeps K < 256 on the staged kernel)
            L.hitadv_debug_g16_ring(ring)
            outs[ring] = (A.gemm_f16x2(cu(x), Wp, range_flag=flag), A.gemm_f16x2(cu(x), Wp, bias=cu(bias), relu=True, range_flag=flag))
        assert int(flag.item()) == 0
        assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])
        big = x.clone()
        big[M // 2, K - 1] = 7e4  # beyond fp16: both kernels must raise the flag
        for ring in (0, 2):
            L.hitadv_debug_g16_ring(ring)
            flag.zero_()
            A.gemm_f16x2(cu(big), Wp, range_flag=flag)
            assert int(flag.item()) == 1, ring
    finally:
        L.hitadv_debug_g16_ring(1)


@pytest.mark.parametrize("B,Np,Cin,C", [(3, 300, 64, 128), (2, 1024, 512, 1024)])
def test_linear_lrelu_pool_ring_kernel_equals_the_staged_kernel(A, B, Np, Cin, C):
    from hit_adv_amd import _lib
    L = _lib.load()
    g = torch.Generator().manual_seed(B + Np + C)
    x = torch.randn(B * Np, Cin, generator=g)
    W = torch.randn(C, Cin, generator=g) / Cin ** 0.5
    bias = torch.randn(C, generator=g) * 0.1
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    Wp, Wtp = A.split_rows_f16x2(cu(W), flag), A.split_rows_f16x2(cu(W.t().contiguous()), flag)
    res = {}
    try:
        for ring in (0, 2):
            L.hitadv_debug_g16_ring(ring)
            res[ring] = A.linear_lrelu_pool(cu(x), Wp, Wtp, cu(bias), B, Np, 0.2, flag, return_arg=True)
    finally:
        L.hitadv_debug_g16_ring(1)
    assert torch.equal(res[0][0], res[2][0]) and torch.equal(res[0][1], res[2][1])


@pytest.mark.parametrize("B,Np,Cin,C", [(2, 1024, 512, 1024), (3, 500, 512, 1024), (2, 77, 128, 128), (1, 256, 512, 256)])
def test_linear_lrelu_pool_forward_and_backward(A, B, Np, Cin, C):
    """DGCNN's embedding layer + LeakyReLU + max / mean pooling in one kernel against the float64 composition: values, the
    arg-max table (lowest point on ties), the sign bit mask through the gradient, bitwise reproducibility."""
    g = torch.Generator().manual_seed(B * 1000 + Np + C)
    x = torch.randn(B * Np, Cin, generator=g)
    x[5] = x[3]  # two points of cloud 0 with identical features: a tie wherever one of them is the maximum
    W = torch.randn(C, Cin, generator=g) / Cin ** 0.5
    bias = torch.randn(C, generator=g) * 0.1
    w = torch.randn(B, 2 * C, generator=g)
    xd = x.double().requires_grad_()
    z = (xd @ W.double().t() + bias.double()).view(B, Np, C)
    h = torch.nn.functional.leaky_relu(z, negative_slope=0.2)
    ref = torch.cat((h.max(dim=1)[0], h.mean(dim=1)), dim=1).detach()
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    Wp, Wtp = A.split_rows_f16x2(cu(W), flag), A.split_rows_f16x2(cu(W.t().contiguous()), flag)
    xc = cu(x).requires_grad_()
    out, arg = A.linear_lrelu_pool(xc, Wp, Wtp, cu(bias), B, Np, 0.2, flag, return_arg=True)
    scale = float(ref.abs().max())
    close(out, ref.float(), rtol=0, atol=1e-6 * scale)
    # the table: a maximiser of the float64 activation to within fp32 noise, and the LOWEST point among exact ties
    h = h.detach()
    hmax = h.max(dim=1)[0]
    at = h.gather(1, arg.cpu().long().unsqueeze(1)).squeeze(1)
    assert float((hmax - at).abs().max()) <= 2e-6 * scale
    assert not bool((arg[0].cpu() == 5).any())  # point 3 wins every tie with its copy
    assert int(flag.item()) == 0
    # gradient: route the max half through the kernel's own table (fp32 near-ties may pick another winner than float64)
    gd_max = torch.zeros(B, Np, C, dtype=torch.float64).scatter_(1, arg.cpu().long().unsqueeze(1), w[:, :C].double().unsqueeze(1))
    gd = (gd_max + w[:, C:].double().unsqueeze(1) / Np) * torch.where(z.detach() > 0, 1.0, 0.2)
    gref = gd.view(B * Np, C) @ W.double()
    out.backward(cu(w))
    # a sign bit may differ from float64's where |z| is at fp32 noise: there the slope is 1 instead of 0.2 or the reverse, which
    # moves the gradient of that point by at most 0.8 |g[p,c]| |W[c,:]|; everything else is held to fp32 roundoff
    amb = (z.detach().abs() < 1e-5 * float(z.detach().abs().max())).double().view(B * Np, C)
    slack = 0.8 * (amb * (gd_max + w[:, C:].double().unsqueeze(1) / Np).abs().view(B * Np, C)) @ W.double().abs()
    diff = (xc.grad.cpu().double() - gref).abs()
    assert float(amb.mean()) < 1e-3
    assert bool((diff <= slack + 2e-6 * float(gref.abs().max())).all()), float((diff - slack).max())
    x2 = cu(x).requires_grad_()
    out2, arg2 = A.linear_lrelu_pool(x2, Wp, Wtp, cu(bias), B, Np, 0.2, flag, return_arg=True)
    out2.backward(cu(w))
    assert torch.equal(out, out2) and torch.equal(arg, arg2) and torch.equal(xc.grad, x2.grad)


@pytest.mark.parametrize("B,Np,D,K", [(2, 1024, 64, 5), (3, 500, 64, 20), (2, 1024, 128, 5), (1, 130, 128, 20), (2, 33, 64, 8)])
def test_knn_features_mfma(A, B, Np, D, K):
    """Fused feature-space kNN (MFMA scores + per-lane selection) against the reference's expression evaluated in float64:
    identical neighbour sets wherever the K-th and (K+1)-th scores are further apart than fp32 noise, self at rank 0,
    descending scores, duplicates resolved towards the lower index."""
    g = torch.Generator().manual_seed(B * 100 + D + K)
    x = torch.randn(B, Np, D, generator=g)
    x[0, 7] = x[0, 3]  # exact duplicate: point 3 and 7 see identical scores everywhere
    xd = x.double()
    inner = -2 * torch.matmul(xd, xd.transpose(2, 1))
    xx = (xd ** 2).sum(2, keepdim=True)
    score = -xx - inner - xx.transpose(2, 1)
    top = score.topk(K + 1, dim=-1)
    idx = A.knn_features(cu(x), K).cpu()
    assert idx.dtype == torch.int64 and idx.shape == (B, Np, K)
    got = score.gather(2, idx)
    assert (got[..., :-1] >= got[..., 1:] - 1e-3).all()                       # closest first
    clear = (top.values[..., K - 1] - top.values[..., K]) > 1e-3               # unambiguous K-set
    assert clear.float().mean() > 0.95
    same = (idx.sort(dim=-1).values == top.indices[..., :K].sort(dim=-1).values).all(-1)
    assert same[clear].all()
    self_first = idx[..., 0] == torch.arange(Np)[None, :]
    self_first[0, 7] = self_first[0, 7] | (idx[0, 7, 0] == 3)                  # the duplicate pair: lower index first
    assert self_first.all() and idx[0, 7, 0] == 3 and idx[0, 3, 0] == 3
    assert torch.equal(idx, A.knn_features(cu(x), K).cpu())


@pytest.mark.parametrize("B,n,C", [(3, 1000, 128), (2, 17, 64), (4, 1024, 1024)])
def test_lrelu_pool_matches_torch(A, B, n, C):
    """DGCNN's activation + max/mean pooling in one pass: max bit-exact, mean to summation order, arg = first maximiser,
    gradient against autograd of the torch composition, bitwise reproducible."""
    g = torch.Generator().manual_seed(C + n)
    z = torch.randn(B, n, C, generator=g)
    z[:, 3, :5] = z[:, 9, :5] = 50.0  # a tie for the maximum: the earlier point takes the gradient
    za, zb = z.cuda().requires_grad_(), z.cuda().requires_grad_()
    w = torch.randn(B, 2 * C, generator=g).cuda()
    out = A.lrelu_pool(za, 0.2)
    h = torch.nn.functional.leaky_relu(zb, negative_slope=0.2)
    ref = torch.cat((h.max(dim=1)[0], h.mean(dim=1)), dim=1)
    assert torch.equal(out[:, :C], ref[:, :C])
    close(out[:, C:], ref[:, C:], rtol=1e-5, atol=1e-6)
    ga, = torch.autograd.grad((out * w).sum(), za)
    # reference gradient with the tie routed to the first maximiser (torch's max picks an unspecified one)
    first = (h == h.max(dim=1, keepdim=True)[0]).float().argmax(dim=1)  # [B,C]
    onehot = torch.zeros_like(h).scatter_(1, first.unsqueeze(1), 1.0)
    slope = torch.where(zb > 0, 1.0, 0.2)
    gb = (onehot * w[:, None, :C] + w[:, None, C:] / n) * slope
    close(ga, gb, rtol=1e-6, atol=1e-7)
    assert float(ga[:, 9, :5].abs().max()) <= float((w[:, C:C + 5].abs() / n).max()) + 1e-7  # tie: no max-gradient at point 9
    out2 = A.lrelu_pool(za, 0.2)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("B,n,S,ns,C", [(2, 300, 70, 32, 128), (3, 1024, 128, 64, 128), (1, 64, 64, 5, 256), (2, 130, 33, 7, 8)])
def test_group_add_relu_matches_torch(A, B, n, S, ns, C):
    """The split first layer of a sample-and-group block: forward bitwise against relu(U[idx] + V), gradients bitwise
    against the sums taken in ascending (list, slot) order, lists with repeated entries (ball-query padding) included,
    reproducible."""
    g = torch.Generator().manual_seed(n + S + ns)
    U = torch.randn(B, n, C, generator=g)
    V = torch.randn(B, S, C, generator=g)
    idx = torch.randint(0, n, (B, S, ns), generator=g)
    idx[:, :, ns // 2:] = idx[:, :, :1]  # padded lists: the first entry repeated
    idx[idx == 3] = 4                    # point 3 is in no list
    w = torch.randn(B, S, ns, C, generator=g)
    Ua, Va = U.cuda().requires_grad_(), V.cuda().requires_grad_()
    H = A.group_add_relu(Ua, Va, idx.cuda())
    ref = torch.relu(U.gather(1, idx.reshape(B, S * ns, 1).expand(B, S * ns, C)).view(B, S, ns, C) + V[:, :, None, :])
    assert torch.equal(H.detach().cpu(), ref)
    gu, gv = torch.autograd.grad((H * w.cuda()).sum(), [Ua, Va])
    gm = w * (ref > 0)
    dv = torch.zeros(B, S, C)
    for t in range(ns):
        dv += gm[:, :, t, :]
    assert torch.equal(gv.cpu(), dv)
    du = torch.zeros(B, n, C)
    for i in range(S):
        for t in range(ns):
            du.scatter_add_(1, idx[:, i, t].view(B, 1, 1).expand(B, 1, C), gm[:, i, t, :].unsqueeze(1))
    assert torch.equal(gu.cpu(), du)
    assert float(gu[:, 3].abs().max()) == 0.0
    gu2, = torch.autograd.grad((A.group_add_relu(Ua, Va, idx.cuda()) * w.cuda()).sum(), [Ua])
    assert torch.equal(gu, gu2)


def test_pointnet_engine_captures_on_a_fresh_stream():
    """The standard torch.cuda.graph pattern (capture on the graph's own side stream, no eager pass on THAT stream first)
    works: the engine's ticket scratch is created inside the capture, and replays reproduce the eager result bit for bit."""
    from hit_adv_amd.model.pointnet import PointNetFeatureModel
    torch.manual_seed(2)
    view = PointNetFeatureModel(40, normal_channel=False).cuda().eval().attack_view()
    data, _ = synth_batch(4, 512, first=77)
    x = data[:, :, :3].transpose(1, 2).contiguous().cuda().requires_grad_()
    w = torch.randn(4, 40, device='cuda')

    def step():
        logits = view(x)[0]
        g, = torch.autograd.grad((logits * w).sum(), x)
        return logits, g
    l_ref, g_ref = (t.detach().clone() for t in step())  # keep no autograd node of the eager pass alive: its
    torch.cuda.synchronize()                             # AccumulateGrad is bound to the default stream
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        l_out, g_out = step()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(l_out, l_ref) and torch.equal(g_out, g_ref)


def test_fused_regulariser_matches_torch_composition(A):
    """hitadv_regulariser_{fwd,bwd} vs the reference's composition of ChamferDist (on [B,3,N], quirk Q1),
    transformation_loss, curv_std_loss and the scale_const weighting, evaluated by the CPU oracle + autograd."""
    g = torch.Generator().manual_seed(11)
    B, Np, C = 5, 700, 48
    ori = torch.randn(B, 3, Np, generator=g) * 0.4
    adv = (ori + 0.05 * torch.randn(B, 3, Np, generator=g)).requires_grad_()
    adv.data[1, 0] = ori[1, 1] + 0.01  # make a cross-row match the nearest one for one cloud
    P = (torch.rand(B, C, 3, generator=g) - 0.5).requires_grad_()
    sig = (0.1 + 1.1 * torch.rand(B, C, generator=g)).requires_grad_()
    kap = torch.rand(B, C, 1, generator=g)
    scale = torch.tensor([10., 45., 80., 5., 27.5])
    cd, ker, hide, lo, hi = 1e-4, 1.0, 1.0, 0.1, 1.2
    dist = (O.chamfer_dist(adv, ori, torch.full((B,), cd, dtype=torch.float64)) + O.transformation_loss(P, sig, C) * ker
            + (O.curv_std_loss(sig, kap, hi, lo) * hide).mean())
    ref = (scale * dist).mean()
    ref.backward()
    ref_hide = ((kap - kap.min()) / (kap.max() - kap.min() + 1e-7)).squeeze(-1)
    Pg, sg, ag = (cu(t.detach()).requires_grad_() for t in (P, sig, adv))
    dist_out = torch.zeros((), device='cuda')
    out = A.regulariser(Pg, sg, ag, cu(ori), cu(ref_hide), cu(scale), (cd, ker, hide), (lo, hi), dist_out)
    close(out, ref, rtol=1e-5)
    close(dist_out, dist, rtol=1e-5)
    out.backward()
    close(Pg.grad, P.grad, rtol=1e-4, atol=1e-7)
    close(sg.grad, sig.grad, rtol=1e-4, atol=1e-7)
    close(ag.grad, adv.grad, rtol=1e-4, atol=1e-9)
    # single terms
    for w in ((cd, 0., 0.), (0., ker, 0.), (0., 0., hide)):
        d2 = torch.zeros((), device='cuda')
        A.regulariser(Pg, sg, ag, cu(ori), cu(ref_hide), cu(scale), w, (lo, hi), d2)
        parts = (O.chamfer_dist(adv, ori, torch.full((B,), 1.0, dtype=torch.float64)), O.transformation_loss(P, sig, C),
                 O.curv_std_loss(sig, kap, hi, lo).mean())
        close(d2, sum(wi * pi for wi, pi in zip(w, parts)), rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("K,largest", [(1, True), (5, True), (20, True), (20, False), (64, False)])
def test_topk_rows_bit_exact(A, K, largest):
    g = torch.Generator().manual_seed(9)
    P = torch.randn(3, 130, 257, generator=g)
    P[0, 0, 5] = P[0, 0, 9] = 7.0  # ties -> lower column first
    P[0, 0, 100] = P[0, 0, 3] = -7.0
    vals, idx = A.topk_rows(cu(P), K, largest=largest)
    order = torch.sort(-P if largest else P, dim=-1, stable=True)
    assert torch.equal(idx.cpu(), order.indices[..., :K])
    assert torch.equal(vals.cpu(), P.gather(-1, order.indices[..., :K]))


@pytest.mark.parametrize("B,Np,C", [(32, 1024, 192), (3, 130, 7), (2, 300, 256)])
def test_engine_first_kernel_with_the_deformation_inside(A, B, Np, C):
    """hitadv_pointnet_rowmlp_fwd_deform == hitadv_deform_fwd followed by stage 0 of the forward chain: the deformed cloud,
    1 / sum k and both activation tensors, bit for bit (ragged last tile included)."""
    g = torch.Generator().manual_seed(B + Np + C)
    ori = cu(torch.randn(B, 3, Np, generator=g) * 0.4)
    central = ori[:, :, :C].contiguous()
    P, S = cu((torch.rand(B, C, 3, generator=g) - 0.5) * 0.5), cu(0.1 + 1.1 * torch.rand(B, C, generator=g))
    W0, b0, W2, b2 = cu(torch.randn(3, 64, generator=g)), cu(torch.randn(64, generator=g)), cu(torch.randn(64, 128, generator=g) * 0.2), cu(torch.randn(128, generator=g))
    adv_a, inv_a = torch.empty_like(ori), torch.empty(B, Np, device='cuda')
    A.deform_fwd_into(ori, central, P, S, adv_a, inv_a)
    o0a, o2a = torch.empty(B * Np, 64, device='cuda'), torch.empty(B * Np, 128, device='cuda')
    A.pointnet_rowmlp_fwd(0, B, Np, W2, b2, o2a, x=adv_a, W0=W0, b0=b0, o0=o0a)
    adv_b, inv_b = torch.zeros_like(ori), torch.zeros(B, Np, device='cuda')
    o0b, o2b = torch.empty_like(o0a), torch.empty_like(o2a)
    A.pointnet_rowmlp_fwd_deform(B, Np, ori, central, P, S, adv_b, inv_b, W0, b0, W2, b2, o0b, o2b)
    assert torch.equal(adv_a, adv_b) and torch.equal(inv_a, inv_b)
    assert torch.equal(o0a, o0b) and torch.equal(o2a, o2b)


def test_iteration_head_evaluates_the_classifiers_last_layer(A):
    """hitadv_iteration_head_reg with (features, last layer) instead of logits: the logits it writes are the layer's output
    (float64 reference), and every other output equals, bit for bit, a call that is handed those logits."""
    g = torch.Generator().manual_seed(77)
    B, K, F_, Np, C, dev = 32, 40, 256, 256, 24, 'cuda'

    def state():
        return dict(bestdist=torch.full((B,), 1e10, device=dev), bestscore=torch.full((B,), -1, dtype=torch.int64, device=dev),
                    o_bestdist=torch.full((B,), 1e10, device=dev), o_bestscore=torch.full((B,), -1, dtype=torch.int64, device=dev),
                    o_bestattack=torch.zeros(B, 3, Np, device=dev), pred=torch.zeros(B, dtype=torch.int64, device=dev),
                    dist_val=torch.zeros(B, device=dev))
    feat = torch.randn(B, F_, generator=g).relu()
    W = torch.randn(F_, K, generator=g) * 0.2
    bias = torch.randn(K, generator=g)
    label = torch.randint(0, K, (B,), generator=g).cuda()
    ori = cu(torch.randn(B, 3, Np, generator=g) * 0.4)
    adv = cu(torch.randn(B, 3, Np, generator=g) * 0.4)
    P, S = cu((torch.rand(B, C, 3, generator=g) - 0.5) * 0.5), cu(0.1 + 1.1 * torch.rand(B, C, generator=g))
    hide_ref, scale = cu(torch.rand(B, C, generator=g)), cu(10. + 70. * torch.rand(B, generator=g))
    regs, rng = (1e-4, 1.0, 1.0), (0.1, 1.2)
    for kind in (A.ADV_UNTARGETED, A.ADV_TARGETED, A.ADV_CROSS_ENTROPY):
        out = []
        logits = torch.empty(B, K, device=dev)
        for head in ((cu(feat), cu(W), cu(bias)), None):
            st, cnt = state(), torch.zeros(1, dtype=torch.int32, device=dev)
            d, loss, dl, sl = torch.empty(B, K, device=dev), torch.zeros((), device=dev), torch.zeros((), device=dev), torch.zeros((), device=dev)
            A.iteration_head_reg(logits, label, P, S, adv, st, cnt, kind, 30., loss, d, A.iteration_head_scratch(B, dev), ori, hide_ref,
                                 scale, regs, rng, torch.zeros(A.regulariser_scratch(B), device=dev), dl, sl, head=head)
            out.append((d, loss, dl, sl, st))
        close(logits, (feat.double() @ W.double() + bias.double()).float(), rtol=2e-6, atol=2e-6, what='logits evaluated in the head kernel')
        (d0, l0, a0, b0, s0), (d1, l1, a1, b1, s1) = out
        assert torch.equal(d0, d1) and torch.equal(l0, l1) and torch.equal(a0, a1) and torch.equal(b0, b1)
        for k in s0:
            assert torch.equal(s0[k], s1[k]), k


# ------------------------------------------------------------------ merged launches of the iteration (csrc/iteration.hip)
@pytest.mark.parametrize("B,K,Np,C", [(32, 40, 1024, 192), (5, 16, 130, 12), (1, 40, 64, 3)])
def test_merged_iteration_launches_equal_the_launches_they_merge(A, B, K, Np, C):
    """iteration_head == best_update + adv_loss, regulariser_fwd_fused == regulariser_fwd, deform_bwd_partials +
    adam_step_partials == deform_bwd + adam_step_sum: same bits, over several calls on the same state (the tickets the
    merged forms keep in their scratch must come back to zero every time)."""
    g = torch.Generator().manual_seed(B * 7 + C)
    dev = 'cuda'

    def state():
        return dict(bestdist=torch.full((B,), 1e10, device=dev), bestscore=torch.full((B,), -1, dtype=torch.int64, device=dev),
                    o_bestdist=torch.full((B,), 1e10, device=dev), o_bestscore=torch.full((B,), -1, dtype=torch.int64, device=dev),
                    o_bestattack=torch.zeros(B, 3, Np, device=dev), pred=torch.zeros(B, dtype=torch.int64, device=dev),
                    dist_val=torch.zeros(B, device=dev))
    label = torch.randint(0, K, (B,), generator=g).cuda()
    ori = cu(torch.randn(B, 3, Np, generator=g) * 0.4)
    central = ori[:, :, :C].contiguous()
    hide_ref = cu(torch.rand(B, C, generator=g))
    scale = cu(10. + 70. * torch.rand(B, generator=g))
    regs, rng = (1e-4, 1.0, 1.0), (0.1, 1.2)
    for kind in (A.ADV_UNTARGETED, A.ADV_TARGETED, A.ADV_CROSS_ENTROPY):
        sa, sb, sc = state(), state(), state()
        ca, cb = torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
        cc = torch.zeros(1, dtype=torch.int32, device=dev)
        head, head_c = A.iteration_head_scratch(B, dev), A.iteration_head_scratch(B, dev)
        reg_c = torch.zeros(A.regulariser_scratch(B), device=dev)
        reg_a = torch.zeros(A.regulariser_scratch(B), device=dev)
        reg_b = torch.zeros(A.regulariser_scratch(B), device=dev)
        Pa = cu((torch.rand(B, C, 3, generator=g) - 0.5) * 0.5)
        Sa = cu(0.1 + 1.1 * torch.rand(B, C, generator=g))
        Pb, Sb = Pa.clone(), Sa.clone()
        Pc, Sc = Pa.clone(), Sa.clone()
        Pd, Sd = Pa.clone(), Sa.clone()
        md = [torch.zeros_like(Pa), torch.zeros_like(Pa), torch.zeros_like(Sa), torch.zeros_like(Sa)]
        part_d = torch.empty(A.deform_bwd_scratch(B, Np, C), device=dev)
        tickets_d = torch.zeros(B, dtype=torch.int32, device=dev)
        ma = [torch.zeros_like(Pa), torch.zeros_like(Pa), torch.zeros_like(Sa), torch.zeros_like(Sa)]
        mb = [t.clone() for t in ma]
        mc = [t.clone() for t in ma]
        part_a = torch.empty(A.deform_bwd_scratch(B, Np, C), device=dev)
        part_b = torch.empty_like(part_a)
        part_c = torch.empty_like(part_a)
        for it in range(4):
            logits = cu(torch.randn(B, K, generator=g) * 3)
            adv, inv = torch.empty_like(ori), torch.empty(B, Np, device=dev)
            A.deform_fwd_into(ori, central, Pa, Sa, adv, inv)
            # reference: the separate launches
            A.best_update(logits, label, Pa, Sa, adv, sa, counter=ca)
            loss_a, d_a = A.adv_loss(kind, logits, label, 30.)
            la, lb = torch.zeros((), device=dev), torch.zeros((), device=dev)
            d_b, da_l, db_l = torch.empty_like(logits), torch.zeros((), device=dev), torch.zeros((), device=dev)
            A.iteration_head(logits, label, Pb, Sb, adv, sb, cb, kind, 30., lb, d_b, head)
            assert torch.equal(d_a, d_b) and torch.equal(loss_a, lb) and int(ca) == int(cb) == it + 1
            for k in sa:
                assert torch.equal(sa[k], sb[k]), k
            A.regulariser_fwd_into(Pa, Sa, adv, ori, hide_ref, scale, regs, rng, reg_a, da_l, la)
            A.regulariser_fwd_fused_into(Pb, Sb, adv, ori, hide_ref, scale, regs, rng, reg_b, db_l, lb)
            assert torch.equal(da_l, db_l) and torch.equal(la, lb)
            assert torch.equal(reg_a[:-1], reg_b[:-1]) and float(reg_b[-1]) == 0.
            # iteration_head + the regularisers' forward pass in ONE launch (a third copy of the state)
            d_c, lc, dc_l, sc_l = torch.empty_like(logits), torch.zeros((), device=dev), torch.zeros((), device=dev), torch.zeros((), device=dev)
            A.iteration_head_reg(logits, label, Pb, Sb, adv, sc, cc, kind, 30., lc, d_c, head_c, ori, hide_ref, scale, regs, rng,
                                 reg_c, dc_l, sc_l)
            assert torch.equal(d_c, d_b) and torch.equal(lc, loss_a) and int(cc) == it + 1
            for k in sa:
                assert torch.equal(sa[k], sc[k]), k
            assert torch.equal(dc_l, da_l) and torch.equal(sc_l, la)
            assert torch.equal(reg_c[:-1], reg_a[:-1]) and float(reg_c[-1]) == 0.
            up = cu(torch.randn(B, 3, Np, generator=g))
            gp, gs = torch.empty_like(Pa), torch.empty_like(Sa)
            gp2, gs2, ga = torch.empty_like(Pa), torch.empty_like(Sa), torch.empty_like(adv)
            A.regulariser_bwd_add(Pa, Sa, adv, ori, hide_ref, reg_a, up, regs, rng, gp2, gs2, ga)
            A.deform_bwd_into(ori, central, Pa, Sa, adv, inv, ga, part_a, gp, gs)
            A.adam_step_sum(Pa, Sa, gp, gp2, gs, gs2, *ma, ca, 0.05, 0.03, (-0.55, 0.55), rng)
            A.deform_bwd_partials_into(ori, central, Pb, Sb, adv, inv, ga, part_b)
            A.adam_step_partials(Pb, Sb, part_b, Np, gp2, gs2, *mb, cb, 0.05, 0.03, (-0.55, 0.55), rng)
            # ... and the _reg forms, which evaluate regulariser_bwd_add's three outputs themselves (state copy c)
            A.deform_bwd_partials_reg_into(ori, central, Pc, Sc, adv, inv, up, reg_a, regs, part_c)
            A.adam_step_partials_reg(Pc, Sc, part_c, Np, hide_ref, reg_a, regs, rng, *mc, cb, 0.05, 0.03, (-0.55, 0.55), rng)
            assert torch.equal(part_c, part_b)
            # ... and both of them as ONE launch (the Adam step as the tail of the deformation's backward; state copy d)
            A.deform_bwd_adam_reg(ori, central, Pd, Sd, adv, inv, up, hide_ref, reg_a, regs, rng, *md, cb, 0.05, 0.03, (-0.55, 0.55), rng,
                                  part_d, tickets_d)
            assert torch.equal(part_d, part_b) and torch.equal(Pd, Pc) and torch.equal(Sd, Sc) and int(tickets_d.abs().sum()) == 0
            for x, y in zip(mc, md):
                assert torch.equal(x, y)
            assert torch.equal(Pa, Pb) and torch.equal(Sa, Sb) and torch.equal(Pa, Pc) and torch.equal(Sa, Sc)
            for x, y, z in zip(ma, mb, mc):
                assert torch.equal(x, y) and torch.equal(x, z)


@pytest.mark.parametrize("pattern", ["crowded", "random", "one_crowded_tile"])
def test_rowmlp_bwd_two_word_tiles_and_their_overflow_path(A, pattern):
    """The fp16 backward chain covers two 64-point words per block and leaves a tile with more than 64 winning points to a
    second launch that takes it word by word.  Against the f32 chain (one word per block, no such path) on the same
    synthetic arg-max tables: 'crowded' sends the 1024 channels to 128 distinct points of the first tile of every cloud (both
    words full: the second launch does all the work), 'random' is the usual sparse case, 'one_crowded_tile' mixes the two;
    N = 1000 leaves a ragged last tile (a full word and 40 points).  All three stages, outputs, per-tile partials (summed over
    the tiles: the two forms cut the cloud differently) and the row-presence tables (the same bits in both layouts)."""
    torch.manual_seed(5)
    B, N, Cout = 3, 1000, 1024
    R = B * N
    dev = 'cuda'
    g = torch.Generator().manual_seed(7)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    if pattern == "crowded":
        idx = (torch.arange(Cout) % 128).expand(B, Cout).contiguous()
    elif pattern == "random":
        idx = torch.randint(0, N, (B, Cout), generator=g)
    else:
        idx = torch.randint(0, N, (B, Cout), generator=g)
        idx[:, :700] = 384 + (torch.arange(700) % 100)  # 100 distinct points of the 128-point tile [384, 512)
    idx = idx.to(dev)
    dg, gmask = rnd(B, Cout), rnd(B, Cout)
    W3r, W2r, W1r, W0r = rnd(Cout, 128) * 0.1, rnd(128, 64) * 0.1, rnd(64, 64) * 0.1, rnd(64, 3) * 0.1
    A2, A1, H1 = rnd(R, 128), rnd(R, 64), rnd(R, 64)
    T64, T3, x, dPin = rnd(B, 64, 64) * 0.1, rnd(B, 9), rnd(B, 3, N), rnd(B, 3, N)

    def run(mode):
        tiles, words = A.pointnet_rowmlp_bwd_tiles(B, N, mode, words=2 if mode else 1)
        pres2 = torch.zeros(B, tiles, words, device=dev, dtype=torch.int64)
        pres1 = torch.zeros(B, tiles, words, device=dev, dtype=torch.int64)
        dT64, dT3 = torch.full((B, tiles, 4096), float('nan'), device=dev), torch.full((B, tiles, 9), float('nan'), device=dev)
        dH1 = torch.zeros(R, 64, device=dev)
        dPts, dX = torch.full((B, 3, N), float('nan'), device=dev), torch.full((B, 3, N), float('nan'), device=dev)
        A.pointnet_rowmlp_bwd(2, B, N, dg, idx, W3r, A2, W2r, dH1, H1=H1, T=T64, dTpart=dT64, pres_out=pres2, mode=mode, words=words)
        A.pointnet_rowmlp_bwd(1, B, N, dg, idx, W3r, A2, W2r, dPts, gmask=gmask, A1=A1, W1r=W1r, H1=H1, dH1in=dH1, W0r=W0r,
                              T=T3, x=x, dTpart=dT3, pres_in=pres2, pres_out=pres1, mode=mode, words=words)
        A.pointnet_rowmlp_bwd(0, B, N, dg, idx, W3r, A2, W2r, dX, gmask=gmask, A1=A1, W0r=W0r, dPin=dPts, pres_in=pres1, mode=mode, words=words)
        torch.cuda.synchronize()
        return dict(dH1=dH1, dT64=dT64.sum(1), dT3=dT3.sum(1), dPts=dPts, dX=dX, pres2=pres2.reshape(B, -1), pres1=pres1.reshape(B, -1),
                    words=words)
    f32, f16 = run(0), run(1)
    assert f32['words'] == 1 and f16['words'] == 2
    assert torch.equal(f32['pres2'], f16['pres2']) and torch.equal(f32['pres1'], f16['pres1'])
    if pattern == "crowded":
        assert int(f16['pres2'][0, 0]) == -1 and int(f16['pres2'][0, 1]) == -1  # both words of the first tile full
    for name in ('dH1', 'dT64', 'dT3', 'dPts', 'dX'):
        a, b = f16[name], f32[name]
        assert torch.isfinite(a).all() and torch.isfinite(b).all(), name
        close(a, b, rtol=1e-4, atol=2e-6 * float(b.abs().max()), what='rowmlp_bwd two-word tiles vs one-word f32 chain: %s (%s)' % (name, pattern))
    again = run(1)
    assert all(torch.equal(again[k], f16[k]) for k in ('dH1', 'dT64', 'dT3', 'dPts', 'dX'))  # bitwise reproducible


@pytest.mark.parametrize("B,N", [(32, 256), (3, 64), (2, 1024), (5, 192)])
def test_offset_attention_norm_matches_torch_composition(A, B, N):
    """PCT's softmax + column renormalisation (model/pct_cls.py:127-131) as two launches each way against torch's own ops in
    float64: values, the gradient of a random linear functional, bitwise reproducibility; widths outside the kernel's take
    torch's ops."""
    g = torch.Generator().manual_seed(N)
    E = (torch.randn(B, N, N, generator=g) * 3).cuda().requires_grad_()
    w = torch.randn(B, N, N, generator=g).cuda()
    out = A.offset_attention_norm(E)
    gE, = torch.autograd.grad((out * w).sum(), E)
    Ed = E.detach().double().requires_grad_()
    S = torch.softmax(Ed, dim=-1)
    ref = S / (1e-9 + S.sum(dim=1, keepdim=True))
    gR, = torch.autograd.grad((ref * w.double()).sum(), Ed)
    close(out, ref.float(), rtol=1e-5, atol=1e-7, what='offset attention: normalised attention vs float64')
    close(gE, gR.float(), rtol=1e-4, atol=2e-6 * float(gR.abs().max()), what='offset attention: dE vs float64 autograd')
    close(out.sum(dim=1), (ref.sum(dim=1)).float(), rtol=1e-5, atol=1e-6)  # columns sum to one (up to the 1e-9)
    out2 = A.offset_attention_norm(E)
    g2, = torch.autograd.grad((out2 * w).sum(), E)
    assert torch.equal(out2, out) and torch.equal(g2, gE)
    odd = torch.randn(2, 100, 100, device='cuda')
    S2 = torch.softmax(odd, dim=-1)
    assert torch.equal(A.offset_attention_norm(odd), S2 / (1e-9 + S2.sum(dim=1, keepdim=True)))


@pytest.mark.parametrize("B,N", [(32, 256), (2, 64)])
def test_pct_offset_attention_layer_as_one_autograd_node(B, N):
    """Round 5: a whole PCT offset-attention layer (model/pct_cls.py:111-139) as ONE autograd node with a hand-written backward pass
    (ops.OffsetAttentionLayer) against the op-by-op composition it replaces (the same forward kernels: equal bits;
    gradient: the same products, accumulated in another order) and against float64 autograd of the reference's formula."""
    from hit_adv_amd.model.pct import SA_Layer
    torch.manual_seed(N)
    layer = SA_Layer(256).eval().cuda()
    with torch.no_grad():
        layer.after_norm.running_mean.normal_(0., 0.1)
        layer.after_norm.running_var.uniform_(0.5, 1.5)
    x = torch.randn(B, N, 256, device='cuda').requires_grad_()
    w = torch.randn(B, N, 256, device='cuda')
    assert layer._fused_layer_ok(x)
    out = layer.forward_pm(x)
    g, = torch.autograd.grad((out * w).sum(), x)
    SA_Layer.FUSED_BACKWARD = False
    try:
        out0 = layer.forward_pm(x)
        g0, = torch.autograd.grad((out0 * w).sum(), x)
    finally:
        SA_Layer.FUSED_BACKWARD = True
    assert torch.equal(out, out0)
    close(g, g0, rtol=0, atol=2e-6 * float(g0.abs().max()), what='offset-attention layer: dx, one node vs the op-by-op composition')
    ld = copy.deepcopy(layer).double()
    xd = x.detach().double().transpose(1, 2).contiguous().requires_grad_()  # the reference's layout [B,C,N]
    od = ld(xd)
    gd, = torch.autograd.grad((od * w.double().transpose(1, 2)).sum(), xd)
    so, sg = float(od.detach().abs().max()), float(gd.abs().max())
    close(out / so, (od.detach().transpose(1, 2).float()) / so, rtol=0, atol=2e-6, what='offset-attention layer vs float64 (over the output scale)')
    close(g / sg, gd.transpose(1, 2).float() / sg, rtol=0, atol=5e-6, what='offset-attention layer: dx vs float64 autograd (over its scale)')
    g2, = torch.autograd.grad((layer.forward_pm(x) * w).sum(), x)
    assert torch.equal(g2, g)


@pytest.mark.parametrize("B,Np", [(3, 1000), (2, 64), (5, 130), (64, 1024), (9, 2048)])
@pytest.mark.parametrize("mode", [1, 2])
def test_rowmlp_stream_equals_the_tile_kernel(A, B, Np, mode):
    """Round 5: the streaming forward kernel of the PointNet engine's shared layers (rowmlp_stream_k: a workgroup takes a
    run of tiles, weights split once, transposed products, results row-wise through LDS) writes the bits of the round-3
    kernel (one 64-point tile per workgroup) -- all three stages, the deformation and the input transform evaluated inside,
    packed and unpacked 128-wide activation, ragged last tiles, runs of tiles that cross clouds (B = 64 x 16 tiles = 2 per
    workgroup ... B = 9 x 32 tiles) -- and raises the range flag on the same inputs."""
    g = torch.Generator().manual_seed(B * 7 + Np + mode)
    C, R = 48, B * Np
    ori = cu(torch.randn(B, 3, Np, generator=g) * 0.4)
    central = ori[:, :, :C].contiguous()
    P, S = cu((torch.rand(B, C, 3, generator=g) - 0.5) * 0.5), cu(0.1 + 1.1 * torch.rand(B, C, generator=g))
    W0, b0 = cu(torch.randn(3, 64, generator=g)), cu(torch.randn(64, generator=g))
    W1, b1 = cu(torch.randn(64, 64, generator=g) * 0.2), cu(torch.randn(64, generator=g))
    W2, b2 = cu(torch.randn(64, 128, generator=g) * 0.2), cu(torch.randn(128, generator=g))
    T3 = cu(torch.eye(3).repeat(B, 1, 1) + 0.1 * torch.randn(B, 3, 3, generator=g)).reshape(B, 9).contiguous()
    T64 = cu(torch.eye(64).repeat(B, 1, 1) + 0.05 * torch.randn(B, 64, 64, generator=g)).contiguous()
    F5, W6, b6 = cu(torch.randn(B, 256, generator=g).relu()), cu(torch.randn(256, 9, generator=g) * 0.05), cu(torch.randn(9, generator=g))
    hin = cu(torch.randn(R, 64, generator=g).relu())

    def run(form):
        before = A.pointnet_rowmlp_form(form)
        try:
            out = {}
            flag = torch.zeros(1, dtype=torch.int32, device='cuda')
            new = lambda *shape: torch.full(shape, float('nan'), device='cuda')  # noqa: E731
            # stage 0 on a given cloud, and with the deformation inside
            o0, o2 = new(R, 64), new(R, 128)
            A.pointnet_rowmlp_fwd(0, B, Np, W2, b2, o2, x=ori, W0=W0, b0=b0, o0=o0, mode=mode, range_flag=flag)
            out['s0'] = (o0, o2)
            adv, inv, o0, o2 = new(B, 3, Np), new(B, Np), new(R, 64), new(R, 128)
            A.pointnet_rowmlp_fwd_deform(B, Np, ori, central, P, S, adv, inv, W0, b0, W2, b2, o0, o2, mode=mode, range_flag=flag)
            out['s0d'] = (adv, inv, o0, o2)
            # stage 1 with a given transform, and with STN3d's last layer inside
            xp, o0, o1, o2 = new(R, 3), new(R, 64), new(R, 64), new(R, 128)
            A.pointnet_rowmlp_fwd(1, B, Np, W2, b2, o2, x=ori, T=T3, W0=W0, b0=b0, W1=W1, b1=b1, xp=xp, o0=o0, o1=o1, mode=mode,
                                  range_flag=flag)
            out['s1'] = (xp, o0, o1, o2)
            Tout, xp, o0, o1, o2 = new(B, 9), new(R, 3), new(R, 64), new(R, 64), new(R, 128)
            A.pointnet_rowmlp_fwd_stn(B, Np, ori, F5, W6, b6, Tout, W0, b0, W1, b1, W2, b2, o0, o1, o2, xp=xp, mode=mode, range_flag=flag)
            out['s1t'] = (Tout, xp, o0, o1, o2)
            # stage 2, with and without its 64-wide output
            o0, o2 = new(R, 64), new(R, 128)
            A.pointnet_rowmlp_fwd(2, B, Np, W2, b2, o2, T=T64, hin=hin, o0=o0, mode=mode, range_flag=flag)
            out['s2'] = (o0, o2)
            o2 = new(R, 128)
            A.pointnet_rowmlp_fwd(2, B, Np, W2, b2, o2, T=T64, hin=hin, mode=mode, range_flag=flag)
            out['s2n'] = (o2,)
            out['flag'] = (flag.clone(),)
            # a value beyond fp16's range raises the flag in both forms (mode 2 watches)
            big = hin.clone()
            big[R // 2, 5] = 7e4
            flag2 = torch.zeros(1, dtype=torch.int32, device='cuda')
            A.pointnet_rowmlp_fwd(2, B, Np, W2 * 50., b2, new(R, 128), T=T64, hin=big, mode=mode, range_flag=flag2)
            out['flag_big'] = (flag2,)
            torch.cuda.synchronize()
            return out
        finally:
            A.pointnet_rowmlp_form(before)
    tile, stream = run(1), run(0)
    for key in tile:
        for i, (x, y) in enumerate(zip(tile[key], stream[key])):
            assert torch.equal(x.view(torch.int32) if x.dtype == torch.float32 else x, y.view(torch.int32) if y.dtype == torch.float32 else y), (key, i)
    assert int(tile['flag'][0]) == 0
    if mode == 2:
        assert int(tile['flag_big'][0]) == 1


def _packed_words(h):
    """fp32 -> one word per value (fp16 hi | fp16 lo << 16, lo = the residual scaled by 2^11): what rowmlp_fwd mode 2 writes."""
    hi = h.half()
    lo = ((h - hi.float()) * 2048.).half()
    # (held in a float32-typed tensor, as the engine's own buffers are: ops._dev() would CONVERT an int32 tensor)
    return ((hi.view(torch.int16).int() & 0xffff) | (lo.view(torch.int16).int() << 16)).contiguous().view(torch.float32)


def _v1_filter():
    """tools/experimental/v1_filter.py, or a skip: the filtered layer is in libhitadv_experimental.so (`make -C hit_adv_amd/csrc
    experimental`), which the product neither builds nor loads."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'experimental'))
    import v1_filter
    if not v1_filter.available():
        pytest.skip("libhitadv_experimental.so is not built (make -C hit_adv_amd/csrc experimental)")
    return v1_filter


@pytest.mark.parametrize("B,Np,blocks", [(256, 1024, 128), (64, 1024, 256), (96, 256, 0), (130, 128, 64)])
def test_filtered_linear_max_equals_the_full_evaluation(A, B, Np, blocks):
    """Round 5, csrc/experimental/victim_filter.hip: the 128 -> 1024 layer + max over the points with ONE fp16 product per value and the
    exact evaluation of the candidates only.  Against the unfiltered fp16x2 kernel on the same packed activation: the
    maxima to fp32 roundoff (both are three exact products per term; the summation orders differ), the arg-max EQUAL wherever
    the two best exact values of a channel are further apart than that roundoff, and in any case a point whose value is the
    maximum to roundoff.  The result does not depend on the seeds: zeros, random points, last call's winners -- the same bits."""
    g = torch.Generator().manual_seed(B + Np)
    h = cu(torch.randn(B * Np, 128, generator=g).relu() * torch.rand(B * Np, 1, generator=g))  # rows of different norms
    W = cu(torch.randn(1024, 128, generator=g) * 0.1)
    bias = cu(torch.randn(1024, generator=g) * 0.1)
    X = _v1_filter()
    assert X.linear_max_filter_supported(B, Np, 128, 1024, blocks)
    xp, W2, wn = _packed_words(h), A.split_weights_f16x2(W), X.weight_row_norms(W)
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    full_v, full_i = A.linear_max_fwd_f16x2(xp, W2, B, Np, bias=bias, relu=True, blocks=blocks, packed=True)
    outs = []
    for seeds in (torch.zeros(B, 1024, dtype=torch.int64, device='cuda'),
                  torch.randint(-5, Np + 5, (B, 1024), generator=g).cuda(),
                  full_i.clone()):
        v, i = X.linear_max_fwd_f16x2_filtered(xp, W2, wn, B, Np, seeds, bias=bias, relu=True, blocks=blocks, range_flag=flag)
        assert torch.equal(seeds, i)  # the winners are left as the next call's seeds
        outs.append((v, i))
    assert int(flag.item()) == 0
    for v, i in outs[1:]:
        assert torch.equal(v, outs[0][0]) and torch.equal(i, outs[0][1])
    v, i = outs[0]
    assert int(i.min()) >= 0 and int(i.max()) < Np
    scale = float(full_v.abs().max())
    close(v, full_v, rtol=0, atol=2e-6 * scale, what='maxima')
    # float64 evaluation of the pieces at both winners
    hi, lo = h.half().double(), ((h - h.half().float()) * 2048.).half().double()
    whi, wlo = W.half().double(), ((W - W.half().float()) * 2048.).half().double()
    def exact(points):  # [B,1024] -> value of channel c at points[b,c]
        rows = (torch.arange(B, device='cuda')[:, None] * Np + points).reshape(-1)
        ah, al = hi[rows].view(B, 1024, 128), lo[rows].view(B, 1024, 128)
        return (ah * whi[None]).sum(-1) + ((al * whi[None]).sum(-1) + (ah * wlo[None]).sum(-1)) / 2048.
    e_f, e_full = exact(i), exact(full_i)
    assert float((e_f - e_full).abs().max()) <= 4e-6 * scale       # the filtered winner IS a maximiser (to roundoff)
    differ = i != full_i
    note('channels whose arg-max differs from the unfiltered kernel (near-ties)', float(differ.sum()))
    assert float(differ.double().mean()) < 2e-3
    assert float((e_f - e_full).abs()[differ].max() if differ.any() else 0.) <= 4e-6 * scale


def test_filtered_linear_max_reports_lists_that_do_not_fit(A):
    """A cloud of IDENTICAL points: every point is a candidate for every channel, no list of 2048 holds them (32 channels x 1024 points) -- the kernel
    must say so (range flag) instead of returning a winner it has not checked."""
    X = _v1_filter()
    B, Np = 256, 1024
    g = torch.Generator().manual_seed(1)
    h = cu(torch.randn(B * Np, 128, generator=g).relu())
    h[:Np] = h[0]
    W = cu(torch.randn(1024, 128, generator=g) * 0.1)
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    seeds = torch.zeros(B, 1024, dtype=torch.int64, device='cuda')
    X.linear_max_fwd_f16x2_filtered(_packed_words(h), A.split_weights_f16x2(W), X.weight_row_norms(W), B, Np, seeds, relu=True,
                                    blocks=128, range_flag=flag)
    assert int(flag.item()) == 1
