"""Edge cases of the HIP kernels: ragged / tiny / multi-chunk sizes, misaligned views, maximum supported sizes,
argument errors.  Same parity bar as test_gpu_kernels.py (bit-exact indices and direct-form distances)."""
import numpy as np
import pytest
import torch

from helpers import close, synth_batch
from oracle import c_oracle as N
from oracle import hitadv_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import hit_adv_amd.ops as ops
    return ops


def pts(b, n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(b, n, 3, generator=g) * 2 - 1


def test_pairwise_misaligned_views_and_tiny_sizes(A):
    big = pts(2, 1030, 1)
    x = big[:, 1:1025].cuda()  # storage offset 3 floats -> not 16-byte aligned after .contiguous()? contiguous() realigns,
    y = big[:, 5:1029]         # so also exercise the raw scalar path through odd M below
    for xs, ys in ((x, y.cuda()), (pts(1, 1, 2).cuda(), pts(1, 1, 3).cuda()), (pts(3, 5, 4).cuda(), pts(3, 1026, 5).cuda()),
                   (pts(2, 33, 6).cuda(), pts(2, 1021, 7).cuda())):
        for form in (A.FORM_DIRECT, A.FORM_GRAM):
            P = A.pairwise_sqdist(xs, ys, form).cpu()
            ref = O.pairwise_sqdist_direct(xs.cpu(), ys.cpu())
            if form == A.FORM_DIRECT:
                assert torch.equal(P, ref)
            else:
                assert (P - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("n,m", [(1, 1), (2, 3000), (5000, 3), (2500, 2300)])
def test_nn_min_multi_chunk_and_degenerate(A, n, m):
    x, y = pts(2, n, 10), pts(2, m, 11)
    mx, ax, my, ay = (t.cpu() for t in A.nn_min(x.cuda(), y.cuda()))
    rx, rax = N.nn_min(x, y)
    ry, ray = N.nn_min(y, x)
    assert torch.equal(mx, rx) and torch.equal(ax, rax) and torch.equal(my, ry) and torch.equal(ay, ray)


def test_nn_min_backward_one_direction_only_and_scatter_collisions(A):
    # many y points share one nearest x -> the owner-computes scatter sums several contributions in j order
    x = torch.tensor([[[0., 0, 0], [10., 0, 0]]])
    y = torch.cat([torch.randn(1, 300, 3, generator=torch.Generator().manual_seed(1)) * 0.1,
                   torch.tensor([[[10.1, 0, 0]]])], 1)
    xg = x.cuda().requires_grad_()
    _, _, my, _ = A.nn_min(xg, y.cuda())
    my.sum().backward()  # only the y->x direction carries gradient
    xr = x.clone().requires_grad_()
    O.pairwise_sqdist_direct(xr, y).min(1).values.sum().backward()
    close(xg.grad.cpu(), xr.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("m,K", [(5000, 17), (3, 3), (70, 64), (2050, 6)])
def test_knn_multi_chunk_and_k_equals_m(A, m, K):
    from hit_adv_amd.pytorch3d_ops import knn_points
    p = pts(2, m, 20)
    p[0, m - 1] = p[0, 0]  # exact tie between the first and the last reference (different chunks / waves)
    q = torch.cat([p[:, :7], pts(2, 30, 21)], 1)
    r = knn_points(q.cuda(), p.cuda(), K=K)
    d, ix = N.knn_points(q, p, K)
    assert torch.equal(r.idx.cpu(), ix) and torch.equal(r.dists.cpu(), d)


def test_deform_many_centres_and_odd_point_counts(A):
    g = torch.Generator().manual_seed(30)
    B, Np, C = 2, 77, 1500  # C > 1024: two LDS passes over the centre table
    ori = torch.randn(B, 3, Np, generator=g) * 0.5
    central = torch.randn(B, 3, C, generator=g) * 0.5
    P = ((torch.rand(B, C, 3, generator=g) - 0.5) * 0.2).requires_grad_()
    sig = (0.2 + torch.rand(B, C, generator=g)).requires_grad_()
    up = torch.randn(B, 3, Np, generator=g)
    att = O.HiTADVOracle.__new__(O.HiTADVOracle)
    ref = O.deform_loop(ori, P, O.kernel_density(central, ori, sig))
    (ref * up).sum().backward()
    Pg, sg = P.detach().cuda().requires_grad_(), sig.detach().cuda().requires_grad_()
    adv = A.deform(ori.cuda(), central.cuda(), Pg, sg)
    close(adv.detach().cpu(), ref.detach(), rtol=1e-5, atol=2e-6)
    (adv * up.cuda()).sum().backward()
    close(Pg.grad.cpu(), P.grad, rtol=2e-4, atol=1e-5 * float(P.grad.abs().max()))
    close(sg.grad.cpu(), sig.grad, rtol=2e-4, atol=1e-5 * float(sig.grad.abs().max()))


def test_fps_maximum_and_unsupported_sizes(A):
    from hit_adv_amd import _lib
    x = pts(1, 16384, 40)
    start = torch.tensor([123])
    assert torch.equal(A.fps_from_start(x.cuda(), 40, start.cuda()).cpu(), N.fps_from_start(x, 40, start))
    with pytest.raises(_lib.HitAdvLibraryError):
        A.fps_from_start(pts(1, 16385, 41).cuda(), 4, start.cuda())
    one = pts(2, 1, 42)
    assert A.fps_from_start(one.cuda(), 3, torch.zeros(2, dtype=torch.int64).cuda()).tolist() == [[0, 0, 0], [0, 0, 0]]


@pytest.mark.parametrize("n", [256, 257, 512, 513, 1024, 1025, 2048, 2049, 4080, 4081])
def test_fps_kernel_boundaries_both_samplers(A, n):
    """Every size at which the sampling launcher changes kernel or shape (sampling.hip::launch_fps: <= 256 the 64-bit-key kernel,
    then fps_lean at 2 / 4 / 8 points per lane on 4 waves, 8 waves above 1024, the cloud out of LDS above 4080), with a NaN
    point, for both samplers and for as many samples as the cloud has points at the small sizes."""
    from hit_adv_amd import _lib
    L = _lib.load()
    x = pts(2, n, 900 + n)
    x[1, 3] = float('nan')  # never chosen after the start: its distances compare false
    start = torch.tensor([n - 1, 0])
    m = n if n <= 513 else 70
    y = pts(2, n, 901 + n)
    want, want_pct = N.fps_from_start(x, m, start), N.fps_pct(y, m, start)
    shipped = L.hitadv_debug_fps_form(-1)
    try:
        for form in (0, 1):  # 0: the shipped 64-bit-key kernel; 1: fps_lean (HITADV_FPS_FORM=1)
            L.hitadv_debug_fps_form(form)
            assert torch.equal(A.fps_from_start(x.cuda(), m, start.cuda()).cpu(), want), form
            assert torch.equal(A.fps_pct(y.cuda(), m, start.cuda(), reference=True).cpu(), want_pct), form
    finally:
        L.hitadv_debug_fps_form(shipped)


def test_natives_ragged_sizes(A):
    from hit_adv_amd.pointnet2_ops import _ext
    x = pts(2, 333, 50)
    q = x[:, :45].contiguous() + 0.01
    for r, ns in ((0.3, 7), (0.05, 70), (3.0, 400)):
        assert torch.equal(_ext.ball_query(q.cuda(), x.cuda(), r, ns).cpu(), N.ball_query(q, x, r, ns))
    d2, ix = _ext.three_nn(q.cuda(), x[:, :2].contiguous().cuda())  # fewer than three known points
    rd2, rix = N.three_nn(q, x[:, :2].contiguous())
    assert torch.equal(ix.cpu(), rix) and torch.equal(d2.cpu()[..., :2], rd2[..., :2]) and torch.isinf(d2[..., 2]).all()
    f = _ext.furthest_point_sampling(x.cuda(), 333)
    assert torch.equal(f.cpu(), N.furthest_point_sampling(x, 333))
    assert _ext.furthest_point_sampling(x.cuda(), 0).shape == (2, 0)


def test_attack_state_kernels_small_shapes(A):
    g = torch.Generator().manual_seed(60)
    B, K, Np, C = 1, 10, 50, 3
    st = {k: v.cuda() for k, v in dict(
        bestdist=torch.full((B,), 1e10), bestscore=torch.full((B,), -1, dtype=torch.int64),
        o_bestdist=torch.full((B,), 1e10), o_bestscore=torch.full((B,), -1, dtype=torch.int64),
        o_bestattack=torch.zeros(B, 3, Np), pred=torch.zeros(B, dtype=torch.int64), dist_val=torch.zeros(B)).items()}
    logits = torch.randn(B, K, generator=g)
    label = (logits.argmax(1) + 1) % K
    P, sig, adv = torch.randn(B, C, 3, generator=g), torch.rand(B, C, generator=g), torch.randn(B, 3, Np, generator=g)
    A.best_update(logits.cuda(), label.cuda(), P.cuda(), sig.cuda(), adv.cuda(), st)
    assert st['pred'].cpu().tolist() == logits.argmax(1).tolist()
    assert torch.equal(st['o_bestattack'].cpu(), adv)
    d = torch.zeros((), device='cuda')
    out = A.regulariser(P.cuda(), sig.cuda(), adv.cuda(), (adv * 0.9).cuda(), torch.rand(B, C, generator=g).cuda(),
                        torch.tensor([7.0]).cuda(), (1e-4, 1.0, 1.0), (0.1, 1.2), d)
    assert torch.isfinite(out) and abs(out.item() - 7.0 * d.item()) < 1e-5 * abs(out.item()) + 1e-9


@pytest.mark.parametrize("B,n,C,k", [(2, 2048, 8, 3), (1, 1100, 4, 40), (3, 33, 64, 1), (2, 3000, 4, 2)])
def test_edge_max_backward_reversed_graph_sizes(A, B, n, C, k):
    """The gather-form backward where the bit matrix of the reversed neighbour table needs several passes (N > 1024), N
    is no multiple of 32, lists repeat an entry, hubs collect thousands of edges, and some points are listed by nobody:
    bit for bit the ascending-i sequential sum."""
    g = torch.Generator().manual_seed(n + k)
    U = torch.randn(B, n, C, generator=g).cuda().requires_grad_()
    V = torch.randn(B, n, C, generator=g).cuda().requires_grad_()
    idx = torch.randint(0, n, (B, n, k), generator=g)
    idx[:, ::3, 0] = 7  # a hub
    idx[:, :, -1] = idx[:, :, 0]  # a repeated entry in every list (k > 1)
    idx[idx == 5] = 6  # nobody lists point 5
    w = torch.randn(B, n, C, generator=g).cuda()
    idx_d = idx.cuda()
    out = A.edge_max(U, V, idx_d, 0.2)
    gu, gv = torch.autograd.grad((out * w).sum(), [U, V])
    nbr = U.detach().cpu().gather(1, idx.reshape(B, n * k, 1).expand(B, n * k, C)).view(B, n, k, C)
    top, slot = nbr.max(dim=2)
    ref = torch.nn.functional.leaky_relu(top + V.detach().cpu(), negative_slope=0.2)
    assert torch.equal(out.detach().cpu(), ref)
    arg = torch.gather(idx.unsqueeze(-1).expand(B, n, k, C), 2, slot.unsqueeze(2)).squeeze(2)
    dv = w.cpu() * torch.where(ref > 0, torch.ones(()), torch.full((), 0.2))
    assert torch.equal(gv.cpu(), dv)
    seq = torch.zeros(B, n, C)
    for i in range(n):
        seq.scatter_add_(1, arg[:, i:i + 1, :], dv[:, i:i + 1, :])
    assert torch.equal(gu.cpu(), seq)
    assert float(gu[:, 5].abs().max()) == 0.0


@pytest.mark.parametrize("B,n", [(1, 1000), (3, 2048), (5, 130)])
def test_hit_adv_pointnet_engine_odd_shapes_graph_equals_eager(B, n):
    """HiT-ADV on the HIP PointNet engine at shapes off the bench's (a single cloud, N not a multiple of the 64-point
    tiles, the 2048-point clouds of cfg4): the captured-graph run and the eager run agree bit for bit, and the victim's
    logits at the returned clouds match the plain module's."""
    from hit_adv_amd.ShapeAttack.HiT_ADV import HiT_ADV
    from hit_adv_amd.model.pointnet import PointNetFeatureModel
    from hit_adv_amd.util.adv_utils import UntargetedLogitsAdvLoss
    torch.manual_seed(1)
    m = PointNetFeatureModel(40, normal_channel=False).cuda().eval()
    data, _ = synth_batch(B, n, first=2000)
    with torch.no_grad():
        label = m(data[:, :, :3].transpose(1, 2).contiguous().cuda())[0].argmax(1)
    res = {}
    for graph in (False, True):
        att = HiT_ADV(m, UntargetedLogitsAdvLoss(30.), binary_step=2, num_iter=6, cd_weight=1e-4, ker_weight=1.,
                      hide_weight=1., curv_loss_knn=16, central_num=min(64, n // 4), total_central_num=min(96, n // 2),
                      max_sigm=1.2, min_sigm=0.1, budget=0.55, verbose=False, use_graph=graph)
        torch.manual_seed(4)
        res[graph] = att.attack(data, label)
        assert att.last_graph_used == graph
    assert np.array_equal(res[False][0], res[True][0]) and int(res[False][1]) == int(res[True][1])
    adv = torch.from_numpy(res[True][0]).float().transpose(1, 2).contiguous().cuda()
    with torch.no_grad():
        close(m.attack_view()(adv)[0].cpu().numpy(), m(adv)[0].cpu().numpy(), rtol=1e-3, atol=1e-4)


def test_index_tables_stay_in_range_when_a_cloud_is_not_finite():
    """A cloud that went NaN / inf (a diverged attack; an fp16-range overflow the caller only learns of when it reads its
    results back) must not turn into a wild read: the selection kernels start their lists at a sentinel index that a
    comparison with NaN never replaces, and until round 5 the sentinel reached the gather of the backward pass (a memory
    access fault under CW.attack_concurrently with a sharpened PCT victim).  Every table that leaves a kernel holds valid
    rows; the VALUES of the broken cloud stay NaN, the other clouds are untouched."""
    from hit_adv_amd import ops
    from hit_adv_amd.util.dist_utils import KNNDist
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 300, 3, generator=g)
    bad = x.clone()
    bad[1] = float('nan')          # a whole cloud
    bad[2, 7] = float('inf')       # one point
    bad[2, 9, 1] = float('nan')
    xb = bad.cuda()
    for K in (1, 6, 17):
        for form in (ops.FORM_DIRECT, ops.FORM_GRAM_KNN):
            d, idx = ops.KnnPoints.apply(xb, xb, K, form)
            assert int(idx.min()) >= 0 and int(idx.max()) < 300, (K, form)
            d0, i0 = ops.KnnPoints.apply(x[:1].cuda(), x[:1].cuda(), K, form)
            assert torch.equal(idx[0], i0[0]) and torch.equal(d[0], d0[0])  # the finite cloud: as if alone
    pc = xb.clone().requires_grad_()
    loss = KNNDist(k=5)(pc, batch_avg=False)
    grad, = torch.autograd.grad(loss.sum(), pc)   # used to read p[0x7fffffff * 3]
    torch.cuda.synchronize()
    assert torch.isfinite(grad[0]).all()
    # the 128 -> 1024 layer + max over the points: the arg-max table of a NaN cloud is a valid row too
    h = torch.randn(2 * 256, 128, generator=g).cuda()
    h[:256] = float('nan')
    W = (torch.randn(1024, 128, generator=g) * 0.1).cuda()
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    for out in (ops.linear_max_fwd_f16x2(h, ops.split_weights_f16x2(W), 2, 256, relu=True, range_flag=flag),
                ops.linear_max_fwd_bf16x3(h, ops.split_weights_bf16x3(W), 2, 256, relu=True),
                ops.linear_max_fwd(h, W.t().contiguous(), 2, 256, relu=True)):
        arg = out[1]
        assert int(arg.min()) >= 0 and int(arg.max()) < 256
    assert int(flag.item()) == 1  # ... and the fp16x2 form said so
