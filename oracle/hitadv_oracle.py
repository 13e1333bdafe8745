"""CPU ORACLE -- test infrastructure, NOT product code.

A plain-PyTorch (CPU, fp32) restatement of the reference's algorithm for the
HiT-ADV hot path.  It exists so that the HIP path can be checked against
something that runs without a GPU and without the reference tree.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module; the product package ``hit_adv_amd`` never does.

Pinning: every function below is checked in ``tests/test_oracle_golden.py``
against vectors captured from the reference itself (imported unmodified in the
build container by ``tests/golden/make_golden.py``).  Two boundaries stay
**parity-unpinned** because the reference delegates them to code that is not in
its tree and ships no tests for them:
  * ``pytorch3d.ops.knn_points`` (pytorch3d==0.7.2, requirements.txt:10): the
    canonical rule used here is fp32 direct difference ``((dx*dx+dy*dy)+dz*dz)``,
    ascending, ties -> lower index;
  * the ``pointnet2_ops`` CUDA extension (restated in ``pointnet2_oracle.c``
    from the .cu sources; no nvcc / GPU here to run the original).

All ``file:line`` citations are relative to the reference root.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# set distances                                   util/set_distance.py:15-74
# --------------------------------------------------------------------------


def pairwise_sqdist_gram(x, y):
    """Gram-expansion squared distances, P[b,i,j] = |x_i|^2 + |y_j|^2 - 2 x_i.y_j.

    Follows util/set_distance.py:15-32 (three bmm's, diagonal gather).
    x: [B,N,D], y: [B,M,D] -> [B,N,M]
    """
    gx = torch.bmm(x, x.transpose(1, 2))
    gy = torch.bmm(y, y.transpose(1, 2))
    gxy = torch.bmm(x, y.transpose(1, 2))
    rx = torch.diagonal(gx, dim1=1, dim2=2)  # [B,N]
    ry = torch.diagonal(gy, dim1=1, dim2=2)  # [B,M]
    return rx[:, :, None] + ry[:, None, :] - 2 * gxy


def chamfer(preds, gts):
    """util/set_distance.py:40-50 -> (loss1[B] preds->gts, loss2[B] gts->preds)."""
    P = pairwise_sqdist_gram(gts, preds)  # [B,N2,N1]
    loss1 = P.min(dim=1).values.mean(dim=1)
    loss2 = P.min(dim=2).values.mean(dim=1)
    return loss1, loss2


def hausdorff(preds, gts):
    """util/set_distance.py:58-70."""
    P = pairwise_sqdist_gram(gts, preds)
    loss1 = P.min(dim=1).values.max(dim=1).values
    loss2 = P.min(dim=2).values.max(dim=1).values
    return loss1, loss2


def _weighted(loss, weights, batch_avg):
    if weights is None:
        weights = torch.ones(loss.shape[0])
    loss = loss * weights.float()
    return loss.mean() if batch_avg else loss


def _pick(method, l1, l2):
    if method == 'adv2ori':
        return l1
    if method == 'ori2adv':
        return l2
    return (l1 + l2) / 2.


def chamfer_dist(adv_pc, ori_pc, weights=None, batch_avg=True, method='adv2ori'):
    """util/dist_utils.py:44-80."""
    l1, l2 = chamfer(adv_pc, ori_pc)
    return _weighted(_pick(method, l1, l2), weights, batch_avg)


def hausdorff_dist(adv_pc, ori_pc, weights=None, batch_avg=True, method='adv2ori'):
    """util/dist_utils.py:83-119."""
    l1, l2 = hausdorff(adv_pc, ori_pc)
    return _weighted(_pick(method, l1, l2), weights, batch_avg)


def l2_dist(adv_pc, ori_pc, weights=None, batch_avg=True):
    """util/dist_utils.py:15-41."""
    d = torch.sqrt(torch.sum((adv_pc - ori_pc) ** 2, dim=[1, 2]) + torch.tensor(1e-7))
    return _weighted(d, weights, batch_avg)


def knn_dist_values(pc, k):
    """Per-point mean squared distance to the k nearest neighbours, [B,K].

    util/dist_utils.py:144-159 (Gram form, topk of the negated matrix, drop rank 0).
    """
    if pc.shape[1] != 3:
        pc = pc.transpose(2, 1)  # [B,3,K]
    inner = -2. * torch.matmul(pc.transpose(2, 1), pc)
    sq = torch.sum(pc ** 2, dim=1, keepdim=True)
    dist = sq + inner + sq.transpose(2, 1)
    neg, _ = (-dist).topk(k=k + 1, dim=-1)
    return torch.mean(-(neg[..., 1:]), dim=-1)


def knn_dist(pc, weights=None, batch_avg=True, k=5, alpha=1.05):
    """util/dist_utils.py:136-175."""
    value = knn_dist_values(pc, k)
    with torch.no_grad():
        thr = value.mean(dim=-1) + alpha * value.std(dim=-1)
        mask = (value > thr[:, None]).float()
    return _weighted(torch.mean(value * mask, dim=1), weights, batch_avg)


def chamfer_knn_dist(adv_pc, ori_pc, weights=None, batch_avg=True, method='adv2ori',
                     knn_k=5, knn_alpha=1.05, chamfer_weight=5., knn_weight=3.):
    """util/dist_utils.py:258-294."""
    c = chamfer_dist(adv_pc, ori_pc, weights, batch_avg, method)
    n = knn_dist(adv_pc, weights, batch_avg, knn_k, knn_alpha)
    return c * chamfer_weight + n * knn_weight


# --------------------------------------------------------------------------
# canonical kNN (stand-in semantics for pytorch3d, see module docstring)
# --------------------------------------------------------------------------


def pairwise_sqdist_direct(p1, p2):
    """((dx*dx + dy*dy) + dz*dz), one fp32 rounding per op, [B,N,M]."""
    acc = None
    for d in range(p1.shape[-1]):
        diff = p1[:, :, None, d] - p2[:, None, :, d]
        sq = diff * diff
        acc = sq if acc is None else acc + sq
    return acc


def knn_points(p1, p2, K):
    """-> (dists[B,N,K] ascending, idx[B,N,K] int64); ties -> lower index."""
    d = pairwise_sqdist_direct(p1.float(), p2.float())
    s = torch.sort(d, dim=-1, stable=True)
    return s.values[..., :K].contiguous(), s.indices[..., :K].contiguous()


def knn_gather(x, idx):
    """x[B,M,U], idx[B,N,K] -> [B,N,K,U]."""
    B, M, U = x.shape
    _, N, K = idx.shape
    return torch.gather(x, 1, idx.reshape(B, N * K, 1).expand(B, N * K, U)).reshape(B, N, K, U)


# --------------------------------------------------------------------------
# adversarial losses                                  util/adv_utils.py:6-85
# --------------------------------------------------------------------------


def _real_other(logits, targets):
    onehot = torch.zeros_like(logits).scatter_(1, targets.view(-1, 1).long(), 1.)
    real = torch.sum(onehot * logits, dim=1)
    other = torch.max((1. - onehot) * logits - onehot * 10000., dim=1)[0]
    return real, other


def logits_adv_loss(logits, targets, kappa=0.):
    """util/adv_utils.py:18-35 (targeted)."""
    real, other = _real_other(logits, targets)
    return torch.clamp(other - real + kappa, min=0.).mean()


def untargeted_logits_adv_loss(logits, targets, kappa=0.):
    """util/adv_utils.py:50-67."""
    real, other = _real_other(logits, targets)
    return torch.clamp(real - other + kappa, min=0.).mean()


def cross_entropy_adv_loss(logits, targets):
    """util/adv_utils.py:77-85."""
    return F.cross_entropy(logits, targets)


# --------------------------------------------------------------------------
# clipping / projection                              util/clip_utils.py:5-170
# --------------------------------------------------------------------------


def clip_points_l2(pc, ori_pc, budget):
    """util/clip_utils.py:17-32."""
    diff = pc - ori_pc
    norm = torch.sum(diff ** 2, dim=[1, 2]) ** 0.5
    scale = torch.clamp(budget / (norm + 1e-9), max=1.)
    return ori_pc + diff * scale[:, None, None]


def clip_points_linf(pc, ori_pc, budget):
    """util/clip_utils.py:75-87."""
    return ori_pc + torch.clamp(pc - ori_pc, min=-budget, max=budget)


def project_inner_points(pc, ori_pc, normal=None):
    """util/clip_utils.py:98-140.  Quirk Q7 kept: the second cross product (:121) is written without ``dim``, which
    under the reference's PyTorch means "the first dimension of size 3" -- the BATCH dimension when B == 3
    (fixture g24 has B = 3, g6 has B = 2)."""
    if normal is None:
        return pc
    diff = pc - ori_pc
    inner = torch.sum(diff * normal, dim=1) < 0.
    vng = torch.cross(normal, diff, dim=1)
    vng_norm = torch.sum(vng ** 2, dim=1) ** 0.5
    vref = torch.cross(vng, normal, dim=0 if pc.shape[0] == 3 else 1)
    vref_norm = torch.sum(vref ** 2, dim=1) ** 0.5
    proj = diff * vref / (vref_norm[:, None, :] + 1e-9)
    opposite = (inner & (vng_norm < 1e-6))[:, None, :].expand_as(proj)
    proj = torch.where(opposite, torch.zeros_like(proj), proj)
    diff = torch.where(inner[:, None, :].expand_as(diff), proj, diff)
    return ori_pc + diff


def project_inner_clip_linf(pc, ori_pc, normal, budget):
    """util/clip_utils.py:157-170."""
    return clip_points_linf(project_inner_points(pc, ori_pc, normal), ori_pc, budget)


# --------------------------------------------------------------------------
# HiT-ADV building blocks                         ShapeAttack/HiT_ADV.py
# --------------------------------------------------------------------------


def _unit(v, dim=1, eps=1e-12):
    """ShapeAttack/HiT_ADV.py:534-535."""
    return v / v.norm(2, dim, keepdim=True).clamp(min=eps).expand_as(v)


def kappa_ori(pc, normal, k):
    """Curvature proxy, ShapeAttack/HiT_ADV.py:318-325.  pc, normal: [B,3,N] -> [B,N]."""
    pts = pc.permute(0, 2, 1)
    _, idx = knn_points(pts, pts, k + 1)
    nbr = knn_gather(pts, idx).permute(0, 3, 1, 2)[:, :, :, 1:].contiguous()  # [B,3,N,k]
    vec = _unit(nbr - pc.unsqueeze(3))
    return torch.abs((vec * normal.unsqueeze(3)).sum(1)).mean(2), idx


def kappa_std_ori(pc, normal, k):
    """Std (unbiased) of the neighbours' curvature proxy, ShapeAttack/HiT_ADV.py:327-339."""
    kap, idx = kappa_ori(pc, normal, k)
    nbr_kap = knn_gather(kap.unsqueeze(2), idx).permute(0, 3, 1, 2)[:, :, :, 1:].contiguous()
    return torch.std(nbr_kap.squeeze(1), dim=2)


def curv_std_dist(ori_data, adv_data, ori_normal, k=5):
    """util/dist_utils.py:464-495 (CurvStdDist.forward)."""
    a = kappa_std_ori(ori_data, ori_normal, k)
    b = kappa_std_ori(adv_data, ori_normal, k)
    return torch.nn.PairwiseDistance(p=2)(a, b).mean()


def fps_from_start(xyz, npoint, start):
    """Iterative farthest point sampling with a given first index.

    ShapeAttack/HiT_ADV.py:489-510 with the CPU ``torch.randint`` draw (:501)
    passed in as ``start`` [B] int64.  xyz [B,N,3] -> [B,npoint] int64.
    """
    B, N, _ = xyz.shape
    out = torch.zeros(B, npoint, dtype=torch.long)
    running = torch.full((B, N), 1e10)
    far = start.clone()
    rows = torch.arange(B)
    for i in range(npoint):
        out[:, i] = far
        c = xyz[rows, far, :].view(B, 1, 3)
        d = torch.sum((xyz - c) ** 2, -1)
        running = torch.where(d < running, d, running)
        far = torch.max(running, -1)[1]
    return out


def take_points(points, idx):
    """points[B,N,C], idx[B,...] -> points[b, idx[b,...], :]  (HiT_ADV.py:470-487)."""
    B = points.shape[0]
    shape = [B] + [1] * (idx.dim() - 1)
    rows = torch.arange(B).view(shape).expand_as(idx)
    return points[rows, idx, :]


def kernel_density(central, pc, delta):
    """exp(-|x_n - c_j| / (2 sigma_j^2)), un-squared norm.  HiT_ADV.py:298-304.

    central [B,3,C], pc [B,3,N], delta [B,C] -> [B,C,N].  Materialised with the
    same two ``repeat``s as the reference so that its cost is representative.
    """
    C = central.shape[2]
    N = pc.shape[2]
    a = pc.unsqueeze(3).repeat(1, 1, 1, C)
    b = central.unsqueeze(2).repeat(1, 1, N, 1)
    nrm = torch.norm(a - b, dim=1)  # [B,N,C]
    dens = torch.exp(-nrm / (2 * delta * delta).unsqueeze(1))
    return dens.transpose(1, 2).contiguous()


def deform_loop(ori, perturb, ker):
    """Kernel-weighted deformation as the C-step accumulation loop, HiT_ADV.py:160-175."""
    B, _, N = ori.shape
    num = torch.zeros_like(ori)
    den = torch.zeros(B, 1, N)
    for j in range(perturb.shape[1]):
        num += (ori + perturb[:, j, :].unsqueeze(2)) * ker[:, j, :].unsqueeze(1)
        den += ker[:, j, :].unsqueeze(1)
    return num / den


def transformation_loss(perturb, delta, central_num, batch_avg=True):
    """HiT_ADV.py:306-316; note the whole-batch norm when batch_avg (quirk Q3)."""
    if batch_avg:
        t = torch.tensor(0.0)
        t = t + torch.norm(perturb)
        t = t + 1 * torch.norm(1 - delta)
    else:
        t = torch.zeros(perturb.shape[0])
        t = t + torch.norm(perturb, dim=(1, 2))
        t = t + 1 * torch.norm(1 - delta, dim=1)
    return t / central_num


def curv_std_loss(delta, central_kappa_std, max_delta, min_delta):
    """HiT_ADV.py:341-346; global (whole-batch) min/max (quirk Q3)."""
    lo = torch.min(central_kappa_std)
    hi = torch.max(central_kappa_std)
    ns = (central_kappa_std - lo) / (hi - lo + 1e-7)
    nd = (delta - min_delta) / (max_delta - min_delta + 1e-7)
    return F.cosine_similarity(ns.squeeze(-1), nd)


def input_gradient(model, data, target):
    """d CE / d xyz and the clean miss count.  HiT_ADV.py:537-559."""
    x = data.clone().detach().float().requires_grad_()
    logits = model(x)
    if isinstance(logits, tuple):
        logits = logits[0]
    F.cross_entropy(logits, target).backward()
    with torch.no_grad():
        miss = (torch.argmax(logits, dim=-1) != target).sum().item()
    return x.grad.detach(), miss


def select_centres(ori, normal, grad, fps_start, k, total_central_num, central_num, alpha=1):
    """Saliency/curvature scoring and centre selection.  HiT_ADV.py:61-93,118-123.

    ori, normal, grad: [B,3,N].  Returns a dict with every intermediate the
    golden fixture G4 pins.
    """
    B = ori.shape[0]
    kstd = kappa_std_ori(ori, normal, k)
    centre = torch.median(ori, dim=-1)[0]
    off = ori - centre[:, :, None]
    r = torch.sum(off ** 2, dim=1) ** 0.5
    sal = -1. * (r ** alpha) * torch.sum(off * grad, dim=1)
    sal_n = (sal - torch.min(sal)) / (torch.max(sal) - torch.min(sal) + 1e-7)
    std_n = (kstd - torch.min(kstd)) / (torch.max(kstd) - torch.min(kstd) + 1e-7)
    score = 0.001 * sal_n + std_n

    pts = ori.transpose(1, 2).contiguous()
    far_idx = fps_from_start(pts, total_central_num, fps_start)
    far_pts = take_points(pts, far_idx)
    _, nbr_idx = knn_points(far_pts, pts, k + 1)  # [B,T,k+1]
    nbr_pts = knn_gather(pts, nbr_idx)  # [B,T,k+1,3]
    nbr_score = take_points(score.unsqueeze(2), nbr_idx)  # [B,T,k+1,1]
    pick = nbr_score.topk(k=1, dim=2)[1].squeeze(dim=-1)  # [B,T,1]

    cand = take_points(nbr_pts.reshape(-1, k + 1, 3), pick.view(-1, 1)).view(B, -1, 3)
    cand_score = take_points(nbr_score.view(-1, k + 1, 1), pick.view(-1, 1)).view(B, -1)
    top_score, top_idx = torch.topk(cand_score, k=central_num)
    central = take_points(cand, top_idx).transpose(1, 2).contiguous()  # [B,3,C]

    kap, _ = kappa_ori(ori, normal, k)
    nbr_kap = take_points(kap.unsqueeze(2), nbr_idx)
    cand_kap = take_points(nbr_kap.view(-1, k + 1, 1), pick.view(-1, 1)).view(B, -1, 1)
    central_kap = take_points(cand_kap, top_idx)  # [B,C,1]
    return dict(kappa_std=kstd, saliency=sal, score=score, far_idx=far_idx, nbr_idx=nbr_idx,
                pick=pick, cand_score=cand_score, top_idx=top_idx, central=central,
                central_kappa=central_kap)


class HiTADVOracle:
    """CPU restatement of ShapeAttack/HiT_ADV.py::HiT_ADV (ctor :18-42, attack :44-287).

    Same algorithm, same RNG draw order from the global CPU generator
    (randint for the FPS start :501, then per binary step rand(B,C,3) :130 and
    rand(B,C) :133), same host-side best/bisection bookkeeping.  ``trace`` (a
    list) receives one dict per inner iteration when given.
    """

    def __init__(self, model, adv_func, attack_lr=1e-2, init_weight=10., max_weight=80.,
                 binary_step=10, num_iter=500, clip_func=None, cd_weight=0, curv_weight=0,
                 ker_weight=0, hide_weight=0, curv_loss_knn=32, central_num=32,
                 total_central_num=128, max_sigm=0.7, min_sigm=0.1, budget=0.1, alpha=1):
        model.eval()  # (not `model = model.eval()`: the reference's FeatureModel.eval() returns None)
        self.model = model
        self.adv_func = adv_func
        self.hp = dict(attack_lr=attack_lr, init_weight=init_weight, max_weight=max_weight,
                       binary_step=binary_step, num_iter=num_iter, cd_weight=cd_weight,
                       ker_weight=ker_weight, hide_weight=hide_weight, k=curv_loss_knn,
                       C=central_num, T=total_central_num, max_sigm=max_sigm, min_sigm=min_sigm,
                       budget=budget, alpha=alpha)

    def _logits(self, x):
        out = self.model(x)
        return out[0] if isinstance(out, tuple) else out

    def inner_iteration(self, st):
        """One pass of HiT_ADV.py:156-246.  ``st`` is the mutable per-step state dict."""
        hp = self.hp
        P, sig, ori, target = st['P'], st['sigma'], st['ori'], st['target']
        with torch.no_grad():
            P.data = torch.clamp(P.data, min=-hp['budget'], max=hp['budget'])
            sig.data = torch.clamp(sig.data, min=hp['min_sigm'], max=hp['max_sigm'])
        ker = kernel_density(st['central'], ori, sig)
        adv = deform_loop(ori, P, ker)
        logits = self._logits(adv)
        pred = torch.argmax(logits, dim=1)
        dist_val = transformation_loss(P, sig, hp['C'], batch_avg=False)

        pred_np = pred.detach().numpy()
        adv_np = adv.detach().numpy()
        for e in range(ori.shape[0]):
            d = dist_val[e].item()
            if pred_np[e] != st['label'][e]:
                if d < st['bestdist'][e]:
                    st['bestdist'][e] = d
                    st['bestscore'][e] = pred_np[e]
                if d < st['o_bestdist'][e]:
                    st['o_bestdist'][e] = d
                    st['o_bestscore'][e] = pred_np[e]
                    st['o_bestattack'][e] = adv_np[e]
                    st['taken'][e] = st.get('at', (-1, -1))  # (binary step, iteration) of the record's last replacement

        adv_loss = self.adv_func(logits, target)
        dist_loss = torch.tensor(0.)
        if hp['cd_weight'] != 0:
            w = torch.full((ori.shape[0],), float(hp['cd_weight']), dtype=torch.float64)
            # quirk Q1: [B,3,N] tensors fed to a [B,K,3] operator (HiT_ADV.py:230)
            dist_loss = dist_loss + chamfer_dist(adv, ori, w)
        if hp['ker_weight'] != 0:
            dist_loss = dist_loss + transformation_loss(P, sig, hp['C']) * hp['ker_weight']
        if hp['hide_weight'] != 0:
            hide = curv_std_loss(sig, st['central_kappa'], hp['max_sigm'], hp['min_sigm'])
            dist_loss = dist_loss + (hide * hp['hide_weight']).mean()
        loss = adv_loss + st['scale_const'].float() * dist_loss
        st['opt'].zero_grad()
        loss.mean().backward()
        st['opt'].step()
        st['last_adv'] = adv_np
        st['last_dist'] = dist_val.detach().numpy()
        return dict(adv_loss=adv_loss.item(), dist_loss=float(dist_loss.detach()),
                    pred=pred_np.copy(), adv=adv_np.copy(), dist_val=st['last_dist'].copy(),
                    loss=loss.mean().item())

    def prepare(self, data, target):
        """HiT_ADV.py:51-123: split, score, select centres, allocate bookkeeping."""
        hp = self.hp
        B, N = data.shape[:2]
        ori = data[:, :, :3].float().clone().detach().transpose(1, 2).contiguous()
        normal = data[:, :, 3:].float().clone().detach().transpose(1, 2).contiguous()
        target = target.long().detach()
        grad, _ = input_gradient(self.model, ori, target)
        with torch.no_grad():
            start = torch.randint(0, N, (B,), dtype=torch.long)
            sel = select_centres(ori, normal, grad, start, hp['k'], hp['T'], hp['C'], hp['alpha'])
        st = dict(ori=ori, normal=normal, target=target, label=target.numpy(),
                  central=sel['central'], central_kappa=sel['central_kappa'], sel=sel,
                  lower=torch.zeros(B), upper=torch.ones(B) * hp['max_weight'],
                  scale_const=torch.ones(B) * hp['init_weight'],
                  o_bestdist=np.array([1e10] * B), o_bestscore=np.array([-1] * B),
                  o_bestattack=np.zeros((B, 3, N)), taken=-np.ones((B, 2), dtype=np.int64), steps=[])
        return st

    def begin_step(self, st):
        """HiT_ADV.py:126-145: fresh parameters + Adam for one binary-search step."""
        hp = self.hp
        B = st['ori'].shape[0]
        P = (torch.rand(B, hp['C'], 3) * torch.tensor(hp['budget']))
        sig = torch.ones((B, hp['C'])) * hp['min_sigm'] + torch.rand((B, hp['C'])) * (
            hp['max_sigm'] - hp['min_sigm'])
        st['P'] = P.requires_grad_()
        st['sigma'] = sig.requires_grad_()
        st['bestdist'] = np.array([1e10] * B)
        st['bestscore'] = np.array([-1] * B)
        st['opt'] = torch.optim.Adam([
            {'params': st['P'], 'lr': hp['attack_lr'] * 5},
            {'params': st['sigma'], 'lr': hp['attack_lr'] * 3}], weight_decay=0.)

    def end_step(self, st):
        """Per-sample bisection of the distance weight, HiT_ADV.py:264-273."""
        for e, label in enumerate(st['label']):
            ok = (st['bestscore'][e] != label and st['bestscore'][e] != -1
                  and st['bestdist'][e] <= st['o_bestdist'][e])
            if ok:
                st['lower'][e] = max(st['lower'][e], st['scale_const'][e])
            else:
                st['upper'][e] = min(st['upper'][e], st['scale_const'][e])
            st['scale_const'][e] = (st['lower'][e] + st['upper'][e]) / 2.
        # the bookkeeping as it stands when a step's bisection is done (fixture g5c holds the reference's)
        st['steps'].append(dict(lower=st['lower'].numpy().astype(np.float64), upper=st['upper'].numpy().astype(np.float64),
                                scale_const=st['scale_const'].numpy().astype(np.float64),
                                o_bestdist=st['o_bestdist'].astype(np.float64), o_bestscore=st['o_bestscore'].astype(np.float64),
                                bestdist=st['bestdist'].astype(np.float64), bestscore=st['bestscore'].astype(np.float64)))

    def finish(self, st):
        """Failure fill and return value, HiT_ADV.py:277-287."""
        fail = (st['lower'] == 0.)
        for e in range(fail.shape[0]):
            if fail[e]:
                st['o_bestattack'][e] = st['last_adv'][e]
                st['o_bestdist'][e] = st['last_dist'][e]
        return st['o_bestattack'].transpose((0, 2, 1)), (st['lower'] > 0.).sum()

    def attack(self, data, target, trace=None):
        st = self.prepare(data, target)
        for step in range(self.hp['binary_step']):
            self.begin_step(st)
            for it in range(self.hp['num_iter']):
                st['at'] = (step, it)
                rec = self.inner_iteration(st)
                if trace is not None:
                    rec.update(step=step, it=it, P=st['P'].detach().clone().numpy(),
                               sigma=st['sigma'].detach().clone().numpy())
                    trace.append(rec)
            self.end_step(st)
        self.state = st
        return self.finish(st)


def cw_knn_attack(model, adv_func, dist_func, clip_func, data, target, attack_lr=1e-3,
                  num_iter=2500, trace=None, pre_head=None, untargeted=False):
    """CPU restatement of CW/kNN.py::CWKNN.attack (:40-151) and, with ``untargeted=True``, of
    CW/UKNN.py::CWUKNN.attack (:41-159).

    data [B,N,3] or [B,N,6] (normals are split off, :58-62; only CWUKNN hands them on, to its clip, UKNN.py:120-122).
    Draws ``torch.randn(B,3,N)`` from the global CPU generator (:65).
    ``dist_func(adv[B,N,3], ori[B,N,3])``; ``clip_func(adv[B,3,N], ori[B,3,N])`` (CWKNN) or
    ``clip_func(adv, ori, normal)`` (CWUKNN).  ``pre_head`` (CWUKNN only, UKNN.py:82-85,141-144) runs in front of
    every victim forward.  Success means ``pred == target`` (kNN.py:86,146) or ``pred != target`` (UKNN.py:95,149).
    """
    B, N = data.shape[:2]
    pc = data.float().detach().transpose(1, 2).contiguous()
    normal = None if pc.shape[1] == 3 else pc[:, 3:, :]
    ori = pc[:, :3, :].clone().detach()
    target = target.long().detach()
    adv = (ori.clone().detach() + torch.randn((B, 3, N)) * 1e-7).requires_grad_()
    opt = torch.optim.Adam([adv], lr=attack_lr, weight_decay=0.)

    def logits_of(x):
        out = model(pre_head(x) if pre_head is not None else x)
        return out[0] if isinstance(out, tuple) else out

    for it in range(num_iter):
        logits = logits_of(adv)
        adv_loss = adv_func(logits, target).mean()
        dist_loss = dist_func(adv.transpose(1, 2).contiguous(),
                              ori.transpose(1, 2).contiguous()).mean() * N
        loss = adv_loss + dist_loss
        opt.zero_grad()
        loss.backward()
        opt.step()
        if clip_func is not None:
            adv.data = clip_func(adv.clone().detach(), ori, normal) if untargeted else clip_func(adv.clone().detach(), ori)
        if trace is not None:
            trace.append(dict(it=it, adv_loss=adv_loss.item(), dist_loss=dist_loss.item(),
                              adv=adv.detach().clone().numpy()))
    with torch.no_grad():
        pred = torch.argmax(logits_of(adv), dim=-1)
        success = ((pred != target) if untargeted else (pred == target)).sum().item()
    return adv.transpose(1, 2).contiguous().detach().numpy(), success


def cw_uknn_attack(model, adv_func, dist_func, clip_func, data, target, attack_lr=1e-3, num_iter=2500, pre_head=None,
                   trace=None):
    """CW/UKNN.py::CWUKNN.attack (:41-159): the kNN attack with the untargeted criterion, ``pre_head`` and a clip
    that sees the normals."""
    return cw_knn_attack(model, adv_func, dist_func, clip_func, data, target, attack_lr=attack_lr, num_iter=num_iter,
                         trace=trace, pre_head=pre_head, untargeted=True)


def cw_perturb_attack(model, adv_func, dist_func, data, target, attack_lr=1e-2, init_weight=10.,
                      max_weight=80., binary_step=10, num_iter=500, clip_func=None, trace=None, _n_points=None):
    """CPU restatement of CW/Perturb.py::CWPerturb.attack (:46-202), host-side bookkeeping as in the
    reference (numpy float64 bounds, strict '<' best tracking, success = pred == target)."""
    B, N = data.shape[:2]
    pc = data.float().detach()
    if _n_points is not None:  # CWPerturbT: the caller already transposed to [B,3,N] (PerturbT.py:53)
        N = _n_points
    else:
        if pc.shape[1] > 6:
            pc = pc.transpose(1, 2).contiguous()
        if pc.shape[1] == 6:
            pc = pc[:, :3, :]
    ori = pc.clone().detach()
    target = target.long().detach()
    label = target.numpy()
    lower, upper = np.zeros((B,)), np.ones((B,)) * max_weight
    weight = np.ones((B,)) * init_weight
    o_bestdist, o_bestscore = np.array([1e10] * B), np.array([-1] * B)
    o_bestattack = np.zeros((B, 3, N))
    last_input = None
    for step in range(binary_step):
        adv = (ori.clone().detach() + torch.randn((B, 3, N)) * 1e-7).requires_grad_()
        bestdist, bestscore = np.array([1e10] * B), np.array([-1] * B)
        opt = torch.optim.Adam([adv], lr=attack_lr, weight_decay=0.)
        for it in range(num_iter):
            out = model(adv)
            logits = out[0] if isinstance(out, tuple) else out
            pred = torch.argmax(logits, dim=1).numpy()
            dist_val = torch.sqrt(torch.sum((adv - ori) ** 2, dim=[1, 2])).detach().numpy()
            last_input = adv.detach().numpy().copy()
            for e in range(B):
                if pred[e] == label[e]:
                    if dist_val[e] < bestdist[e]:
                        bestdist[e], bestscore[e] = dist_val[e], pred[e]
                    if dist_val[e] < o_bestdist[e]:
                        o_bestdist[e], o_bestscore[e] = dist_val[e], pred[e]
                        o_bestattack[e] = last_input[e]
            adv_loss = adv_func(logits, target).mean()
            dist_loss = dist_func(adv, ori, torch.from_numpy(weight)).mean()
            opt.zero_grad()
            (adv_loss + dist_loss).backward()
            opt.step()
            if clip_func is not None:
                adv.data = clip_func(adv.clone().detach(), ori)
            if trace is not None:
                trace.append(dict(step=step, it=it, adv_loss=adv_loss.item(), dist_loss=dist_loss.item(),
                                  adv=adv.detach().clone().numpy()))
        for e in range(B):
            if bestscore[e] == label[e] and bestscore[e] != -1 and bestdist[e] <= o_bestdist[e]:
                lower[e] = max(lower[e], weight[e])
            else:
                upper[e] = min(upper[e], weight[e])
            weight[e] = (lower[e] + upper[e]) / 2.
    for e in range(B):
        if lower[e] == 0.:
            o_bestattack[e] = last_input[e]
    return o_bestattack.transpose((0, 2, 1)), int((lower > 0.).sum()), dict(lower=lower, upper=upper)


def laplace_eig(pc, k=30):
    """CW/AOF.py:12-51 on CPU: Gram-form kNN (topk of the negated distances), A = exp(-d^2) (direct form) on the
    symmetrised graph, L = D - A, eigendecomposition (torch.symeig there, torch.linalg.eigh here)."""
    with torch.no_grad():
        inner = -2 * torch.matmul(pc.transpose(2, 1), pc)
        xx = torch.sum(pc ** 2, dim=1, keepdim=True)
        idx = (-xx - inner - xx.transpose(2, 1)).topk(k=k, dim=-1)[1]
        p = pc.transpose(2, 1).contiguous()
        A = torch.exp(-torch.sum((p.unsqueeze(2) - p.unsqueeze(1)).square(), dim=3))
        mask = torch.zeros_like(A).scatter_(2, idx, 1)
        mask = mask + mask.transpose(2, 1)
        mask[mask > 1] = 1
        A = A * mask
        L = torch.diag_embed(torch.sum(A, dim=2)) - A
        return torch.linalg.eigh(L)


def cw_aof_attack(model, adv_func, clip_func, data, target, attack_lr=1e-2, binary_step=2, num_iter=200,
                  GAMMA=0.5, low_pass=100, trace=None):
    """CPU restatement of CW/AOF.py::CWAOF.attack (:83-241)."""
    B, N = data.shape[:2]
    pc = data.float().detach().transpose(1, 2).contiguous()
    if pc.shape[1] == 6:
        pc = pc[:, :3, :]
    ori = pc.clone().detach()
    target = target.long().detach()
    label = target.numpy()
    o_bestdist, o_bestscore = np.array([1e10] * B), np.array([-1] * B)
    o_bestattack = np.zeros((B, 3, N))
    input_val = None

    def logits_of(x):
        out = model(x)
        return out[0] if isinstance(out, tuple) else out

    for step in range(binary_step):
        adv = ori.clone().detach() + torch.randn((B, 3, N)) * 1e-7
        _, V = laplace_eig(adv)
        projs = torch.bmm(adv, V)
        hfc = torch.bmm(projs[..., low_pass:], V[..., low_pass:].transpose(2, 1)).detach().clone()
        lfc = torch.bmm(projs[..., :low_pass], V[..., :low_pass].transpose(2, 1)).detach().clone().requires_grad_()
        opt = torch.optim.Adam([lfc], lr=attack_lr, weight_decay=0.)
        for it in range(num_iter):
            adv_loss = (1 - GAMMA) * adv_func(logits_of(lfc + hfc), target).mean()
            opt.zero_grad()
            adv_loss.backward()
            (GAMMA * adv_func(logits_of(lfc), target).mean()).backward()
            opt.step()
            with torch.no_grad():
                adv = clip_func((lfc + hfc).detach().clone(), ori)
                coeff = torch.bmm(adv, V)
                hfc.data = torch.bmm(coeff[..., low_pass:], V[..., low_pass:].transpose(2, 1))
                lfc.data = torch.bmm(coeff[..., :low_pass], V[..., :low_pass].transpose(2, 1))
                pred = torch.argmax(logits_of(adv), dim=1).numpy()
                lfc_pred = torch.argmax(logits_of(lfc), dim=1).numpy()
            dist_val = torch.sqrt(torch.sum((adv - ori) ** 2, dim=[1, 2])).numpy()
            input_val = adv.detach().numpy().copy()
            for e in range(B):
                if dist_val[e] < o_bestdist[e] and pred[e] != label[e] and (lfc_pred[e] != label[e] or GAMMA < 0.001):
                    o_bestdist[e], o_bestscore[e] = dist_val[e], pred[e]
                    o_bestattack[e] = input_val[e]
            if trace is not None:
                trace.append(dict(step=step, it=it, adv=input_val.copy()))
    for e in range(B):
        if o_bestscore[e] < 0:
            o_bestattack[e] = input_val[e]
    adv_pc = clip_func(torch.tensor(o_bestattack).to(ori), ori)
    success = (torch.argmax(logits_of(adv_pc), dim=-1) != target).sum().item()
    return adv_pc.detach().numpy().transpose((0, 2, 1)), success



def cw_perturbt_attack(model, adv_func, dist_func, data, target, **kw):
    """CW/PerturbT.py::CWPerturbT.attack (:44-183) = CWPerturb.attack on the always-transposed input (:53), no pre_head."""
    return cw_perturb_attack(model, adv_func, dist_func, data.transpose(1, 2).contiguous(), target,
                             _n_points=data.shape[1], **kw)


def cw_family_attack(model, adv_func, clip_func, data, target, y_truth=None, ae_model=None, spectral=False,
                     targeted=False, fresh=True, final_clip=True, attack_lr=1e-2, binary_step=2, num_iter=200,
                     GAMMA=0.5, low_pass=100, trace=None):
    """CPU restatement of the four un-weighted CW variants (host-side numpy bookkeeping as in the reference):
      CW/AdvPC.py::CWAdvPC.attack    (:40-180)  ae_model, targeted, fresh
      CW/UAdvPC.py::CWUAdvPC.attack  (:40-167)  ae_model
      CW/TAOF.py::CWTAOF.attack      (:83-242)  spectral, targeted, fresh, no final clip
      CW/UAEAOF.py::CWUAEAOF.attack  (:85-241)  spectral, ae_model
    Order of the backward calls as in the reference: full cloud, auto-encoder view, low-frequency view."""
    B, N = data.shape[:2]
    ori = data.float().detach().transpose(1, 2).contiguous().clone()
    target = target.long().detach()
    label = target.numpy()
    y = y_truth.long().numpy() if y_truth is not None else None
    o_bestdist, o_bestscore = np.array([1e10] * B), np.array([-1] * B)
    o_bestattack = np.zeros((B, 3, N))
    input_val = None
    if spectral and ae_model is not None:
        w_full, w_lfc, w_ae = 1 - 2 * GAMMA, GAMMA, GAMMA
    elif spectral:
        w_full, w_lfc, w_ae = 1 - GAMMA, GAMMA, 0.
    else:
        w_full, w_lfc, w_ae = 1 - GAMMA, 0., GAMMA

    def logits_of(x):
        out = model(x)
        return out[0] if isinstance(out, tuple) else out

    for step in range(binary_step):
        adv = ori.clone().detach() + torch.randn((B, 3, N)) * 1e-7
        if spectral:
            _, V = laplace_eig(adv)
            projs = torch.bmm(adv, V)
            hfc = torch.bmm(projs[..., low_pass:], V[..., low_pass:].transpose(2, 1)).detach().clone()
            var = torch.bmm(projs[..., :low_pass], V[..., :low_pass].transpose(2, 1)).detach().clone().requires_grad_()
        else:
            var, hfc = adv.requires_grad_(), None
        opt = torch.optim.Adam([var], lr=attack_lr, weight_decay=0.)
        for it in range(num_iter):
            full = var + hfc if spectral else var
            logits = logits_of(full)
            opt.zero_grad()
            (w_full * adv_func(logits, target).mean()).backward()
            ae_logits = lfc_logits = None
            if ae_model is not None:
                ae_logits = logits_of(ae_model(full))
                (w_ae * adv_func(ae_logits, target).mean()).backward()
            if spectral:
                lfc_logits = logits_of(var)
                (w_lfc * adv_func(lfc_logits, target).mean()).backward()
            opt.step()
            with torch.no_grad():
                adv = clip_func((var + hfc if spectral else var).detach().clone(), ori)
                if spectral:
                    coeff = torch.bmm(adv, V)
                    hfc.data = torch.bmm(coeff[..., low_pass:], V[..., low_pass:].transpose(2, 1))
                    var.data = torch.bmm(coeff[..., :low_pass], V[..., :low_pass].transpose(2, 1))
                else:
                    var.data = adv
                if fresh:
                    pred = torch.argmax(logits_of(adv), dim=1).numpy()
                    lfc_pred = torch.argmax(logits_of(var), dim=1).numpy() if spectral else None
                    ae_pred = torch.argmax(logits_of(ae_model(adv)), dim=1).numpy() if ae_model is not None else None
                else:
                    pred = torch.argmax(logits, dim=1).numpy()
                    lfc_pred = torch.argmax(lfc_logits, dim=1).numpy() if spectral else None
                    ae_pred = torch.argmax(ae_logits, dim=1).numpy() if ae_model is not None else None
            dist_val = torch.sqrt(torch.sum((adv - ori) ** 2, dim=[1, 2])).numpy()
            input_val = adv.detach().numpy().copy()
            for e in range(B):
                other = lfc_pred if spectral else ae_pred
                if targeted:
                    ok = pred[e] == label[e] and other[e] != y[e]
                elif spectral and ae_model is not None:
                    ok = pred[e] != label[e] and lfc_pred[e] != label[e] and ae_pred[e] != label[e]
                else:
                    ok = pred[e] != label[e] and (other[e] != label[e] or GAMMA < 0.001)
                if dist_val[e] < o_bestdist[e] and ok:
                    o_bestdist[e], o_bestscore[e] = dist_val[e], pred[e]
                    o_bestattack[e] = input_val[e]
            if trace is not None:
                trace.append(dict(step=step, it=it, adv=input_val.copy()))
    fail = o_bestscore < 0
    o_bestattack[fail] = input_val[fail]
    adv_pc = torch.tensor(o_bestattack).to(ori)
    if final_clip:
        adv_pc = clip_func(adv_pc, ori)
    preds = torch.argmax(logits_of(adv_pc), dim=-1)
    success = ((preds == target) if targeted else (preds != target)).sum().item()
    return o_bestdist, adv_pc.detach().numpy().transpose((0, 2, 1)), success


def critical_points(model, pc, label, num):
    """CW/Add.py::get_critical_points (:14-43): the ``num`` points with the largest squared input gradient of the CE."""
    x = pc.clone().detach().float().requires_grad_()
    out = model(x)
    logits = out[0] if isinstance(out, tuple) else out
    torch.nn.functional.cross_entropy(logits, label.long()).backward()
    _, idx = torch.sum(x.grad.data ** 2, dim=1).topk(k=num, dim=-1)
    return torch.stack([pc[i, :, idx[i]] for i in range(label.shape[0])], dim=0).clone().detach()


def cw_add_attack(model, adv_func, dist_func, data, target, init_points, attack_lr=1e-2, init_weight=5e3,
                  max_weight=4e4, binary_step=10, num_iter=500):
    """CPU restatement of CW/Add.py::CWAdd.attack (:79-220) and, with ``init_points`` = the flattened cluster
    initialisation, of CW/Add_Cluster.py::CWAddClusters.attack (:132-278).  dist_func(adv[B,n,3], ori[B,K,3], weights=,
    batch_avg=)."""
    B, N = data.shape[:2]
    ori = data.float().detach().transpose(1, 2).contiguous()
    target = target.long().detach()
    label = target.numpy()
    lower, upper = np.zeros((B,)), np.ones((B,)) * max_weight
    weight = np.ones((B,)) * init_weight
    o_bestdist, o_bestscore = np.array([1e10] * B), np.array([-1] * B)
    n_add = init_points.shape[2]
    o_bestattack = np.zeros((B, 3, n_add))
    input_val = None
    for step in range(binary_step):
        adv = (init_points + torch.randn((B, 3, n_add)) * 1e-7).requires_grad_()
        bestdist, bestscore = np.array([1e10] * B), np.array([-1] * B)
        opt = torch.optim.Adam([adv], lr=attack_lr, weight_decay=0.)
        for it in range(num_iter):
            out = model(torch.cat([ori, adv], dim=-1))
            logits = out[0] if isinstance(out, tuple) else out
            pred = torch.argmax(logits, dim=-1).numpy()
            dist_val = dist_func(adv.transpose(1, 2).contiguous(), ori.transpose(1, 2).contiguous(),
                                 batch_avg=False).detach().numpy()
            input_val = adv.detach().numpy().copy()
            for e in range(B):
                if pred[e] == label[e]:
                    if dist_val[e] < bestdist[e]:
                        bestdist[e], bestscore[e] = dist_val[e], pred[e]
                    if dist_val[e] < o_bestdist[e]:
                        o_bestdist[e], o_bestscore[e] = dist_val[e], pred[e]
                        o_bestattack[e] = input_val[e]
            adv_loss = adv_func(logits, target).mean()
            dist_loss = dist_func(adv.transpose(1, 2).contiguous(), ori.transpose(1, 2).contiguous(),
                                  weights=torch.from_numpy(weight)).mean()
            opt.zero_grad()
            (adv_loss + dist_loss).backward()
            opt.step()
        for e in range(B):
            if bestscore[e] == label[e] and bestscore[e] != -1 and bestdist[e] <= o_bestdist[e]:
                lower[e] = max(lower[e], weight[e])
            else:
                upper[e] = min(upper[e], weight[e])
            weight[e] = (lower[e] + upper[e]) / 2.
    fail = lower == 0.
    o_bestattack[fail] = input_val[fail]
    out = np.concatenate([ori.numpy(), o_bestattack], axis=-1)
    return o_bestdist, out.transpose((0, 2, 1)), int((lower > 0.).sum())



def cw_add_objects_attack(model, adv_func, dist_func, data, target, object_pc, centers, attack_lr=1e-2, init_weight=5.,
                          max_weight=40., binary_step=5, num_iter=500):
    """CPU restatement of CW/Add_Objects.py::CWAddObjects.attack (:187-367) given the processed ``object_pc``
    [num_add,obj_num_p,3] (ctor :88-92) and the cluster ``centers`` [B,num_add,3] (``_init_centers`` :100-146).
    dist_func(adv[B,n,3], ori[B,K,3], adv_obj, ori_obj, weights=, batch_avg=)."""
    B, N = data.shape[:2]
    num_add, obj_num_p = object_pc.shape[:2]
    ori = data.float().detach().transpose(1, 2).contiguous()
    target = target.long().detach()
    label = target.numpy()
    lower, upper = np.zeros((B,)), np.ones((B,)) * max_weight
    weight = np.ones((B,)) * init_weight
    o_bestdist, o_bestscore = np.array([1e10] * B), np.array([-1] * B)
    n_add = num_add * obj_num_p
    o_bestattack = np.zeros((B, 3, n_add))
    shifts = torch.from_numpy(np.asarray(centers)).float()
    objects = torch.from_numpy(np.tile(object_pc, (B, 1, 1, 1))).float()
    input_val = None

    def rotate_shift(points, angles, sh):  # :148-185, rotation about y by angles[...,0]
        c, s_ = torch.cos(angles[..., 0]), torch.sin(angles[..., 0])
        z, o = torch.zeros_like(c), torch.ones_like(c)
        rot = torch.stack([c, z, s_, z, o, z, -s_, z, c], dim=-1).view(B * num_add, 3, 3)
        return torch.bmm(points.view(B * num_add, obj_num_p, 3), rot).view(B, num_add, obj_num_p, 3) + sh[:, :, None, :]

    for step in range(binary_step):
        adv_objects = (objects + torch.randn((B, num_add, obj_num_p, 3)) * 1e-7).requires_grad_()
        adv_shifts = (shifts + torch.randn((B, num_add, 3)) * 1e-7).requires_grad_()
        adv_angles = (torch.rand((B, num_add, 3)) * np.pi).requires_grad_()
        bestdist, bestscore = np.array([1e10] * B), np.array([-1] * B)
        opt = torch.optim.Adam([adv_objects, adv_shifts, adv_angles], lr=attack_lr, weight_decay=0.)
        for it in range(num_iter):
            adv = rotate_shift(adv_objects, adv_angles, adv_shifts).view(B, n_add, 3).transpose(1, 2).contiguous()
            out = model(torch.cat([ori, adv], dim=-1))
            logits = out[0] if isinstance(out, tuple) else out
            pred = torch.argmax(logits, dim=-1).numpy()
            adv_t, ori_t = adv.transpose(1, 2).contiguous(), ori.transpose(1, 2).contiguous()
            dist_val = dist_func(adv_t, ori_t, adv_objects, objects, batch_avg=False).detach().numpy()
            input_val = adv.detach().numpy().copy()
            for e in range(B):
                if pred[e] == label[e]:
                    if dist_val[e] < bestdist[e]:
                        bestdist[e], bestscore[e] = dist_val[e], pred[e]
                    if dist_val[e] < o_bestdist[e]:
                        o_bestdist[e], o_bestscore[e] = dist_val[e], pred[e]
                        o_bestattack[e] = input_val[e]
            adv_loss = adv_func(logits, target).mean()
            dist_loss = dist_func(adv_t, ori_t, adv_objects, objects, weights=torch.from_numpy(weight)).mean()
            opt.zero_grad()
            (adv_loss + dist_loss).backward()
            opt.step()
            with torch.no_grad():
                adv_angles.data = adv_angles.data % (2. * np.pi)
        for e in range(B):
            if bestscore[e] == label[e] and bestscore[e] != -1 and bestdist[e] <= o_bestdist[e]:
                lower[e] = max(lower[e], weight[e])
            else:
                upper[e] = min(upper[e], weight[e])
            weight[e] = (lower[e] + upper[e]) / 2.
    fail = lower == 0.
    o_bestattack[fail] = input_val[fail]
    out = np.concatenate([ori.numpy(), o_bestattack], axis=-1)
    return o_bestdist, out.transpose((0, 2, 1)), int((lower > 0.).sum())

# --------------------------------------------------------------------------
# remaining distance operators               util/dist_utils.py:178-229, 297-409, 498-561
# --------------------------------------------------------------------------


def laplacian_knn_indices(x, k):
    """LaplacianDist.KNN_indices (:217-229): float64 Gram distances, top-(k+1), self dropped."""
    pc = x.clone().detach().double()
    inner = -2. * torch.matmul(pc.transpose(2, 1), pc)
    xx = torch.sum(pc ** 2, dim=1, keepdim=True)
    dist = xx + inner + xx.transpose(2, 1)
    neg_value, idx = (-dist).topk(k=k + 1, dim=-1)
    return -(neg_value[..., 1:]), idx[..., 1:]


def laplacian_dist(adv_pc, ori_pc, nearest_indices, weights=None, batch_avg=True):
    """LaplacianDist.forward (:186-215)."""
    delta = adv_pc - ori_pc
    delta = delta.unsqueeze(3).expand(-1, -1, -1, nearest_indices.shape[2])
    nbr = torch.gather(delta, 2, nearest_indices.unsqueeze(1).expand(-1, 3, -1, -1))
    return _weighted(torch.sum(torch.norm(nbr, dim=1) ** 2, dim=[1, 2]), weights, batch_avg)


def farthest_dist(adv_pc, weights=None, batch_avg=True):
    """FarthestDist.forward (:304-325); adv_pc [B,num_add,cl_num_p,3]."""
    delta = adv_pc[:, :, None, :, :] - adv_pc[:, :, :, None, :] + 1e-7
    norm = torch.norm(delta, p=2, dim=-1)
    far = torch.sum(torch.max(torch.max(norm, dim=2)[0], dim=2)[0], dim=1)
    return _weighted(far, weights, batch_avg)


def far_chamfer_dist(adv_pc, ori_pc, num_add, weights=None, batch_avg=True, method='adv2ori', chamfer_weight=0.1):
    """FarChamferDist.forward (:347-365)."""
    B = adv_pc.shape[0]
    return (farthest_dist(adv_pc.view(B, num_add, -1, 3), weights, batch_avg) +
            chamfer_dist(adv_pc, ori_pc, weights, batch_avg, method) * chamfer_weight)


def l2_chamfer_dist(adv_pc, ori_pc, adv_obj, ori_obj, weights=None, batch_avg=True, method='adv2ori',
                    chamfer_weight=0.2):
    """L2ChamferDist.forward (:387-409)."""
    B = adv_pc.shape[0]
    return (l2_dist(adv_obj.view(B, -1, 3), ori_obj.view(B, -1, 3), weights, batch_avg) +
            chamfer_weight * chamfer_dist(adv_pc, ori_pc, weights, batch_avg, method))


def curv_dist(ori_data, adv_data, ori_normal, curv_loss_knn=2):
    """CurvDist.forward (:503-508) with the canonical kNN (direct-form distances, ties -> lower index)."""
    ori_kappa = kappa_ori(ori_data, ori_normal, 2)[0]
    adv_pts, ori_pts = adv_data.permute(0, 2, 1), ori_data.permute(0, 2, 1)
    nn_idx = knn_points(adv_pts, ori_pts, 1)[1]
    normal = knn_gather(ori_normal.permute(0, 2, 1), nn_idx).permute(0, 3, 1, 2).squeeze(3).contiguous()
    adv_kappa = kappa_ori(adv_data, normal, curv_loss_knn)[0]
    return ((adv_kappa - torch.gather(ori_kappa, 1, nn_idx.squeeze(-1))) ** 2).mean(-1).mean()

# --------------------------------------------------------------------------
# eval_ASR metric phase                      util/other_utils.py:15-101
# --------------------------------------------------------------------------


def asr_counts(logits, adv_logits, label):
    """(at_num, at_denom) increments of util/other_utils.py:83-88."""
    ok_ori = torch.argmax(logits, dim=-1) == label
    ok_adv = torch.argmax(adv_logits, dim=-1) == label
    denom = ok_ori.sum().float().item()
    return denom - (ok_ori * ok_adv).sum().float().item(), denom


def uniform_loss(adv_pc, natives, percentages=(0.004, 0.006, 0.008, 0.010, 0.012), radius=1.0, k=2):
    """FGM/GeoA3_args.py:258-302, with the CUDA-extension calls served by ``natives``
    (the C restatement loaded by ``oracle/c_oracle.py``)."""
    if adv_pc.size(1) == 3:
        adv_pc = adv_pc.permute(0, 2, 1).contiguous()
    b, n, _ = adv_pc.shape
    npoint = int(n * 0.05)
    total = None
    flipped = adv_pc.transpose(1, 2).contiguous()
    for p in percentages:
        p = p * 4
        nsample = int(n * p)
        r = math.sqrt(p * radius)
        expect = torch.sqrt(torch.Tensor([math.pi * (radius ** 2) * p / nsample]))
        fidx = natives.furthest_point_sampling(adv_pc, npoint)
        new_xyz = natives.gather_points(flipped, fidx).transpose(1, 2).contiguous()
        idx = natives.ball_query(new_xyz, adv_pc, r, nsample)
        grouped = natives.group_points(flipped, idx).permute(0, 2, 3, 1).contiguous()
        grouped = torch.cat(torch.unbind(grouped, axis=1), axis=0)  # [b*npoint, nsample, 3]
        d, _ = knn_points(grouped, grouped, k + 1)
        u = torch.sqrt(torch.abs(d[:, :, 1:].contiguous()) + 1e-12).mean(axis=[-1])
        u = ((u - expect) ** 2 / (expect + 1e-12)).reshape(-1)
        m = u.mean() * math.pow(p * 100, 2)
        total = m if total is None else total + m
    return total / len(percentages)
