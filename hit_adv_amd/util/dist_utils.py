"""Distance operators with the reference's nn.Module protocol
``dist(adv_pc, ori_pc, weights=None, batch_avg=True)`` (util/dist_utils.py:15-175, 258-294, 464-495).
The set reductions and the kNN search run in HIP kernels; what is left in torch is O(B*N) glue.
"""
import torch
import torch.nn as nn

from .. import ops
from ..pytorch3d_ops import knn_gather, knn_points
from .set_distance import ChamferDistance, HausdorffDistance


def _apply_weights(loss, weights, batch_avg):
    if weights is not None:  # None = all ones (the reference multiplies by torch.ones, dist_utils.py:33-35): x * 1 == x,
        loss = loss * weights.float().to(loss.device)  # and leaving it out keeps a host->device copy out of the loop
    return loss.mean() if batch_avg else loss


def _select(method, forward_term, backward_term):
    if method == 'adv2ori':
        return forward_term
    if method == 'ori2adv':
        return backward_term
    return (forward_term + backward_term) / 2.


class L2Dist(nn.Module):
    """util/dist_utils.py:15-41."""

    def forward(self, adv_pc, ori_pc, weights=None, batch_avg=True):
        dist = torch.sqrt(torch.sum((adv_pc - ori_pc) ** 2, dim=[1, 2]) + 1e-7)
        return _apply_weights(dist, weights, batch_avg)


class ChamferDist(nn.Module):
    """util/dist_utils.py:44-80."""

    def __init__(self, method='adv2ori', reference_arithmetic=None):
        super().__init__()
        self.method = method
        self.chamfer = ChamferDistance(reference_arithmetic)

    def forward(self, adv_pc, ori_pc, weights=None, batch_avg=True):
        fwd, bwd = self.chamfer(adv_pc, ori_pc)
        return _apply_weights(_select(self.method, fwd, bwd), weights, batch_avg)


class HausdorffDist(nn.Module):
    """util/dist_utils.py:83-119."""

    def __init__(self, method='adv2ori', reference_arithmetic=None):
        super().__init__()
        self.method = method
        self.hausdorff = HausdorffDistance(reference_arithmetic)

    def forward(self, adv_pc, ori_pc, weights=None, batch_avg=True):
        fwd, bwd = self.hausdorff(adv_pc, ori_pc)
        return _apply_weights(_select(self.method, fwd, bwd), weights, batch_avg)


class KNNDist(nn.Module):
    """kNN-distance penalty of the AAAI'20 attack, util/dist_utils.py:122-175.

    The reference builds the full Gram matrix and takes topk(k+1); here one HIP kernel returns the
    k+1 smallest squared distances per point directly (rank 0 is the point itself and is dropped,
    as in :157-158)."""

    def __init__(self, k=5, alpha=1.05, reference_arithmetic=None):
        super().__init__()
        self.k = k
        self.alpha = alpha
        self.reference_arithmetic = reference_arithmetic  # True: the Gram matrix of :148-150, bit for bit

    def forward(self, pc, weights=None, batch_avg=True):
        if pc.shape[1] == 3:  # [B,3,K] -> [B,K,3]; a [B,K,3] input passes through (:146-147)
            pc = pc.transpose(2, 1)
        pc = pc.contiguous()
        dists, _ = ops.KnnPoints.apply(pc, pc, self.k + 1, ops.knn_dist_matrix_form(self.reference_arithmetic))
        value = dists[..., 1:].mean(dim=-1)  # [B,K]
        with torch.no_grad():
            threshold = value.mean(dim=-1) + self.alpha * value.std(dim=-1)
            mask = (value > threshold[:, None]).float()
        return _apply_weights((value * mask).mean(dim=1), weights, batch_avg)


class ChamferkNNDist(nn.Module):
    """util/dist_utils.py:258-294."""

    def __init__(self, chamfer_method='adv2ori', knn_k=5, knn_alpha=1.05, chamfer_weight=5., knn_weight=3.,
                 reference_arithmetic=None):
        super().__init__()
        self.chamfer_dist = ChamferDist(method=chamfer_method, reference_arithmetic=reference_arithmetic)
        self.knn_dist = KNNDist(k=knn_k, alpha=knn_alpha, reference_arithmetic=reference_arithmetic)
        self.w1 = chamfer_weight
        self.w2 = knn_weight

    def forward(self, adv_pc, ori_pc, weights=None, batch_avg=True):
        return (self.chamfer_dist(adv_pc, ori_pc, weights=weights, batch_avg=batch_avg) * self.w1 +
                self.knn_dist(adv_pc, weights=weights, batch_avg=batch_avg) * self.w2)


def curvature_proxy(pc, normal, k):
    """kappa[b,n] = mean_k |unit(x_nbr - x_n) . normal_n| over the k nearest neighbours and the kNN
    index table it used.  pc, normal: [B,3,N].  (ShapeAttack/HiT_ADV.py:318-325, dist_utils.py:476-486)"""
    pts = pc.permute(0, 2, 1).contiguous()
    idx = knn_points(pts, pts, K=k + 1).idx
    nbr = knn_gather(pts, idx)[:, :, 1:, :]  # [B,N,k,3]
    vec = nbr - pts.unsqueeze(2)
    vec = vec / vec.norm(2, dim=3, keepdim=True).clamp(min=1e-12)
    kappa = (vec * normal.permute(0, 2, 1).unsqueeze(2)).sum(3).abs().mean(2)
    return kappa, idx


def curvature_std(pc, normal, k):
    """Unbiased std of the neighbours' curvature proxy, [B,N] (HiT_ADV.py:327-339)."""
    kappa, idx = curvature_proxy(pc, normal, k)
    nbr_kappa = knn_gather(kappa.unsqueeze(2), idx)[:, :, 1:, 0]  # [B,N,k]
    return nbr_kappa.std(dim=2), kappa, idx


class CurvStdDist(nn.Module):
    """util/dist_utils.py:464-495 (an eval_ASR metric, util/other_utils.py:39,75)."""

    def __init__(self, k=5):
        super().__init__()
        self.k = k

    def forward(self, ori_data, adv_data, ori_normal):
        a = curvature_std(ori_data, ori_normal, self.k)[0]
        b = curvature_std(adv_data, ori_normal, self.k)[0]
        return torch.nn.PairwiseDistance(p=2)(a, b).mean()


class LaplacianDist(nn.Module):
    """util/dist_utils.py:178-229: sum over every point's k neighbours of |delta_neighbour|^2 (delta = adv - ori,
    [B,3,K]; ``nearest_indices`` [B,K,k] from ``KNN_indices``).  The neighbour search runs on the HIP kNN kernel
    (fp32 direct form) where the reference builds a float64 Gram matrix and a full top-k."""

    def __init__(self, k):
        super().__init__()
        self.k = k

    def forward(self, adv_pc, ori_pc, nearest_indices, weights=None, batch_avg=True):
        delta = adv_pc - ori_pc  # [B,3,K]
        B, _, K = delta.shape
        k = nearest_indices.shape[2]
        nbr = torch.gather(delta, 2, nearest_indices.reshape(B, 1, K * k).expand(-1, 3, -1))  # [B,3,K*k]
        dist = torch.sum(torch.norm(nbr.view(B, 3, K, k), dim=1) ** 2, dim=[1, 2])
        return _apply_weights(dist, weights, batch_avg)

    def KNN_indices(self, x):
        """x [B,3,K] -> (squared distances [B,K,k] float64, indices [B,K,k]) of the k nearest other points."""
        pts = x.clone().detach().float().transpose(2, 1).contiguous()
        nn_ = knn_points(pts, pts, K=self.k + 1)
        return nn_.dists[..., 1:].double(), nn_.idx[..., 1:]


class FarthestDist(nn.Module):
    """util/dist_utils.py:297-325: per added cluster the largest pairwise point distance, summed over clusters.
    adv_pc [B,num_add,cl_num_p,3]."""

    def forward(self, adv_pc, weights=None, batch_avg=True):
        delta = adv_pc[:, :, None, :, :] - adv_pc[:, :, :, None, :] + 1e-7
        norm = torch.norm(delta, p=2, dim=-1)  # [B,na,np,np]
        far = norm.max(dim=2)[0].max(dim=2)[0].sum(dim=1)
        return _apply_weights(far, weights, batch_avg)


class FarChamferDist(nn.Module):
    """util/dist_utils.py:328-365 (constraint of the adding-clusters attack)."""

    def __init__(self, num_add, chamfer_method='adv2ori', chamfer_weight=0.1):
        super().__init__()
        self.num_add = num_add
        self.far_dist = FarthestDist()
        self.chamfer_dist = ChamferDist(method=chamfer_method)
        self.cd_w = chamfer_weight

    def forward(self, adv_pc, ori_pc, weights=None, batch_avg=True):
        B = adv_pc.shape[0]
        chamfer_loss = self.chamfer_dist(adv_pc, ori_pc, weights=weights, batch_avg=batch_avg)
        far_loss = self.far_dist(adv_pc.view(B, self.num_add, -1, 3), weights=weights, batch_avg=batch_avg)
        return far_loss + chamfer_loss * self.cd_w


class L2ChamferDist(nn.Module):
    """util/dist_utils.py:368-409 (constraint of the adding-objects attack)."""

    def __init__(self, num_add, chamfer_method='adv2ori', chamfer_weight=0.2):
        super().__init__()
        self.num_add = num_add
        self.chamfer_dist = ChamferDist(method=chamfer_method)
        self.cd_w = chamfer_weight
        self.l2_dist = L2Dist()

    def forward(self, adv_pc, ori_pc, adv_obj, ori_obj, weights=None, batch_avg=True):
        B = adv_pc.shape[0]
        chamfer_loss = self.chamfer_dist(adv_pc, ori_pc, weights=weights, batch_avg=batch_avg)
        l2_loss = self.l2_dist(adv_obj.view(B, -1, 3), ori_obj.view(B, -1, 3), weights=weights, batch_avg=batch_avg)
        return l2_loss + self.cd_w * chamfer_loss


class CurvDist(nn.Module):
    """util/dist_utils.py:498-561 (GeoA3's curvature consistency): mean_n (kappa_adv(n) - kappa_ori(nn_ori(n)))^2, the
    adversarial curvature measured against the normal of the nearest original point.  All three neighbour searches
    are HIP kNN launches."""

    def __init__(self, curv_loss_knn=2):
        super().__init__()
        self.curv_loss_knn = curv_loss_knn

    @staticmethod
    def _kappa(pc, normal, k):
        pts = pc.permute(0, 2, 1).contiguous()
        nbr = knn_gather(pts, knn_points(pts, pts, K=k + 1).idx)[:, :, 1:, :]  # [B,N,k,3]
        vec = nbr - pts.unsqueeze(2)
        vec = vec / vec.norm(2, dim=3, keepdim=True).clamp(min=1e-12)
        return (vec * normal.permute(0, 2, 1).unsqueeze(2)).sum(3).abs().mean(2)

    def forward(self, ori_data, adv_data, ori_normal):
        ori_kappa = self._kappa(ori_data, ori_normal, 2)  # the reference's _get_kappa_ori default k=2 (:506,510)
        adv_pts = adv_data.permute(0, 2, 1).contiguous()
        nn_idx = knn_points(adv_pts, ori_data.permute(0, 2, 1).contiguous(), K=1).idx  # [B,N,1]
        normal = knn_gather(ori_normal.permute(0, 2, 1).contiguous(), nn_idx).squeeze(2).permute(0, 2, 1)
        adv_kappa = self._kappa(adv_data, normal, self.curv_loss_knn)
        return ((adv_kappa - torch.gather(ori_kappa, 1, nn_idx.squeeze(-1))) ** 2).mean(-1).mean()
