"""AOF attack ("Boosting 3D adversarial attacks with attacking on frequency"), interface of the reference's
CW/AOF.py (``knn`` :12-27, ``get_Laplace_from_pc`` :30-51, ``CWAOF`` :54-241): Adam on the low-frequency
component of the cloud in the eigenbasis of its kNN-graph Laplacian, GAMMA-weighted loss on the full cloud
and on its low-frequency part.

GPU plan: the 30-NN graph comes from ``hitadv_knn_points`` (no [B,N,N] distance matrix, no full top-k), the
dense Laplacian is assembled with one scatter, and the eigendecomposition is ``torch.linalg.eigh`` (rocSOLVER) --
the reference calls ``torch.symeig`` (:50), which no longer exists in torch >= 2.  Best-result tracking is
device-resident.
"""
import torch
import torch.optim as optim

from ..pytorch3d_ops import knn_points


def knn(x, k):
    """x [B,3,N] -> idx [B,N,k]: the k nearest points (self included), nearest first."""
    pts = x.detach().transpose(2, 1).contiguous()
    return knn_points(pts, pts, K=k).idx


@torch.no_grad()
def get_Laplace_from_pc(ori_pc, k=30):
    """ori_pc [B,3,N] -> (eigenvalues [B,N] ascending, eigenvectors [B,N,N]) of L = D - A with
    A_ij = exp(-|x_i - x_j|^2) on the symmetrised k-NN graph."""
    pts = ori_pc.detach().transpose(2, 1).contiguous()
    nn = knn_points(pts, pts, K=k)
    B, N, _ = pts.shape
    A = torch.zeros(B, N, N, device=pts.device).scatter_(2, nn.idx, torch.exp(-nn.dists))
    A = torch.maximum(A, A.transpose(2, 1))  # (i,j) kept when either point is among the other's neighbours
    L = torch.diag_embed(A.sum(dim=2)) - A
    e, v = torch.linalg.eigh(L)
    return e.to(ori_pc), v.to(ori_pc)

from ._victim import Victim

class CWAOF:
    """Class for the AOF attack (constructor of CW/AOF.py:58-81)."""

    def __init__(self, model, adv_func, dist_func, attack_lr=1e-2, binary_step=2, num_iter=200, GAMMA=0.5,
                 low_pass=100, clip_func=None, verbose=True, fast_victim=True):
        self.model = model.cuda()
        self.model.eval()
        self._victim = Victim(self.model, fast_victim)
        self.adv_func = adv_func
        self.dist_func = dist_func  # stored, unused (as in the reference)
        self.attack_lr = attack_lr
        self.binary_step = binary_step
        self.num_iter = num_iter
        self.GAMMA = GAMMA
        self.low_pass = low_pass
        self.clip_func = clip_func
        self.verbose = verbose

    def _logits(self, x):
        return self._victim(x)

    def _split(self, pc, V):
        coeff = torch.bmm(pc, V)
        lp = self.low_pass
        return (torch.bmm(coeff[..., :lp], V[..., :lp].transpose(2, 1)),
                torch.bmm(coeff[..., lp:], V[..., lp:].transpose(2, 1)))

    def attack(self, data, target):
        """data [B,num_points,3|6], target [B] (true labels; untargeted) -> (float32 ndarray [B,num_points,3], successes)."""
        self._victim.prepare()
        B, K = data.shape[:2]
        data = data.float().cuda().detach().transpose(1, 2).contiguous()
        if data.shape[1] == 6:
            data = data[:, :3, :]
        ori = data.clone().detach().contiguous()
        target = target.long().cuda().detach()
        dev = ori.device
        o_bestdist = torch.full((B,), 1e10, device=dev)
        o_bestscore = torch.full((B,), -1, device=dev, dtype=torch.int64)
        o_bestattack = torch.zeros(B, 3, K, device=dev)
        report_every = max(1, self.num_iter // 5)
        adv = ori
        for binary_step in range(self.binary_step):
            adv = ori.clone() + torch.randn((B, 3, K)).cuda() * 1e-7
            _, V = get_Laplace_from_pc(adv)
            lfc, hfc = self._split(adv, V)
            lfc = lfc.detach().clone().requires_grad_()
            hfc = hfc.detach().clone()
            opt = optim.Adam([lfc], lr=self.attack_lr, weight_decay=0.)
            for iteration in range(self.num_iter):
                adv_loss = (1 - self.GAMMA) * self.adv_func(self._logits(lfc + hfc), target).mean()
                opt.zero_grad()
                adv_loss.backward()
                lfc_adv_loss = self.GAMMA * self.adv_func(self._logits(lfc), target).mean()
                lfc_adv_loss.backward()
                opt.step()
                with torch.no_grad():
                    adv = self.clip_func((lfc + hfc).detach().clone(), ori)
                    new_l, new_h = self._split(adv, V)
                    lfc.data, hfc.data = new_l, new_h
                    pred = self._logits(adv).argmax(dim=1)
                    lfc_pred = self._logits(lfc).argmax(dim=1)
                    dist_val = torch.sqrt(torch.sum((adv - ori) ** 2, dim=[1, 2]))
                    ok = (pred != target) & ((lfc_pred != target) | (self.GAMMA < 0.001)) & (dist_val < o_bestdist)
                    o_bestdist = torch.where(ok, dist_val, o_bestdist)
                    o_bestscore = torch.where(ok, pred, o_bestscore)
                    o_bestattack = torch.where(ok[:, None, None], adv, o_bestattack)
                if self.verbose and iteration % report_every == 0:
                    n_ok = ((pred != target) & (lfc_pred != target)).sum().item()
                    print('Step {}, iteration {}, success {}/{}\nadv_loss: {:.4f}, dist_loss: {:.4f}'.format(
                        binary_step, iteration, n_ok, B, adv_loss.item() + lfc_adv_loss.item(), 0.))
        with torch.no_grad():
            best = torch.where((o_bestscore < 0)[:, None, None], adv, o_bestattack)
            adv_pc = self.clip_func(best, ori)
            success_num = (self._logits(adv_pc).argmax(dim=-1) != target).sum().item()
        if self.verbose:
            print('Successfully attack {}/{}'.format(success_num, B))
        return adv_pc.detach().cpu().numpy().transpose((0, 2, 1)), success_num
