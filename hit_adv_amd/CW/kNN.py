"""kNN attack (AAAI'20 "Robust Adversarial Objects"), interface of the reference's CW/kNN.py::CWKNN.

Every iteration runs the victim forward/backward on PyTorch-ROCm plus ``dist_func`` -- normally
ChamferkNNDist, i.e. the fused Chamfer NN-min kernel and the top-(k+1) kNN kernel with their HIP
backwards -- then ``clip_func``.  Success is ``pred == target`` (targeted attack, CW/kNN.py:86,146).
"""
import torch
import torch.optim as optim

from ._victim import Victim

class CWKNN:
    """Class for CW attack (constructor of CW/kNN.py:18-38)."""

    def __init__(self, model, adv_func, dist_func, clip_func, attack_lr=1e-3, num_iter=2500, verbose=True,
                 fast_victim=True):
        self.model = model.cuda()
        self.model.eval()
        self._victim = Victim(self.model, fast_victim)
        self.adv_func = adv_func
        self.dist_func = dist_func
        self.clip_func = clip_func
        self.attack_lr = attack_lr
        self.num_iter = num_iter
        self.verbose = verbose

    def _logits(self, x):
        return self._victim(x)

    def _clip(self, adv, ori, normal):
        return self.clip_func(adv, ori)

    def attack(self, data, target):
        """data [B,num_points,3 or 6], target [B] -> (float32 ndarray [B,num_points,3], success count)."""
        self._victim.prepare()
        B, K = data.shape[:2]
        pc = data.float().cuda().detach().transpose(1, 2).contiguous()
        normal = None if pc.shape[1] == 3 else pc[:, 3:, :]
        ori = pc[:, :3, :].contiguous().clone().detach()
        target = target.long().cuda().detach()
        # the reference draws the jitter on the CPU generator and moves it over (:64-65)
        adv = (ori.clone() + torch.randn((B, 3, K)).cuda() * 1e-7).requires_grad_()
        opt = optim.Adam([adv], lr=self.attack_lr, weight_decay=0.)
        ori_pts = ori.transpose(1, 2).contiguous()
        adv_loss = torch.zeros((), device=ori.device)
        dist_loss = torch.zeros((), device=ori.device)
        report_every = max(1, self.num_iter // 5)
        for iteration in range(self.num_iter):
            logits = self._logits(adv)
            if self.verbose and iteration % report_every == 0:
                hit = (logits.argmax(dim=1) == target).sum().item()
                print('Iteration {}/{}, success {}/{}\nadv_loss: {:.4f}, dist_loss: {:.4f}'.format(
                    iteration, self.num_iter, hit, B, adv_loss.item(), dist_loss.item()))
            adv_loss = self.adv_func(logits, target).mean()
            dist_loss = self.dist_func(adv.transpose(1, 2).contiguous(), ori_pts).mean() * K
            opt.zero_grad()
            (adv_loss + dist_loss).backward()
            opt.step()
            if self.clip_func is not None:
                adv.data = self._clip(adv.clone().detach(), ori, normal)
        with torch.no_grad():
            success_num = (self._logits(adv).argmax(dim=-1) == target).sum().item()
        if self.verbose:
            print('Successfully attack {}/{}'.format(success_num, B))
        return adv.transpose(1, 2).contiguous().detach().cpu().numpy(), success_num
