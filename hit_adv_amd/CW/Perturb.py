"""CW point-perturbation attack (CVPR'19 "Generating 3D Adversarial Point Clouds"), interface of the
reference's CW/Perturb.py::CWPerturb (ctor :16-44, attack :46-202): binary search over a per-sample distance
weight, Adam on the xyz of every point, best-result tracking.  Success means ``pred == target`` (the class
is written for targeted attacks, :107,134,141,178).

Device-resident restatement: the per-iteration ``.cpu().numpy()`` copies and the Python loop over samples
(:127-145) become a few [B]-sized tensor ops; the bisection keeps the reference's float64 bounds.
"""
import torch
import torch.optim as optim

from ._victim import Victim

class CWPerturb:
    """Class for CW attack."""

    def __init__(self, model, adv_func, dist_func, attack_lr=1e-2, init_weight=10., max_weight=80.,
                 binary_step=10, num_iter=500, pre_head=None, clip_func=None, verbose=True, fast_victim=True):
        self.model = model.cuda()
        self.model.eval()
        self._victim = Victim(self.model, fast_victim)
        self.adv_func = adv_func
        self.dist_func = dist_func
        self.attack_lr = attack_lr
        self.init_weight = init_weight
        self.max_weight = max_weight
        self.binary_step = binary_step
        self.num_iter = num_iter
        self.clip_func = clip_func
        self.pre_head = pre_head
        self.verbose = verbose

    def _logits(self, x):
        return self._victim(self.pre_head(x) if self.pre_head is not None else x)

    def attack(self, data, target, _channel_first=False):
        """data [B,num_points,3 or 6] (or channel-first [B,3|6,num_points>6]), target [B]
        -> (float64 ndarray [B,num_points,3], number of samples with a successful step)."""
        self._victim.prepare()
        B, K = data.shape[:2]
        data = data.float().cuda().detach()
        if _channel_first:  # CWPerturbT hands over [B,3,num_points] (PerturbT.py:53)
            K = data.shape[2]
        else:
            if data.shape[1] > 6:
                data = data.transpose(1, 2).contiguous()
            if data.shape[1] == 6:
                data = data[:, :3, :]
        ori = data.clone().detach().contiguous()
        target = target.long().cuda().detach()
        dev = ori.device
        f64 = dict(device=dev, dtype=torch.float64)
        lower = torch.zeros(B, **f64)
        upper = torch.full((B,), float(self.max_weight), **f64)
        weight = torch.full((B,), float(self.init_weight), **f64)
        o_bestdist = torch.full((B,), 1e10, **f64)
        o_bestscore = torch.full((B,), -1, device=dev, dtype=torch.int64)
        o_bestattack = torch.zeros(B, 3, K, device=dev)
        report_every = max(1, self.num_iter // 5)
        last_input = ori
        for binary_step in range(self.binary_step):
            adv = (ori.clone() + torch.randn((B, 3, K)).cuda() * 1e-7).requires_grad_()
            bestdist = torch.full((B,), 1e10, **f64)
            bestscore = torch.full((B,), -1, device=dev, dtype=torch.int64)
            opt = optim.Adam([adv], lr=self.attack_lr, weight_decay=0.)
            adv_loss = torch.zeros((), device=dev)
            dist_loss = torch.zeros((), device=dev)
            for iteration in range(self.num_iter):
                logits = self._logits(adv)
                pred = logits.argmax(dim=1)
                if self.verbose and iteration % report_every == 0:
                    print('Step {}, iteration {}, success {}/{}\nadv_loss: {:.4f}, dist_loss: {:.4f}'.format(
                        binary_step, iteration, (pred == target).sum().item(), B, adv_loss.item(), dist_loss.item()))
                with torch.no_grad():
                    last_input = adv.detach().clone()  # what the reference calls input_val (:125)
                    dist_val = torch.sqrt(torch.sum((adv - ori) ** 2, dim=[1, 2])).double()
                    hit = pred == target
                    better = hit & (dist_val < bestdist)
                    bestdist = torch.where(better, dist_val, bestdist)
                    bestscore = torch.where(better, pred, bestscore)
                    o_better = hit & (dist_val < o_bestdist)
                    o_bestdist = torch.where(o_better, dist_val, o_bestdist)
                    o_bestscore = torch.where(o_better, pred, o_bestscore)
                    o_bestattack = torch.where(o_better[:, None, None], adv.detach(), o_bestattack)
                adv_loss = self.adv_func(logits, target).mean()
                dist_loss = self.dist_func(adv, ori, weight).mean()
                opt.zero_grad()
                (adv_loss + dist_loss).backward()
                opt.step()
                if self.clip_func is not None:
                    adv.data = self.clip_func(adv.clone().detach(), ori)
            with torch.no_grad():  # bisection, :172-184
                ok = (bestscore == target) & (bestscore != -1) & (bestdist <= o_bestdist)
                lower = torch.where(ok, torch.maximum(lower, weight), lower)
                upper = torch.where(ok, upper, torch.minimum(upper, weight))
                weight = (lower + upper) / 2.
        with torch.no_grad():
            fail = lower == 0.
            # failures get input_val, i.e. the iterate that ENTERED the last iteration (:189-193)
            best = torch.where(fail[:, None, None], last_input, o_bestattack)
        success_num = int((lower > 0.).sum().item())
        if self.verbose:
            print('Successfully attack {}/{}'.format(success_num, B))
        return best.double().cpu().numpy().transpose((0, 2, 1)), success_num
