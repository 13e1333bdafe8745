"""CW point-adding attack (CVPR'19 "Generating 3D Adversarial Point Clouds"), interface of the reference's CW/Add.py
(``get_critical_points`` :14-43, ``CWAdd`` ctor :50-77, attack :79-220): ``num_add`` new points are initialised on the
most salient points of the cloud and optimised with Adam under a Chamfer / Hausdorff constraint to the original cloud
(``dist_func`` on [B,num_add,3] vs [B,K,3]: the ragged-size case of the HIP nearest-neighbour reductions), with the
usual per-sample bisection of the constraint weight.  Targeted: success means ``pred == target``.

Best-result tracking and the bisection are device-resident (float64 bounds as in the reference); one iteration -- victim
forward / backward, both losses, Adam, best tracking -- is captured into a hipGraph and replayed (util/graph_loop.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .. import ops
from ..model._pointwise import degrade_on_fp16_range


def get_critical_points(model, pc, label, num):
    """The ``num`` points of pc [B,3,K] with the largest squared input-gradient of the cross-entropy -> [B,3,num]."""
    x = pc.clone().detach().float().cuda().requires_grad_()
    label = label.long().cuda()
    model.eval()
    logits = model(x)
    if isinstance(logits, tuple):
        logits = logits[0]
    grad, = torch.autograd.grad(F.cross_entropy(logits, label), x)
    with torch.no_grad():
        # descending, ties -> lower point index.  (The reference's torch.topk leaves the order of tied scores -- e.g. the
        # zero gradients of every non-critical point under a max-pooling victim -- to the backend.)
        score = torch.sum(grad ** 2, dim=1).contiguous()
        if num <= 64:
            _, idx = ops.topk_rows(score, num, largest=True)  # [B,num], HIP row top-k
        else:
            idx = torch.sort(score, dim=-1, descending=True, stable=True).indices[:, :num]
        return torch.gather(pc.to(idx.device), 2, idx.unsqueeze(1).expand(-1, pc.shape[1], -1)).clone().detach()

from ._victim import Victim

class CWAdd:
    """Class for CW attack (adding points)."""

    def __init__(self, model, adv_func, dist_func, attack_lr=1e-2, init_weight=5e3, max_weight=4e4, binary_step=10,
                 num_iter=500, num_add=512, verbose=True, fast_victim=True, use_graph='auto'):
        self.model = model.cuda()
        self.model.eval()
        self._victim = Victim(self.model, fast_victim)
        self.use_graph = use_graph  # 'auto': replay one captured iteration when the victim and the losses allow it
        self.last_graph_used = False
        self.adv_func = adv_func
        self.dist_func = dist_func
        self.attack_lr = attack_lr
        self.init_weight = init_weight
        self.max_weight = max_weight
        self.binary_step = binary_step
        self.num_iter = num_iter
        self.num_add = num_add
        self.verbose = verbose

    def _logits(self, x):
        return self._victim(x)

    def _init_points(self, ori, target):
        """[B,3,n_add] starting positions of the added points (overridden by the cluster / object variants)."""
        return get_critical_points(self.model, ori, target, self.num_add)

    def _dist(self, adv, ori, weights=None, batch_avg=True):
        return self.dist_func(adv.transpose(1, 2).contiguous(), ori.transpose(1, 2).contiguous(), weights=weights,
                              batch_avg=batch_avg)

    @degrade_on_fp16_range
    def attack(self, data, target):
        """data [B,num_points,3], target [B] -> (o_bestdist float64 [B], float64 [B,num_points+n_add,3], successes)."""
        from ..util.graph_loop import IterationGraph
        self._victim.prepare()
        B, K = data.shape[:2]
        ori = data.float().cuda().detach().transpose(1, 2).contiguous()
        target = target.long().cuda().detach()
        dev = ori.device
        f64 = dict(device=dev, dtype=torch.float64)
        lower = torch.zeros(B, **f64)
        upper = torch.full((B,), float(self.max_weight), **f64)
        init = self._init_points(ori, target)
        n_add = init.shape[2]
        # state of the loop at fixed addresses (updated in place), so that one iteration can be captured and replayed
        weight = torch.full((B,), float(self.init_weight), **f64)
        o_bestdist = torch.full((B,), 1e10, **f64)
        o_bestscore = torch.full((B,), -1, device=dev, dtype=torch.int64)
        o_bestattack = torch.zeros(B, 3, n_add, device=dev)
        bestdist = torch.full((B,), 1e10, **f64)
        bestscore = torch.full((B,), -1, device=dev, dtype=torch.int64)
        last_input = init.clone()
        adv = init.clone().requires_grad_()
        m, v = torch.zeros_like(init), torch.zeros_like(init)
        step = torch.zeros(1, device=dev, dtype=torch.int32)
        adv_loss, dist_loss = torch.zeros((), device=dev), torch.zeros((), device=dev)
        shown_adv, shown_dist = torch.zeros((), device=dev), torch.zeros((), device=dev)
        hits = torch.zeros((), device=dev, dtype=torch.int64)

        def iteration():
            logits = self._logits(torch.cat([ori, adv], dim=-1))
            pred = logits.argmax(dim=-1)
            with torch.no_grad():
                ops.assign(shown_adv, adv_loss)    # the progress line shows the losses of the PREVIOUS iteration (:113-116)
                ops.assign(shown_dist, dist_loss)
                ops.assign(hits, (pred == target).sum())
                ops.assign(last_input, adv)
                dist_val = self._dist(adv, ori, batch_avg=False).detach().double()
                hit = pred == target
                better = hit & (dist_val < bestdist)
                ops.assign(bestdist, torch.where(better, dist_val, bestdist))
                ops.assign(bestscore, torch.where(better, pred, bestscore))
                o_better = hit & (dist_val < o_bestdist)
                ops.assign(o_bestdist, torch.where(o_better, dist_val, o_bestdist))
                ops.assign(o_bestscore, torch.where(o_better, pred, o_bestscore))
                ops.assign(o_bestattack, torch.where(o_better[:, None, None], adv.detach(), o_bestattack))
            a = self.adv_func(logits, target).mean()
            d = self._dist(adv, ori, weights=weight).mean()
            g, = torch.autograd.grad(a + d, adv)
            with torch.no_grad():
                ops.assign(adv_loss, a.detach())
                ops.assign(dist_loss, d.detach())
                ops.adam_single(adv, g, m, v, step, self.attack_lr)  # torch.optim.Adam's update (defaults)

        def start_step(first):
            with torch.no_grad():
                adv.copy_(first)
                m.zero_()
                v.zero_()
                step.zero_()
                bestdist.fill_(1e10)
                bestscore.fill_(-1)
                adv_loss.zero_()
                dist_loss.zero_()

        def start_search():
            with torch.no_grad():
                weight.fill_(float(self.init_weight))
                o_bestdist.fill_(1e10)
                o_bestscore.fill_(-1)
                o_bestattack.zero_()
                last_input.copy_(init)

        total = self.binary_step * self.num_iter
        graph = self.use_graph if total >= 16 else False
        loop = IterationGraph(iteration, graph, 'the point-adding iteration')
        if graph not in (False, 'never'):
            self._victim.open_feed(B, K + n_add, total, dev)  # a sampling victim's draws, device-resident
        if loop.probe():
            start_search()
            start_step(init)
            loop.capture()
        start_search()
        report_every = max(1, self.num_iter // 5)
        for binary_step in range(self.binary_step):
            start_step(init + torch.randn((B, 3, n_add)).cuda() * 1e-7)
            self._victim.load(binary_step * self.num_iter, self.num_iter)
            loop.enter()
            for it in range(self.num_iter):
                loop.step()
                if self.verbose and it % report_every == 0:
                    print('Step {}, iteration {}, success {}/{}\nadv_loss: {:.4f}, dist_loss: {:.4f}'.format(
                        binary_step, it, hits.item(), B, shown_adv.item(), shown_dist.item()))
            loop.leave_step()
            with torch.no_grad():  # bisection (:196-206)
                ok = (bestscore == target) & (bestscore != -1) & (bestdist <= o_bestdist)
                lower = torch.where(ok, torch.maximum(lower, weight), lower)
                upper = torch.where(ok, upper, torch.minimum(upper, weight))
                weight.copy_((lower + upper) / 2.)
        loop.leave()
        self._victim.close_feed()
        self.last_graph_used = loop.reason is None
        with torch.no_grad():
            best = torch.where((lower == 0.)[:, None, None], last_input, o_bestattack)
        success_num = int((lower > 0.).sum().item())
        if self.verbose:
            print('Successfully attack {}/{}'.format(success_num, B))
        out = np.concatenate([ori.cpu().numpy().astype(np.float64), best.double().cpu().numpy()], axis=-1)
        return o_bestdist.cpu().numpy(), out.transpose((0, 2, 1)), success_num
