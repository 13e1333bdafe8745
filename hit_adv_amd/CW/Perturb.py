"""CW point-perturbation attack (CVPR'19 "Generating 3D Adversarial Point Clouds"), interface of the
reference's CW/Perturb.py::CWPerturb (ctor :16-44, attack :46-202): binary search over a per-sample distance
weight, Adam on the xyz of every point, best-result tracking.  Success means ``pred == target`` (the class
is written for targeted attacks, :107,134,141,178).

Device-resident restatement: the per-iteration ``.cpu().numpy()`` copies and the Python loop over samples
(:127-145) become a few [B]-sized tensor ops on fixed buffers; the bisection keeps the reference's float64 bounds.  One
iteration -- victim forward / backward, losses, Adam, clip, best tracking -- is captured into a hipGraph and replayed
``binary_step x num_iter`` times when nothing in it needs the host (util/graph_loop.py).
"""
import torch

from .. import ops
from ..model._pointwise import degrade_on_fp16_range
from ..util.graph_loop import IterationGraph
from ._victim import Victim


class CWPerturb:
    """Class for CW attack."""

    def __init__(self, model, adv_func, dist_func, attack_lr=1e-2, init_weight=10., max_weight=80.,
                 binary_step=10, num_iter=500, pre_head=None, clip_func=None, verbose=True, fast_victim=True,
                 use_graph='auto'):
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
        self.use_graph = use_graph  # 'auto': replay one captured iteration when the victim and the losses allow it
        self.last_graph_used = False

    def _logits(self, x):
        return self._victim(self.pre_head(x) if self.pre_head is not None else x)

    @degrade_on_fp16_range
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
        # state of the loop: fixed addresses (updated in place), so that one iteration can be captured and replayed
        weight = torch.full((B,), float(self.init_weight), **f64)
        o_bestdist = torch.full((B,), 1e10, **f64)
        o_bestscore = torch.full((B,), -1, device=dev, dtype=torch.int64)
        o_bestattack = torch.zeros(B, 3, K, device=dev)
        bestdist = torch.full((B,), 1e10, **f64)
        bestscore = torch.full((B,), -1, device=dev, dtype=torch.int64)
        last_input = ori.clone()
        adv = ori.clone().requires_grad_()
        m, v = torch.zeros_like(ori), torch.zeros_like(ori)
        step = torch.zeros(1, device=dev, dtype=torch.int32)
        adv_loss, dist_loss = torch.zeros((), device=dev), torch.zeros((), device=dev)
        hits = torch.zeros((), device=dev, dtype=torch.int64)

        def iteration():
            logits = self._logits(adv)
            pred = logits.argmax(dim=1)
            with torch.no_grad():
                ops.assign(last_input, adv)  # what the reference calls input_val (:125)
                dist_val = torch.sqrt(torch.sum((adv - ori) ** 2, dim=[1, 2])).double()
                hit = pred == target
                ops.assign(hits, hit.sum())
                better = hit & (dist_val < bestdist)
                ops.assign(bestdist, torch.where(better, dist_val, bestdist))
                ops.assign(bestscore, torch.where(better, pred, bestscore))
                o_better = hit & (dist_val < o_bestdist)
                ops.assign(o_bestdist, torch.where(o_better, dist_val, o_bestdist))
                ops.assign(o_bestscore, torch.where(o_better, pred, o_bestscore))
                ops.assign(o_bestattack, torch.where(o_better[:, None, None], adv, o_bestattack))
            a = self.adv_func(logits, target).mean()
            d = self.dist_func(adv, ori, weight).mean()
            g, = torch.autograd.grad(a + d, adv)
            with torch.no_grad():
                ops.assign(adv_loss, a.detach())
                ops.assign(dist_loss, d.detach())
                ops.adam_single(adv, g, m, v, step, self.attack_lr)  # torch.optim.Adam's update (:119, defaults)
                if self.clip_func is not None:
                    ops.assign(adv, self.clip_func(ops.copy_of(adv), ori))

        def start_step(init):
            with torch.no_grad():
                adv.copy_(init)
                m.zero_()
                v.zero_()
                step.zero_()
                bestdist.fill_(1e10)
                bestscore.fill_(-1)

        def start_search():
            with torch.no_grad():
                weight.fill_(float(self.init_weight))
                o_bestdist.fill_(1e10)
                o_bestscore.fill_(-1)
                o_bestattack.zero_()
                last_input.copy_(ori)

        total = self.binary_step * self.num_iter
        graph = self.use_graph if total >= 16 else False
        loop = IterationGraph(iteration, graph, 'the CW perturbation iteration')
        if graph not in (False, 'never'):
            self._victim.open_feed(B, K, total, dev)  # a sampling victim's draws, device-resident
        if loop.probe():
            start_search()
            start_step(ori)
            loop.capture()
        start_search()
        report_every = max(1, self.num_iter // 5)
        for binary_step in range(self.binary_step):
            start_step(ori.clone() + torch.randn((B, 3, K)).cuda() * 1e-7)
            self._victim.load(binary_step * self.num_iter, self.num_iter)  # this step's forward passes, drawn now
            loop.enter()
            for it in range(self.num_iter):
                loop.step()
                if self.verbose and it % report_every == 0:
                    print('Step {}, iteration {}, success {}/{}\nadv_loss: {:.4f}, dist_loss: {:.4f}'.format(
                        binary_step, it, hits.item(), B, adv_loss.item(), dist_loss.item()))
            loop.leave_step()
            with torch.no_grad():  # bisection, :172-184
                ok = (bestscore == target) & (bestscore != -1) & (bestdist <= o_bestdist)
                lower = torch.where(ok, torch.maximum(lower, weight), lower)
                upper = torch.where(ok, upper, torch.minimum(upper, weight))
                weight.copy_((lower + upper) / 2.)
        loop.leave()
        self._victim.close_feed()
        self.last_graph_used = loop.reason is None
        with torch.no_grad():
            fail = lower == 0.
            # failures get input_val, i.e. the iterate that ENTERED the last iteration (:189-193)
            best = torch.where(fail[:, None, None], last_input, o_bestattack)
        success_num = int((lower > 0.).sum().item())
        if self.verbose:
            print('Successfully attack {}/{}'.format(success_num, B))
        return best.double().cpu().numpy().transpose((0, 2, 1)), success_num
