"""kNN attack (AAAI'20 "Robust Adversarial Objects"), interface of the reference's CW/kNN.py::CWKNN.

Every iteration runs the victim forward/backward plus ``dist_func`` -- normally ChamferkNNDist, i.e. the fused Chamfer
NN-min kernel and the top-(k+1) kNN kernel with their HIP backwards -- then Adam and ``clip_func``; the iteration works
on fixed buffers and is replayed as one hipGraph when nothing in it needs the host (util/graph_loop.py).  Success is
``pred == target`` (targeted attack, CW/kNN.py:86,146).
"""
import torch

from .. import ops
from ..model._pointwise import degrade_on_fp16_range
from ..util.graph_loop import drive
from ..util.graph_loop import IterationGraph
from ._family import TURN
from ._victim import Victim


class CWKNN:
    """Class for CW attack (constructor of CW/kNN.py:18-38)."""

    def __init__(self, model, adv_func, dist_func, clip_func, attack_lr=1e-3, num_iter=2500, verbose=True,
                 fast_victim=True, use_graph='auto'):
        self.model = model.cuda()
        self.model.eval()
        self._victim = Victim(self.model, fast_victim)
        self.adv_func = adv_func
        self.dist_func = dist_func
        self.clip_func = clip_func
        self.attack_lr = attack_lr
        self.num_iter = num_iter
        self.verbose = verbose
        self.use_graph = use_graph  # 'auto': replay one captured iteration when the victim and the losses allow it
        self.last_graph_used = False

    def _logits(self, x):
        return self._victim(x)

    def _clip(self, adv, ori, normal):
        return self.clip_func(adv, ori)

    @staticmethod
    def _success(pred, target):
        """What the progress lines and the returned count call a success (:90,:141: the targeted criterion)."""
        return pred == target

    @degrade_on_fp16_range
    def attack(self, data, target):
        """data [B,num_points,3 or 6], target [B] -> (float32 ndarray [B,num_points,3], success count)."""
        return drive(self.steps(data, target))

    @property
    def total_iterations(self):
        return self.num_iter

    def steps(self, data, target):
        """``attack`` as a generator with two stops (CW/_family.py::_run_steps): 'ready' (all random numbers drawn, the
        iteration captured) and 'enqueued' (every iteration queued, results not read back yet)."""
        self._victim.prepare()
        B, K = data.shape[:2]
        pc = data.float().cuda().detach().transpose(1, 2).contiguous()
        normal = None if pc.shape[1] == 3 else pc[:, 3:, :].contiguous()
        ori = pc[:, :3, :].contiguous().clone().detach()
        target = target.long().cuda().detach()
        ori_pts = ori.transpose(1, 2).contiguous()
        dev = ori.device
        # the reference draws the jitter on the CPU generator and moves it over (:64-65)
        start = (ori.clone() + torch.randn((B, 3, K)).cuda() * 1e-7)
        # state of the loop: fixed addresses, so that one iteration can be captured and replayed
        adv = start.clone().requires_grad_()
        m, v = torch.zeros_like(start), torch.zeros_like(start)
        step = torch.zeros(1, device=dev, dtype=torch.int32)
        adv_loss, dist_loss = torch.zeros((), device=dev), torch.zeros((), device=dev)
        hits = torch.zeros((), device=dev, dtype=torch.int64)

        def iteration():
            logits = self._logits(adv)
            a = self.adv_func(logits, target).mean()
            d = self.dist_func(adv.transpose(1, 2).contiguous(), ori_pts).mean() * K
            g, = torch.autograd.grad(a + d, adv)
            with torch.no_grad():
                ops.assign(hits, self._success(logits.argmax(dim=1), target).sum())
                ops.assign(adv_loss, a.detach())
                ops.assign(dist_loss, d.detach())
                ops.adam_single(adv, g, m, v, step, self.attack_lr)  # torch.optim.Adam's update (:74, defaults)
                if self.clip_func is not None:
                    ops.assign(adv, self._clip(ops.copy_of(adv), ori, normal))

        def reset():
            with torch.no_grad():
                adv.copy_(start)
                m.zero_()
                v.zero_()
                step.zero_()

        graph = self.use_graph if self.num_iter >= 16 else False
        loop = IterationGraph(iteration, graph, 'the kNN attack iteration')
        self._victim.open_feed(B, K, self.num_iter + 1, dev)  # a sampling victim's draws, device-resident
        starts = self._victim.draw(self.num_iter + 1)  # drawn where the reference's first forward pass would start drawing
        self._victim.put_all([starts])
        capturable = loop.probe()
        yield 'probed'  # (every attack's eager passes before any capture: CW/_family.py)
        if capturable:
            reset()
            loop.capture()
        yield 'ready'
        reset()
        self._victim.seek(0)
        loop.enter()
        report_every = max(1, self.num_iter // 5)
        for it in range(self.num_iter):
            loop.step()
            if self.verbose and it % report_every == 0:
                print('Iteration {}/{}, success {}/{}\nadv_loss: {:.4f}, dist_loss: {:.4f}'.format(
                    it, self.num_iter, hits.item(), B, adv_loss.item(), dist_loss.item()))
            if it % TURN == TURN - 1:
                yield 'turn'
        loop.leave()
        self.last_graph_used = loop.reason is None
        with torch.no_grad():
            hit = self._success(self._logits(adv).argmax(dim=-1), target).sum()
        yield 'enqueued'
        success_num = hit.item()
        self._victim.close_feed()
        if self.verbose:
            print('Successfully attack {}/{}'.format(success_num, B))
        return adv.transpose(1, 2).contiguous().detach().cpu().numpy(), success_num
