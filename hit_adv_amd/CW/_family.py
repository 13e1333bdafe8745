"""Shared engine of the reference's un-weighted CW variants -- AdvPC / UAdvPC (CW/AdvPC.py, CW/UAdvPC.py), TAOF
(CW/TAOF.py) and UAEAOF (CW/UAEAOF.py).  They are one loop with four switches:

  spectral   optimise the low-frequency component of the cloud in the eigenbasis of its kNN-graph Laplacian (AOF family)
  ae_model   add an adversarial term on the auto-encoder's reconstruction (AdvPC family)
  targeted   success means ``pred == target`` and the second view must differ from ``y_truth`` (AdvPC, TAOF)
  fresh      take the predictions used for best-tracking from a no-grad forward AFTER the clip (AOF, TAOF, AdvPC) or
             from the logits the loss was computed on, before the step (UAdvPC, UAEAOF)

The per-iteration ``.cpu().numpy()`` copies and the Python loop over samples of the reference become [B]-sized device
ops on fixed buffers, the iteration is replayed as a hipGraph (``use_graph=`` in the constructors, ``last_graph_used``
afterwards); the spectral split uses the HIP-kNN Laplacian of ``CW/_spectral.py``.  The auto-encoder is any module mapping [B,3,K] ->
[B,3,K'] (the reference ships none).
"""
import torch

from .. import ops
from ._spectral import get_Laplace_from_pc
from ._victim import Victim
from ..model._pointwise import degrade_on_fp16_range
from ..util.graph_loop import drive


# Iterations an attack queues before a driver of several attacks (``CW.attack_concurrently``) moves on to the next one.  The
# runtime lets the host run only so far ahead of the GPU (a captured PCT iteration is thousands of queue packets): a host that
# queues one attack's whole loop first spends that attack's full GPU time doing so, and the others start when it is done.
TURN = 4


class _CWFamily:
    spectral = False
    targeted = False
    fresh = True
    final_clip = True
    freeze_model = False

    def _setup(self, model, adv_func, dist_func, attack_lr, binary_step, num_iter, GAMMA, clip_func, verbose,
               ae_model=None, low_pass=None):
        self.model = model.cuda()
        self.model.eval()
        self._victim = Victim(self.model, getattr(self, 'fast_victim', True))
        self.last_graph_used = False
        self.ae_model = None
        if ae_model is not None:
            self.ae_model = ae_model.cuda()
            self.ae_model.eval()
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

    def _weights(self):
        """(full cloud, low-frequency part, auto-encoder view) loss weights."""
        g = self.GAMMA
        if self.spectral and self.ae_model is not None:
            return 1 - 2 * g, g, g
        if self.spectral:
            return 1 - g, g, 0.
        return 1 - g, 0., g

    def _better(self, pred, lfc_pred, ae_pred, target, y_truth):
        """Which samples count as attacked at this iterate (the reference's per-sample `if`)."""
        if self.targeted:
            other = lfc_pred if self.spectral else ae_pred
            return (pred == target) & (other != y_truth)
        ok = pred != target
        if self.spectral and self.ae_model is not None:  # UAEAOF.py:202
            return ok & (lfc_pred != target) & (ae_pred != target)
        other = lfc_pred if self.spectral else ae_pred
        return ok & ((other != target) | (self.GAMMA < 0.001))

    @degrade_on_fp16_range
    def _run(self, data, target, y_truth=None):
        return drive(self._run_steps(data, target, y_truth))

    @property
    def total_iterations(self):
        return self.binary_step * self.num_iter

    def steps(self, *args):
        """The attack as a generator (``CW.attack_concurrently``): same arguments and return value as ``attack``."""
        out = yield from self._run_steps(*args)
        return self._shape_result(out)

    @staticmethod
    def _shape_result(out):
        """What ``attack`` makes of ``_run``'s (o_bestdist, clouds, successes); the AOF classes drop the first."""
        return out

    def _run_steps(self, data, target, y_truth=None):
        """The loop of CW/AdvPC.py:63-79 / CW/AOF.py:106-134 on fixed buffers: one iteration -- up to three victim passes with
        their input gradients (summed in the reference's order of ``backward`` calls), Adam, clip, the spectral re-split,
        the fresh predictions and the best-so-far bookkeeping -- is one body that is captured into a hipGraph and replayed
        ``num_iter`` times per binary step when nothing in it needs the host (util/graph_loop.py).

        A generator with stops (``util/graph_loop.py::drive`` runs it through; ``CW.attack_concurrently`` interleaves
        several): ``'probed'`` after the eager probing passes, ``'ready'`` when every random number of the attack has been drawn -- in the reference's order: per binary
        step the jitter, then the victim's FPS starts pass by pass, last the pass that counts the successes --, uploaded, and
        the iteration is captured; ``'turn'`` every ``TURN`` queued iterations; ``'enqueued'`` when everything is queued on
        the attack's stream and the next thing is the host reading results back."""
        from .. import ops
        from ..util.graph_loop import IterationGraph
        self._victim.prepare()
        B, K = data.shape[:2]
        ori = data.float().cuda().detach().transpose(1, 2).contiguous()
        if ori.shape[1] == 6:
            ori = ori[:, :3, :].contiguous()
        target = target.long().cuda().detach()
        if y_truth is not None:
            y_truth = y_truth.long().cuda().detach()
        if self.freeze_model:
            for p in self.model.parameters():
                p.requires_grad = False
        dev = ori.device
        spectral, ae = self.spectral, self.ae_model
        o_bestdist = torch.full((B,), 1e10, device=dev)
        o_bestscore = torch.full((B,), -1, device=dev, dtype=torch.int64)
        o_bestattack = torch.zeros(B, 3, K, device=dev)
        w_full, w_lfc, w_ae = self._weights()
        report_every = max(1, self.num_iter // 5)
        # state of the loop at fixed addresses
        var = ori.clone().requires_grad_()            # the optimised tensor: the cloud, or its low-frequency part
        hfc = torch.zeros_like(ori) if spectral else None
        V = torch.zeros(B, K, K, device=dev) if spectral else None
        adv = ori.clone()                             # the clipped iterate
        m, v = torch.zeros_like(ori), torch.zeros_like(ori)
        step = torch.zeros(1, device=dev, dtype=torch.int32)
        shown = torch.zeros((), device=dev)
        n_ok = torch.zeros((), device=dev, dtype=torch.int64)

        def iteration():
            full = var + hfc if spectral else var
            logits = self._logits(full)
            adv_loss = w_full * self.adv_func(logits, target).mean()
            g, = torch.autograd.grad(adv_loss, var)
            total = adv_loss.detach()
            lfc_logits = ae_logits = None
            if ae is not None:  # the reference's order of backward calls: full cloud, AE view, low-pass view
                ae_logits = self._logits(ae(full))
                ae_loss = w_ae * self.adv_func(ae_logits, target).mean()
                g = g + torch.autograd.grad(ae_loss, var)[0]
                if not spectral:
                    total = total + ae_loss.detach()
            if spectral:
                lfc_logits = self._logits(var)
                lfc_loss = w_lfc * self.adv_func(lfc_logits, target).mean()
                g = g + torch.autograd.grad(lfc_loss, var)[0]
                total = total + lfc_loss.detach()
            with torch.no_grad():
                ops.assign(shown, total)
                ops.adam_single(var, g, m, v, step, self.attack_lr)  # torch.optim.Adam's update (defaults, no weight decay)
                ops.assign(adv, self.clip_func(var + hfc if spectral else ops.copy_of(var), ori))
                if spectral:
                    new_l, new_h = self._split(adv, V)
                    ops.assign(var, new_l)
                    ops.assign(hfc, new_h)
                else:
                    ops.assign(var, adv)
                if self.fresh:
                    pred = self._logits(adv).argmax(dim=1)
                    lfc_pred = self._logits(var).argmax(dim=1) if spectral else None
                    ae_pred = self._logits(ae(adv)).argmax(dim=1) if ae is not None else None
                else:
                    pred = logits.argmax(dim=1)
                    lfc_pred = lfc_logits.argmax(dim=1) if lfc_logits is not None else None
                    ae_pred = ae_logits.argmax(dim=1) if ae_logits is not None else None
                dist_val = torch.sqrt(torch.sum((adv - ori) ** 2, dim=[1, 2]))
                ok = self._better(pred, lfc_pred, ae_pred, target, y_truth) & (dist_val < o_bestdist)
                ops.assign(o_bestdist, torch.where(ok, dist_val, o_bestdist))
                ops.assign(o_bestscore, torch.where(ok, pred, o_bestscore))
                ops.assign(o_bestattack, torch.where(ok[:, None, None], adv, o_bestattack))
                ops.assign(n_ok, self._progress(pred, lfc_pred, ae_pred, target))

        def start_step(init):
            with torch.no_grad():
                adv.copy_(init)
                if spectral:
                    V.copy_(get_Laplace_from_pc(init)[1])
                    lfc0, hfc0 = self._split(init, V)
                    var.copy_(lfc0)
                    hfc.copy_(hfc0)
                else:
                    var.copy_(init)
                m.zero_()
                v.zero_()
                step.zero_()

        def start_search():
            with torch.no_grad():
                o_bestdist.fill_(1e10)
                o_bestscore.fill_(-1)
                o_bestattack.zero_()

        passes = (1 + (ae is not None) + spectral) * (2 if self.fresh else 1)  # victim forward passes per iteration
        total_iters = self.binary_step * self.num_iter
        graph = getattr(self, 'use_graph', 'auto') if total_iters >= 16 else False
        loop = IterationGraph(iteration, graph, 'the %s iteration' % type(self).__name__)
        # a sampling victim's FPS starts live in a device table (+ 1 row: the pass that counts the successes at the end)
        self._victim.open_feed(B, K, total_iters * passes + 1, dev)
        # every random number of the attack, now, in the reference's order
        per_step = self.num_iter * passes
        jitter, starts = [], []
        for _ in range(self.binary_step):
            jitter.append(torch.randn((B, 3, K)).cuda() * 1e-7)
            starts.append(self._victim.draw(per_step))
        last_starts = self._victim.draw(1)
        self._victim.put_all(starts + [last_starts])  # one upload, here: none inside the loop (it would hold the host)
        capturable = loop.probe()
        yield 'probed'  # a driver of several attacks runs EVERY attack's eager probing passes before ANY capture: on this stack a
        #                 replay that follows eager victim work issued after a capture can fault
        if capturable:
            start_search()
            if spectral:
                with torch.no_grad():
                    V.zero_()  # the probing passes only need SOME basis; the first binary step computes the real one
            loop.capture()
        yield 'ready'
        start_search()
        for binary_step in range(self.binary_step):
            start_step(ori.clone() + jitter[binary_step])
            self._victim.seek(binary_step * per_step)
            loop.enter()
            for it in range(self.num_iter):
                loop.step()
                if self.verbose and it % report_every == 0:
                    print('Step {}, iteration {}, success {}/{}\nadv_loss: {:.4f}, dist_loss: {:.4f}'.format(
                        binary_step, it, n_ok.item(), B, shown.item(), 0.))
                if it % TURN == TURN - 1:
                    yield 'turn'  # a driver of several attacks goes round here (see TURN)
            loop.leave_step()
            yield 'turn'
        loop.leave()
        self.last_graph_used = loop.reason is None
        with torch.no_grad():
            best = torch.where((o_bestscore < 0)[:, None, None], adv, o_bestattack)  # failures: the last iterate
            adv_pc = self.clip_func(best, ori) if self.final_clip else best
            self._victim.seek(total_iters * passes)
            preds = self._logits(adv_pc).argmax(dim=-1)
        yield 'enqueued'
        success_num = ((preds == target) if self.targeted else (preds != target)).sum().item()
        self._victim.close_feed()
        if self.verbose:
            print('Successfully attack {}/{}'.format(success_num, B))
        return (o_bestdist.double().cpu().numpy(), adv_pc.detach().cpu().numpy().transpose((0, 2, 1)), success_num)

    def _progress(self, pred, lfc_pred, ae_pred, target):
        """Number of samples the progress line counts as attacked (a 0-d device tensor: read only when a line is printed)."""
        if self.targeted:
            return (pred == target).sum()
        ok = pred != target
        if lfc_pred is not None:
            ok = ok & (lfc_pred != target)
        if ae_pred is not None:
            ok = ok & (ae_pred != target)
        return ok.sum()
