"""Shared engine of the reference's un-weighted CW variants -- AdvPC / UAdvPC (CW/AdvPC.py, CW/UAdvPC.py), TAOF
(CW/TAOF.py) and UAEAOF (CW/UAEAOF.py).  They are one loop with four switches:

  spectral   optimise the low-frequency component of the cloud in the eigenbasis of its kNN-graph Laplacian (AOF family)
  ae_model   add an adversarial term on the auto-encoder's reconstruction (AdvPC family)
  targeted   success means ``pred == target`` and the second view must differ from ``y_truth`` (AdvPC, TAOF)
  fresh      take the predictions used for best-tracking from a no-grad forward AFTER the clip (AOF, TAOF, AdvPC) or
             from the logits the loss was computed on, before the step (UAdvPC, UAEAOF)

The per-iteration ``.cpu().numpy()`` copies and the Python loop over samples of the reference become [B]-sized device
ops; the spectral split uses ``CW/AOF.py``'s HIP-kNN Laplacian.  The auto-encoder is any module mapping [B,3,K] ->
[B,3,K'] (the reference ships none).
"""
import torch
import torch.optim as optim

from .AOF import get_Laplace_from_pc

from ._victim import Victim

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

    def _run(self, data, target, y_truth=None):
        self._victim.prepare()
        B, K = data.shape[:2]
        ori = data.float().cuda().detach().transpose(1, 2).contiguous()
        target = target.long().cuda().detach()
        if y_truth is not None:
            y_truth = y_truth.long().cuda().detach()
        if self.freeze_model:
            for p in self.model.parameters():
                p.requires_grad = False
        dev = ori.device
        o_bestdist = torch.full((B,), 1e10, device=dev)
        o_bestscore = torch.full((B,), -1, device=dev, dtype=torch.int64)
        o_bestattack = torch.zeros(B, 3, K, device=dev)
        w_full, w_lfc, w_ae = self._weights()
        report_every = max(1, self.num_iter // 5)
        adv = ori
        for binary_step in range(self.binary_step):
            adv = ori.clone() + torch.randn((B, 3, K)).cuda() * 1e-7
            if self.spectral:
                _, V = get_Laplace_from_pc(adv)
                lfc, hfc = self._split(adv, V)
                var = lfc.detach().clone().requires_grad_()
                hfc = hfc.detach().clone()
            else:
                var = adv.requires_grad_()
                hfc = None
            opt = optim.Adam([var], lr=self.attack_lr, weight_decay=0.)
            for iteration in range(self.num_iter):
                full = var + hfc if self.spectral else var
                logits = self._logits(full)
                adv_loss = w_full * self.adv_func(logits, target).mean()
                opt.zero_grad()
                adv_loss.backward()
                shown = adv_loss.item() if self.verbose and iteration % report_every == 0 else 0.
                lfc_logits = ae_logits = None
                if self.ae_model is not None:  # the reference's order of backward calls: full cloud, AE view, low-pass view
                    ae_logits = self._logits(self.ae_model(full))
                    ae_loss = w_ae * self.adv_func(ae_logits, target).mean()
                    ae_loss.backward()
                    if not self.spectral:
                        shown += ae_loss.item() if self.verbose and iteration % report_every == 0 else 0.
                if self.spectral:
                    lfc_logits = self._logits(var)
                    lfc_loss = w_lfc * self.adv_func(lfc_logits, target).mean()
                    lfc_loss.backward()
                    shown += lfc_loss.item() if self.verbose and iteration % report_every == 0 else 0.
                opt.step()
                with torch.no_grad():
                    adv = self.clip_func((var + hfc if self.spectral else var).detach().clone(), ori)
                    if self.spectral:
                        var.data, hfc.data = self._split(adv, V)
                    else:
                        var.data = adv
                    if self.fresh:
                        pred = self._logits(adv).argmax(dim=1)
                        lfc_pred = self._logits(var).argmax(dim=1) if self.spectral else None
                        ae_pred = self._logits(self.ae_model(adv)).argmax(dim=1) if self.ae_model is not None else None
                    else:
                        pred = logits.argmax(dim=1)
                        lfc_pred = lfc_logits.argmax(dim=1) if lfc_logits is not None else None
                        ae_pred = ae_logits.argmax(dim=1) if ae_logits is not None else None
                    dist_val = torch.sqrt(torch.sum((adv - ori) ** 2, dim=[1, 2]))
                    ok = self._better(pred, lfc_pred, ae_pred, target, y_truth) & (dist_val < o_bestdist)
                    o_bestdist = torch.where(ok, dist_val, o_bestdist)
                    o_bestscore = torch.where(ok, pred, o_bestscore)
                    o_bestattack = torch.where(ok[:, None, None], adv, o_bestattack)
                if self.verbose and iteration % report_every == 0:
                    n_ok = self._progress(pred, lfc_pred, ae_pred, target)
                    print('Step {}, iteration {}, success {}/{}\nadv_loss: {:.4f}, dist_loss: {:.4f}'.format(
                        binary_step, iteration, n_ok, B, shown, 0.))
        with torch.no_grad():
            best = torch.where((o_bestscore < 0)[:, None, None], adv, o_bestattack)  # failures: the last iterate
            adv_pc = self.clip_func(best, ori) if self.final_clip else best
            preds = self._logits(adv_pc).argmax(dim=-1)
            success_num = ((preds == target) if self.targeted else (preds != target)).sum().item()
        if self.verbose:
            print('Successfully attack {}/{}'.format(success_num, B))
        return (o_bestdist.double().cpu().numpy(), adv_pc.detach().cpu().numpy().transpose((0, 2, 1)), success_num)

    def _progress(self, pred, lfc_pred, ae_pred, target):
        if self.targeted:
            return (pred == target).sum().item()
        ok = pred != target
        if lfc_pred is not None:
            ok = ok & (lfc_pred != target)
        if ae_pred is not None:
            ok = ok & (ae_pred != target)
        return ok.sum().item()
