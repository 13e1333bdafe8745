"""Adversarial losses on logits (interface of the reference's util/adv_utils.py).

Written with gather / masked_fill instead of a freshly allocated one-hot so that the forward is
allocation-pattern-stable and hipGraph-capturable; the values are bit-identical for finite logits.
"""
import torch.nn as nn
import torch.nn.functional as F


def _true_and_best_other(logits, targets):
    t = targets.long().view(-1, 1)
    true = logits.gather(1, t).squeeze(1)
    other = logits.masked_fill(
        F.one_hot(t.squeeze(1), logits.shape[1]).bool(), -10000.).max(dim=1)[0]
    return true, other


class LogitsAdvLoss(nn.Module):
    """Targeted margin loss, util/adv_utils.py:6-35."""

    def __init__(self, kappa=0.):
        super().__init__()
        self.kappa = kappa

    def forward(self, logits, targets):
        true, other = _true_and_best_other(logits, targets)
        return (other - true + self.kappa).clamp(min=0.).mean()

    def fused(self, logits, targets, loss_out=None):
        """(loss, d loss / d logits) in one HIP launch (hitadv_adv_loss); CUDA tensors only."""
        from .. import ops
        return ops.adv_loss(ops.ADV_TARGETED, logits, targets, self.kappa, loss_out)

    def fused_kind(self):
        """(kind, kappa) for kernels that evaluate this loss inside a larger launch (hitadv_iteration_head)."""
        return 1, self.kappa


class UntargetedLogitsAdvLoss(nn.Module):
    """Untargeted margin loss, util/adv_utils.py:38-67 (the one eval.py:84 hands to HiT-ADV)."""

    def __init__(self, kappa=0.):
        super().__init__()
        self.kappa = kappa

    def forward(self, logits, targets):
        true, other = _true_and_best_other(logits, targets)
        return (true - other + self.kappa).clamp(min=0.).mean()

    def fused(self, logits, targets, loss_out=None):
        """(loss, d loss / d logits) in one HIP launch (hitadv_adv_loss); CUDA tensors only."""
        from .. import ops
        return ops.adv_loss(ops.ADV_UNTARGETED, logits, targets, self.kappa, loss_out)

    def fused_kind(self):
        return 0, self.kappa


class CrossEntropyAdvLoss(nn.Module):
    """util/adv_utils.py:70-85."""

    def forward(self, logits, targets):
        return F.cross_entropy(logits, targets)

    def fused(self, logits, targets, loss_out=None):
        from .. import ops
        return ops.adv_loss(ops.ADV_CROSS_ENTROPY, logits, targets, 0., loss_out)

    def fused_kind(self):
        return 2, 0.
