"""Set distances on the GPU.  Interface of the reference's util/set_distance.py:
``chamfer(preds, gts) -> (loss1[B], loss2[B])`` and ``hausdorff(...)`` module singletons.

The two min-reductions never see a materialised [B,N2,N1] matrix: one fused HIP kernel
(hitadv_nn_min) yields both directions' minima and arg-minima, and autograd flows through the
saved arg-minima exactly as it does through ``torch.min`` in the reference.
"""
import torch.nn as nn

from .. import ops


class _Distance(nn.Module):
    """``reference_arithmetic`` = True: squared distances in the reference's own Gram form (values equal the
    reference's bit for bit), False: direct form, None (default): follow ``ops.reference_arithmetic``."""

    def __init__(self, reference_arithmetic=None):
        super().__init__()
        self.reference_arithmetic = reference_arithmetic

    def forward(self, preds, gts):
        raise NotImplementedError

    def batch_pairwise_dist(self, x, y):
        """Materialised squared-distance matrix [B,Nx,Ny] (util/set_distance.py:15-32).
        Kept for callers that want P itself; evaluated in the reference's Gram form."""
        return ops.pairwise_sqdist(x, y, ops.FORM_GRAM)

    def _nearest(self, preds, gts):
        # rows = gts (N2), columns = preds (N1), as in `P = batch_pairwise_dist(gts, preds)`
        min_gt, _, min_pred, _ = ops.nn_min(gts, preds, self.reference_arithmetic)
        return min_pred, min_gt  # [B,N1] nearest gt of every pred, [B,N2] nearest pred of every gt


class ChamferDistance(_Distance):
    """util/set_distance.py:35-50."""

    def forward(self, preds, gts):
        to_gt, to_pred = self._nearest(preds, gts)
        return to_gt.mean(dim=1), to_pred.mean(dim=1)


class HausdorffDistance(_Distance):
    """util/set_distance.py:53-70."""

    def forward(self, preds, gts):
        to_gt, to_pred = self._nearest(preds, gts)
        return to_gt.max(dim=1)[0], to_pred.max(dim=1)[0]


chamfer = ChamferDistance()
hausdorff = HausdorffDistance()
