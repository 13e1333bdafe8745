"""Clipping / projection operators for the CW attack family (reference: util/clip_utils.py).
Elementwise work on [B,3,K] tensors; stays on torch elementwise kernels."""
import torch
import torch.nn as nn


class ClipPointsL2(nn.Module):
    """Global L2 budget per cloud, util/clip_utils.py:5-32."""

    def __init__(self, budget):
        super().__init__()
        self.budget = budget

    @torch.no_grad()
    def forward(self, pc, ori_pc):
        delta = pc - ori_pc
        length = delta.pow(2).sum(dim=[1, 2]).pow(0.5)
        shrink = (self.budget / (length + 1e-9)).clamp(max=1.)
        return ori_pc + delta * shrink[:, None, None]


class ClipPointsLinf(nn.Module):
    """Per-coordinate budget, util/clip_utils.py:63-87."""

    def __init__(self, budget):
        super().__init__()
        self.budget = budget

    @torch.no_grad()
    def forward(self, pc, ori_pc):
        return (ori_pc + (pc - ori_pc).clamp(-self.budget, self.budget)).detach()


class ProjectInnerPoints(nn.Module):
    """Push points that moved inside the surface back onto it, util/clip_utils.py:90-140.

    Reference quirk kept (Q7): its second cross product (:121) names no ``dim``, which under the reference's PyTorch
    selects the first dimension of size 3 -- the batch dimension for a batch of exactly three clouds."""

    @torch.no_grad()
    def forward(self, pc, ori_pc, normal=None):
        if normal is None:
            return pc
        delta = pc - ori_pc
        inside = (delta * normal).sum(dim=1) < 0.
        vng = torch.cross(normal, delta, dim=1)
        vng_len = vng.pow(2).sum(dim=1).pow(0.5)
        vref = torch.cross(vng, normal, dim=0 if pc.shape[0] == 3 else 1)
        vref_len = vref.pow(2).sum(dim=1).pow(0.5)
        proj = delta * vref / (vref_len[:, None, :] + 1e-9)
        proj = torch.where((inside & (vng_len < 1e-6))[:, None, :], torch.zeros_like(proj), proj)
        return ori_pc + torch.where(inside[:, None, :], proj, delta)


class ProjectInnerClipLinf(nn.Module):
    """util/clip_utils.py:143-170."""

    def __init__(self, budget):
        super().__init__()
        self.project_inner = ProjectInnerPoints()
        self.clip_linf = ClipPointsLinf(budget=budget)

    @torch.no_grad()
    def forward(self, pc, ori_pc, normal=None):
        return self.clip_linf(self.project_inner(pc, ori_pc, normal), ori_pc)
