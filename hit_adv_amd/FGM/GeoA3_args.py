"""``uniform_loss`` -- the evaluation metric through which the reference reaches its CUDA extension
(FGM/GeoA3_args.py:258-302; called from util/other_utils.py:38,74).  Same signature.  FPS, gather,
ball query and grouping run in the HIP re-writes of the pointnet2_ops natives; the in-ball kNN runs in
the HIP kNN kernel.
"""
import math

import torch

from ..pointnet2_ops import pointnet2_utils
from ..pytorch3d_ops import knn_points


def uniform_loss(adv_pc, percentages=[0.004, 0.006, 0.008, 0.010, 0.012], radius=1.0, k=2):
    """adv_pc [B,3,N] or [B,N,3] -> scalar: mean over five ball sizes of the squared deviation of the
    in-ball nearest-neighbour spacing from the spacing of a uniform disc sample."""
    if adv_pc.size(1) == 3:
        adv_pc = adv_pc.permute(0, 2, 1).contiguous()
    adv_pc = adv_pc.float().contiguous()
    b, n, _ = adv_pc.size()
    npoint = int(n * 0.05)
    channels_first = adv_pc.transpose(1, 2).contiguous()
    # the seeds do not depend on the ball size: sample them once (the reference recomputes the same
    # deterministic FPS five times, GeoA3_args.py:271-273)
    seeds = pointnet2_utils.furthest_point_sample(adv_pc, npoint)
    new_xyz = pointnet2_utils.gather_operation(channels_first, seeds).transpose(1, 2).contiguous()
    total = None
    for p in percentages:
        p = p * 4
        nsample = int(n * p)
        r = math.sqrt(p * radius)
        expect_len = math.sqrt(math.pi * (radius ** 2) * p / nsample)
        idx = pointnet2_utils.ball_query(r, nsample, adv_pc, new_xyz)  # [B,npoint,nsample]
        grouped = pointnet2_utils.grouping_operation(channels_first, idx)  # [B,3,npoint,nsample]
        grouped = grouped.permute(2, 0, 3, 1).reshape(npoint * b, nsample, 3).contiguous()
        d = knn_points(grouped, grouped, K=k + 1).dists[:, :, 1:]
        spacing = torch.sqrt(torch.abs(d) + 1e-12).mean(dim=-1)
        dev = ((spacing - expect_len) ** 2 / (expect_len + 1e-12)).reshape(-1)
        term = dev.mean() * math.pow(p * 100, 2)
        total = term if total is None else total + term
    return total / len(percentages)
