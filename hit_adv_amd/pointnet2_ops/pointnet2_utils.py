"""autograd front end of the pointnet2_ops natives; public names and call signatures of the
reference's pointnet2_ops/pointnet2_utils.py (:34-276): furthest_point_sample, gather_operation,
three_nn, three_interpolate, grouping_operation, ball_query, QueryAndGroup, GroupAll.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _ext


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        """xyz (B,N,3) float -> (B,npoint) int32 indices  (pointnet2_utils.py:34-63)"""
        out = _ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        return ()


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint) int32 -> (B,C,npoint)  (:68-99)"""
        ctx.save_for_backward(idx)
        ctx.n = features.shape[2]
        return _ext.gather_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, = ctx.saved_tensors
        return _ext.gather_points_grad(grad_out.contiguous(), idx, ctx.n), None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        """unknown (B,n,3), known (B,m,3) -> (dist (B,n,3) l2 distance, idx (B,n,3) int32)  (:104-134)"""
        dist2, idx = _ext.three_nn(unknown, known)
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, grad_dist, grad_idx):
        return ()


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        """features (B,c,m), idx (B,n,3), weight (B,n,3) -> (B,c,n)  (:139-189)"""
        ctx.save_for_backward(idx, weight)
        ctx.m = features.shape[2]
        return _ext.three_interpolate(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        return _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight, ctx.m), None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint,nsample) int32 -> (B,C,npoint,nsample)  (:194-238)"""
        ctx.save_for_backward(idx)
        ctx.n = features.shape[2]
        return _ext.group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, = ctx.saved_tensors
        return _ext.group_points_grad(grad_out.contiguous(), idx, ctx.n), None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        """radius, nsample, xyz (B,N,3), new_xyz (B,npoint,3) -> (B,npoint,nsample) int32  (:243-273).
        Note the Python argument order (xyz before new_xyz) against the native one."""
        out = _ext.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        return ()


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """Ball-query grouping (:279-333)."""

    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        grouped = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        grouped = grouped - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            return grouped
        feats = grouping_operation(features, idx)
        return torch.cat([grouped, feats], dim=1) if self.use_xyz else feats


class GroupAll(nn.Module):
    """Single group holding every point (:336-375)."""

    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped
        feats = features.unsqueeze(2)
        return torch.cat([grouped, feats], dim=1) if self.use_xyz else feats
