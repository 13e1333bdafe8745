"""PointNet++ SSG victim (cfg4 of BASELINE.json).  Parameter names follow the reference's
model/pointnet2_cls_ssg.py::get_model (:6-42) and model/pointnet2_utils.py::PointNetSetAbstraction (:161-205),
79 state_dict entries (tests/golden/g8_state_dicts.json).  The MLPs stay PyTorch-ROCm; the geometry runs in HIP:

* ``farthest_point_sample`` (pointnet2_utils.py:63-84): the random first index is drawn from the CPU generator
  exactly as the reference does (:75) -- or read from an attack's pre-drawn feed (_sampling.py), which makes the
  forward pass capturable --, the 512/128 sequential arg-max steps run in ``hitadv_fps_from_start``;
* ``query_ball_point`` (:87-107): ``hitadv_query_ball_point_victim`` (not ``d^2 > r^2`` on the reference's Gram-form
  ``square_distance`` values, first nsample in index order, padded with the first hit) instead of a [B,S,N] distance
  matrix + full sort; the table is the reference's bit for bit (fixture g11).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from . import _sampling
from ._pointwise import (fast_pm, grouped_first_two_then_max_pm, linear_lrelu_maxpool_pm, linear_relu_max_pm, linear_relu_pm,
                         linear_relu_then_max_pm, split_first_layer)


def index_points(points, idx):
    """points [B,N,C], idx [B,S] or [B,S,K] -> [B,S,C] / [B,S,K,C]."""
    B, _, C = points.shape
    flat = idx.reshape(B, -1, 1).expand(-1, -1, C)
    return points.gather(1, flat).reshape(*idx.shape, C)


def farthest_point_sample(xyz, npoint):
    B, N, _ = xyz.shape
    return ops.fps_from_start(xyz, npoint, _sampling.next_start(B, N, xyz.device))


def query_ball_point(radius, nsample, xyz, new_xyz):
    return ops.query_ball_point(radius, nsample, xyz, new_xyz)


def sample_and_group(npoint, radius, nsample, xyz, points):
    B, N, C = xyz.shape
    new_xyz = index_points(xyz, farthest_point_sample(xyz, npoint))
    idx = query_ball_point(radius, nsample, xyz, new_xyz)
    grouped = index_points(xyz, idx) - new_xyz.view(B, npoint, 1, C)
    if points is not None:
        grouped = torch.cat([grouped, index_points(points, idx)], dim=-1)
    return new_xyz, grouped


def sample_and_group_all(xyz, points):
    B, N, C = xyz.shape
    new_xyz = torch.zeros(B, 1, C, device=xyz.device)
    grouped = xyz.view(B, 1, N, C)
    if points is not None:
        grouped = torch.cat([grouped, points.view(B, 1, N, -1)], dim=-1)
    return new_xyz, grouped


class PointNetSetAbstraction(nn.Module):
    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all):
        super().__init__()
        self.npoint, self.radius, self.nsample, self.group_all = npoint, radius, nsample, group_all
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last = in_channel
        for out in mlp:
            self.mlp_convs.append(nn.Conv2d(last, out, 1))
            self.mlp_bns.append(nn.BatchNorm2d(out))
            last = out

    def forward(self, xyz, points):
        """xyz [B,3,N], points [B,D,N] or None -> (new_xyz [B,3,S], features [B,D',S])."""
        xyz = ops.points_major(xyz)
        if points is not None:
            points = points.permute(0, 2, 1)
        if (not self.group_all and fast_pm(self.mlp_convs[0], self.mlp_bns[0], xyz) and
                ops.group_add_relu_supported(self.mlp_convs[0].out_channels, self.nsample)):
            return self._grouped_from_points(xyz, points)
        if self.group_all:
            new_xyz, grouped = sample_and_group_all(xyz, points)
        else:
            new_xyz, grouped = sample_and_group(self.npoint, self.radius, self.nsample, xyz, points)
        if fast_pm(self.mlp_convs[0], self.mlp_bns[0], grouped):
            h = grouped  # [B,npoint,nsample,C+D] is already points-major: the shared MLP is a chain of GEMMs
            layers = list(zip(self.mlp_convs, self.mlp_bns))
            if self.group_all and h.shape[1] == 1:
                # the last shared layer, its ReLU and the max over the cloud's points as one kernel on the fp16 matrix cores
                # (the pooled GEMM of DGCNN's embedding layer with slope 0: relu and max commute); the [B, n, 1024] activation,
                # its ReLU pass and its max pass never exist, forward or backward
                for conv, bn in layers[:-1]:
                    h = linear_relu_pm(conv, bn, h)
                out = linear_lrelu_maxpool_pm(layers[-1][0], layers[-1][1], h[:, 0], slope=0.)  # [B, C]
                return new_xyz.permute(0, 2, 1), out.unsqueeze(-1)
            for conv, bn in layers:
                h = linear_relu_pm(conv, bn, h)
            return new_xyz.permute(0, 2, 1), h.max(dim=2)[0].permute(0, 2, 1)
        h = grouped.permute(0, 3, 2, 1)  # [B,C+D,nsample,npoint]
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            h = F.relu(bn(conv(h)))
        return new_xyz.permute(0, 2, 1), torch.max(h, 2)[0]


def _grouped_from_points(self, xyz, points):
    """The grouping path without the grouped tensor (eval mode on the GPU).  The first 1x1 convolution acts on
    [x_j - c_i ; f_j]; split over its inputs it is W [x_j ; f_j] - Wx c_i: one GEMM over the N points (U), one over the S
    centres (V) and ``hitadv_group_add_relu`` replace the [B,S,nsample,3+D] gather / subtract / concat and a GEMM over
    S*nsample rows.  Same random FPS start, same ball query as ``sample_and_group``."""
    B, N, _ = xyz.shape
    new_xyz = index_points(xyz, farthest_point_sample(xyz, self.npoint))
    idx = query_ball_point(self.radius, self.nsample, xyz, new_xyz)
    W, t = split_first_layer(self.mlp_convs[0], self.mlp_bns[0], 3)
    src = xyz if points is None else torch.cat([xyz, points], dim=-1)
    U = torch.matmul(src, W.t())
    V = torch.addmm(t, new_xyz.reshape(-1, 3), -W[:, :3].t()).view(B, self.npoint, W.shape[0])
    rest = list(zip(self.mlp_convs, self.mlp_bns))[1:]
    if len(rest) == 2:
        # three shared layers (the reference's blocks): the gather / add / ReLU of the first one runs inside the middle layer's
        # kernel, the last one is fused with the max over the neighbours (csrc/rows_linear.hip, csrc/group_mlp.hip)
        out = grouped_first_two_then_max_pm(U, V, idx, rest[0][0], rest[0][1], rest[1][0], rest[1][1])
        return new_xyz.permute(0, 2, 1), out.permute(0, 2, 1)
    h = ops.group_add_relu(U, V, idx)
    for conv, bn in rest[:-2]:
        h = linear_relu_pm(conv, bn, h)
    # the last shared layer and the max over the neighbours in one kernel (csrc/group_mlp.hip) where the shape allows; the layer
    # in front of it hands its ReLU backward to that kernel
    if len(rest) >= 2:
        out = linear_relu_then_max_pm(rest[-2][0], rest[-2][1], rest[-1][0], rest[-1][1], h)
    else:
        out = linear_relu_max_pm(rest[-1][0], rest[-1][1], h)
    return new_xyz.permute(0, 2, 1), out.permute(0, 2, 1)


PointNetSetAbstraction._grouped_from_points = _grouped_from_points


class get_model(nn.Module):
    """``forward(xyz[B,3|6,N]) -> (logits, l3_points)`` -- a tuple, like the reference (pointnet2_cls_ssg.py:42)."""

    def __init__(self, num_class, normal_channel=True):
        super().__init__()
        in_channel = 6 if normal_channel else 3
        self.normal_channel = normal_channel
        self.sa1 = PointNetSetAbstraction(512, 0.2, 32, in_channel, [64, 64, 128], False)
        self.sa2 = PointNetSetAbstraction(128, 0.4, 64, 128 + 3, [128, 128, 256], False)
        self.sa3 = PointNetSetAbstraction(None, None, None, 256 + 3, [256, 512, 1024], True)
        self.fc1 = nn.Linear(1024, 512)
        self.bn1 = nn.BatchNorm1d(512)
        self.drop1 = nn.Dropout(0.4)
        self.fc2 = nn.Linear(512, 256)
        self.bn2 = nn.BatchNorm1d(256)
        self.drop2 = nn.Dropout(0.4)
        self.fc3 = nn.Linear(256, num_class)

    def fps_start_plan(self, N):
        """One ``randint(0, high, (B,))`` per FPS call of a forward pass, in call order (sa1 samples the N input
        points, sa2 the 512 centres of sa1; sa3 groups everything and draws nothing)."""
        return [N, self.sa1.npoint]

    def forward(self, xyz):
        B = xyz.shape[0]
        norm = xyz[:, 3:, :] if self.normal_channel else None
        xyz = xyz[:, :3, :]
        l1_xyz, l1 = self.sa1(xyz, norm)
        l2_xyz, l2 = self.sa2(l1_xyz, l1)
        _, l3 = self.sa3(l2_xyz, l2)
        h = l3.view(B, 1024)
        h = self.drop1(F.relu(self.bn1(self.fc1(h))))
        h = self.drop2(F.relu(self.bn2(self.fc2(h))))
        return self.fc3(h), l3
