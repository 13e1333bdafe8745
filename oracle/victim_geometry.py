"""CPU restatement of the geometric functions inside the reference's PointNet++ and PCT victims (pure torch, op for op).

TEST INFRASTRUCTURE, like everything under oracle/: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import it.  The product runs these steps in HIP (hit_adv_amd/model/pointnet2.py, pct.py -> libhitadv_hip.so) and
raises on CPU tensors; this file is what lets the checker run the same victims on the CPU.

Pinned: with these functions in place of the HIP ones, the victims' plain nn.Module forward reproduces fixtures g11
(PointNet++: FPS table, ball-query table, logits, input gradient) and g12 (PCT: FPS table, logits, input gradient), both
captured from the unmodified reference (tests/test_victims_cpu.py).

``cpu_geometry()`` swaps them into the product's modules for the duration of a ``with`` block.

Two restatements of every sampler live here.  The torch one repeats the reference's tensor operations; what arithmetic
that executes depends on the BLAS torch dispatches to on the host at hand (a K = 3 product is one FMA chain in MKL's GEMM
kernel, ``fma(q1, p1, q0 p0) + q2 p2`` in its one-row path).  The C one (oracle/pointnet2_oracle.c, forms 3 and 4) spells
that arithmetic out; tests/test_oracle_gram.py holds the two equal bit for bit on the build host, and both reproduce the
tables of fixtures g11 / g12.  ``cpu_geometry()`` uses the C one, so the oracle's tables do not depend on which host the
checker runs on (the GPU box's CPU is not the build container's).
"""
import contextlib

import torch

from . import c_oracle as _C


def square_distance(src, dst):
    """model/pointnet2_utils.py:19-41 (and model/pct_utils.py:40-58): Gram form, in this order of operations."""
    B, N, _ = src.shape
    _, M, _ = dst.shape
    dist = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    dist += torch.sum(src ** 2, -1).view(B, N, 1)
    dist += torch.sum(dst ** 2, -1).view(B, 1, M)
    return dist


def index_points(points, idx):
    """model/pointnet2_utils.py:44-60."""
    B = points.shape[0]
    view_shape = list(idx.shape)
    view_shape[1:] = [1] * (len(view_shape) - 1)
    repeat_shape = list(idx.shape)
    repeat_shape[0] = 1
    batch = torch.arange(B, dtype=torch.long).view(view_shape).repeat(repeat_shape)
    return points[batch, idx, :]


def farthest_point_sample(xyz, npoint):
    """model/pointnet2_utils.py:63-84: random start from the global CPU generator (:75), direct-form distances,
    strict ``<`` update, ``torch.max`` arg-max."""
    B, N, _ = xyz.shape
    centroids = torch.zeros(B, npoint, dtype=torch.long)
    distance = torch.ones(B, N, dtype=xyz.dtype) * 1e10
    farthest = torch.randint(0, N, (B,), dtype=torch.long)
    batch = torch.arange(B, dtype=torch.long)
    for i in range(npoint):
        centroids[:, i] = farthest
        centroid = xyz[batch, farthest, :].view(B, 1, 3)
        dist = torch.sum((xyz - centroid) ** 2, -1)
        mask = dist < distance
        distance[mask] = dist[mask]
        farthest = torch.max(distance, -1)[1]
    return centroids


def query_ball_point(radius, nsample, xyz, new_xyz):
    """model/pointnet2_utils.py:87-107: Gram-form distances, ``> r^2`` excluded, first ``nsample`` in index order,
    padded with the first hit."""
    B, N, _ = xyz.shape
    _, S, _ = new_xyz.shape
    group_idx = torch.arange(N, dtype=torch.long).view(1, 1, N).repeat([B, S, 1])
    sqrdists = square_distance(new_xyz, xyz)
    group_idx[sqrdists > radius ** 2] = N
    group_idx = group_idx.sort(dim=-1)[0][:, :, :nsample]
    group_first = group_idx[:, :, 0].view(B, S, 1).repeat([1, 1, nsample])
    mask = group_idx == N
    group_idx[mask] = group_first[mask]
    return group_idx


def get_dists_squared(points1, points2):
    """util/other_utils.py:237-250: the clamped Gram form, everything of ``get_dists`` but its last line."""
    B, M, _ = points1.shape
    _, N, _ = points2.shape
    dists = torch.sum(torch.pow(points1, 2), dim=-1).view(B, M, 1) + torch.sum(torch.pow(points2, 2), dim=-1).view(B, 1, N)
    dists -= 2 * torch.matmul(points1, points2.permute(0, 2, 1))
    return torch.where(dists < 0, torch.ones_like(dists) * 1e-7, dists)


def get_dists(points1, points2):
    """util/other_utils.py:237-251: sqrt of the clamped Gram form.  NOTE: ``torch.sqrt`` of a contiguous fp32 CPU tensor is
    MKL VML's vsSqrt in its default accuracy mode (< 1 ulp, NOT correctly rounded): on the build host 0.5 % of the values
    are one ulp off the IEEE result.  The C restatement and the HIP kernel round correctly (tests/test_oracle_gram.py)."""
    return torch.sqrt(get_dists_squared(points1, points2)).to(points1.dtype)


def pct_fps(xyz, M):
    """util/other_utils.py:254-272 (PCT's sampler): start from the CPU generator (:264), distances by ``get_dists``."""
    B, N, _ = xyz.shape
    centroids = torch.zeros(size=(B, M), dtype=torch.long)
    dists = torch.ones(B, N, dtype=xyz.dtype) * 1e5
    inds = torch.randint(0, N, size=(B,), dtype=torch.long)
    batch = torch.arange(0, B, dtype=torch.long)
    for i in range(M):
        centroids[:, i] = inds
        cur_point = xyz[batch, inds, :]
        cur_dist = torch.squeeze(get_dists(torch.unsqueeze(cur_point, 1), xyz), dim=1)
        dists[cur_dist < dists] = cur_dist[cur_dist < dists]
        inds = torch.max(dists, dim=1)[1]
    return centroids


def pct_knn_point(nsample, xyz, new_xyz):
    """model/pct_utils.py:98-109."""
    sqrdists = square_distance(new_xyz, xyz)
    _, group_idx = torch.topk(sqrdists, nsample, dim=-1, largest=False, sorted=False)
    return group_idx


# ---- the same four samplers with the arithmetic spelled out in C (same draws from the CPU generator)
def c_farthest_point_sample(xyz, npoint):
    """model/pointnet2_utils.py:63-84 (direct-form distances: nothing depends on a BLAS)."""
    return _C.fps_from_start(xyz, npoint, torch.randint(0, xyz.shape[1], (xyz.shape[0],), dtype=torch.long))


def c_query_ball_point(radius, nsample, xyz, new_xyz):
    """model/pointnet2_utils.py:87-107 on form-3 distances."""
    return _C.query_ball_point(radius, nsample, xyz, new_xyz, _C.FORM_SQUARE_DISTANCE)


def c_pct_fps(xyz, M):
    """util/other_utils.py:254-272 on form-4 distances."""
    return _C.fps_pct(xyz, M, torch.randint(0, xyz.shape[1], size=(xyz.shape[0],), dtype=torch.long))


def c_pct_knn_point(nsample, xyz, new_xyz):
    """model/pct_utils.py:98-109: the nsample smallest form-3 distances (ascending here; the reference's
    ``sorted=False`` order is unspecified and its consumer max-pools over the neighbours)."""
    return _C.knn_points(new_xyz, xyz, nsample, _C.FORM_SQUARE_DISTANCE)[1]


@contextlib.contextmanager
def cpu_geometry(torch_ops=False):
    """Inside: the product's PointNet++ / PCT modules sample and group through the functions above (CPU tensors);
    ``torch_ops`` selects the torch restatement instead of the C one."""
    from hit_adv_amd.model import pct as PCT
    from hit_adv_amd.model import pointnet2 as P2
    saved = (P2.farthest_point_sample, P2.query_ball_point, PCT.fps, PCT.knn_point)
    f_fps, f_ball, f_pfps, f_knn = ((farthest_point_sample, query_ball_point, pct_fps, pct_knn_point) if torch_ops else
                                    (c_farthest_point_sample, c_query_ball_point, c_pct_fps, c_pct_knn_point))
    P2.farthest_point_sample = lambda xyz, npoint: f_fps(xyz.detach(), npoint)
    P2.query_ball_point = lambda radius, nsample, xyz, new_xyz: f_ball(radius, nsample, xyz.detach(), new_xyz.detach())
    PCT.fps = lambda xyz, M: f_pfps(xyz.detach(), M)
    PCT.knn_point = lambda nsample, xyz, new_xyz: f_knn(nsample, xyz.detach(), new_xyz.detach())
    try:
        yield
    finally:
        P2.farthest_point_sample, P2.query_ball_point, PCT.fps, PCT.knn_point = saved


class CpuVictim(torch.nn.Module):
    """A PointNet++ / PCT module whose forward runs under ``cpu_geometry()`` -- what the CPU oracle attacks."""

    def __init__(self, model, torch_ops=False):
        super().__init__()
        self.model = model
        self.torch_ops = torch_ops

    def forward(self, x):
        with cpu_geometry(self.torch_ops):
            return self.model(x)
