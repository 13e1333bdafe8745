"""CPU restatement of the geometric functions inside the reference's PointNet++ and PCT victims (pure torch, op for op).

TEST INFRASTRUCTURE, like everything under oracle/: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import it.  The product runs these steps in HIP (hit_adv_amd/model/pointnet2.py, pct.py -> libhitadv_hip.so) and
raises on CPU tensors; this file is what lets the checker run the same victims on the CPU.

Pinned: with these functions in place of the HIP ones, the victims' plain nn.Module forward reproduces fixtures g11
(PointNet++: FPS table, ball-query table, logits, input gradient) and g12 (PCT: FPS table, logits, input gradient), both
captured from the unmodified reference (tests/test_victims_cpu.py).

``cpu_geometry()`` swaps them into the product's modules for the duration of a ``with`` block.
"""
import contextlib

import torch


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


def get_dists(points1, points2):
    """util/other_utils.py:237-251: sqrt of the clamped Gram form."""
    B, M, _ = points1.shape
    _, N, _ = points2.shape
    dists = torch.sum(torch.pow(points1, 2), dim=-1).view(B, M, 1) + torch.sum(torch.pow(points2, 2), dim=-1).view(B, 1, N)
    dists -= 2 * torch.matmul(points1, points2.permute(0, 2, 1))
    dists = torch.where(dists < 0, torch.ones_like(dists) * 1e-7, dists)
    return torch.sqrt(dists).to(points1.dtype)


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


@contextlib.contextmanager
def cpu_geometry():
    """Inside: the product's PointNet++ / PCT modules sample and group through the functions above (CPU tensors)."""
    from hit_adv_amd.model import pct as PCT
    from hit_adv_amd.model import pointnet2 as P2
    saved = (P2.farthest_point_sample, P2.query_ball_point, PCT.fps, PCT.knn_point)
    P2.farthest_point_sample = lambda xyz, npoint: farthest_point_sample(xyz.detach(), npoint)
    P2.query_ball_point = lambda radius, nsample, xyz, new_xyz: query_ball_point(radius, nsample, xyz.detach(), new_xyz.detach())
    PCT.fps = lambda xyz, M: pct_fps(xyz.detach(), M)
    PCT.knn_point = lambda nsample, xyz, new_xyz: pct_knn_point(nsample, xyz.detach(), new_xyz.detach())
    try:
        yield
    finally:
        P2.farthest_point_sample, P2.query_ball_point, PCT.fps, PCT.knn_point = saved


class CpuVictim(torch.nn.Module):
    """A PointNet++ / PCT module whose forward runs under ``cpu_geometry()`` -- what the CPU oracle attacks."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, x):
        with cpu_geometry():
            return self.model(x)
