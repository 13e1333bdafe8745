"""Python face of the nine native entry points, with the names, argument order and argument
checks of the reference's pybind module ``pointnet2_ops._ext`` (_ext-src/src/bindings.cpp:6-19,
CHECK_* macros of include/utils.h:5-25): CUDA, contiguous, float32 / int32 -- anything else raises
RuntimeError, as AT_ASSERT does.  A failed launch raises too (the reference exit()s).
"""
import ctypes

import torch

from .. import _lib
from ..ops import _p, _stream


def _chk(t, name, dtype):
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor (CPU not supported)" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be a contiguous tensor" % name)
    if t.dtype != dtype:
        raise RuntimeError("%s must be a %s tensor" % (name, "float" if dtype == torch.float32 else "int"))


def furthest_point_sampling(points, nsamples):
    _chk(points, "points", torch.float32)
    B, N, _ = points.shape
    out = torch.zeros(B, nsamples, device=points.device, dtype=torch.int32)
    if nsamples > 0:
        _lib.call("hitadv_furthest_point_sampling", B, N, nsamples, _p(points), _p(None), _p(out), _stream())
    return out


def gather_points(points, idx):
    _chk(points, "points", torch.float32)
    _chk(idx, "idx", torch.int32)
    B, C, N = points.shape
    m = idx.shape[1]
    out = torch.zeros(B, C, m, device=points.device)
    _lib.call("hitadv_gather_points", B, C, N, m, _p(points), _p(idx), _p(out), _stream())
    return out


def gather_points_grad(grad_out, idx, n):
    _chk(grad_out, "grad_out", torch.float32)
    _chk(idx, "idx", torch.int32)
    B, C, m = grad_out.shape
    out = torch.zeros(B, C, n, device=grad_out.device)
    _lib.call("hitadv_gather_points_grad", B, C, n, m, _p(grad_out), _p(idx), _p(out), _stream())
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    _chk(new_xyz, "new_xyz", torch.float32)
    _chk(xyz, "xyz", torch.float32)
    B, m, _ = new_xyz.shape
    n = xyz.shape[1]
    out = torch.zeros(B, m, nsample, device=xyz.device, dtype=torch.int32)
    _lib.call("hitadv_query_ball_point", B, n, m, ctypes.c_float(radius), nsample, _p(new_xyz), _p(xyz),
              _p(out), _stream())
    return out


def group_points(points, idx):
    _chk(points, "points", torch.float32)
    _chk(idx, "idx", torch.int32)
    B, C, N = points.shape
    _, npts, ns = idx.shape
    out = torch.zeros(B, C, npts, ns, device=points.device)
    _lib.call("hitadv_group_points", B, C, N, npts, ns, _p(points), _p(idx), _p(out), _stream())
    return out


def group_points_grad(grad_out, idx, n):
    _chk(grad_out, "grad_out", torch.float32)
    _chk(idx, "idx", torch.int32)
    B, C, npts, ns = grad_out.shape
    out = torch.zeros(B, C, n, device=grad_out.device)
    _lib.call("hitadv_group_points_grad", B, C, n, npts, ns, _p(grad_out), _p(idx), _p(out), _stream())
    return out


def three_nn(unknowns, knows):
    _chk(unknowns, "unknowns", torch.float32)
    _chk(knows, "knows", torch.float32)
    B, n, _ = unknowns.shape
    m = knows.shape[1]
    dist2 = torch.zeros(B, n, 3, device=unknowns.device)
    idx = torch.zeros(B, n, 3, device=unknowns.device, dtype=torch.int32)
    _lib.call("hitadv_three_nn", B, n, m, _p(unknowns), _p(knows), _p(dist2), _p(idx), _stream())
    return [dist2, idx]


def three_interpolate(points, idx, weight):
    _chk(points, "points", torch.float32)
    _chk(idx, "idx", torch.int32)
    _chk(weight, "weight", torch.float32)
    B, C, m = points.shape
    n = idx.shape[1]
    out = torch.zeros(B, C, n, device=points.device)
    _lib.call("hitadv_three_interpolate", B, C, m, n, _p(points), _p(idx), _p(weight), _p(out), _stream())
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    _chk(grad_out, "grad_out", torch.float32)
    _chk(idx, "idx", torch.int32)
    _chk(weight, "weight", torch.float32)
    B, C, n = grad_out.shape
    out = torch.zeros(B, C, m, device=grad_out.device)
    _lib.call("hitadv_three_interpolate_grad", B, C, n, m, _p(grad_out), _p(idx), _p(weight), _p(out), _stream())
    return out
