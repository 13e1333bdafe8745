"""CPU ORACLE loader -- test infrastructure, NOT product code.

ctypes front-end to ``liboracle_natives.so`` (built from pointnet2_oracle.c by
``oracle/Makefile``).  Function names and argument order mirror the pybind
module of the reference (pointnet2_ops/_ext-src/src/bindings.cpp:6-19) so tests
read like calls into the original extension.  Tensors are CPU torch tensors.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle_natives.so")
    src = os.path.join(_HERE, "pointnet2_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liboracle_natives.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _f(t):
    t = t.detach().contiguous().float()
    return t, ctypes.c_void_p(t.data_ptr())


def _i32(t):
    t = t.detach().contiguous().to(torch.int32)
    return t, ctypes.c_void_p(t.data_ptr())


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def furthest_point_sampling(points, nsamples):
    """points f32[B,N,3] -> i32[B,nsamples]   (src/sampling.cpp:66-87)"""
    pts, pp = _f(points)
    B, N, _ = pts.shape
    tmp = torch.full((B, N), 1e10, dtype=torch.float32)
    out = torch.zeros(B, nsamples, dtype=torch.int32)
    lib().oracle_fps_ext(B, N, nsamples, pp, _p(tmp), _p(out))
    return out


def gather_points(points, idx):
    """points f32[B,C,N], idx i32[B,m] -> f32[B,C,m]"""
    pts, pp = _f(points)
    ix, ip = _i32(idx)
    B, C, N = pts.shape
    m = ix.shape[1]
    out = torch.zeros(B, C, m)
    lib().oracle_gather_points(B, C, N, m, pp, ip, _p(out))
    return out


def gather_points_grad(grad_out, idx, n):
    g, gp = _f(grad_out)
    ix, ip = _i32(idx)
    B, C, m = g.shape
    out = torch.zeros(B, C, n)
    lib().oracle_gather_points_grad(B, C, n, m, gp, ip, _p(out))
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    """new_xyz f32[B,m,3], xyz f32[B,n,3] -> i32[B,m,nsample]"""
    q, qp = _f(new_xyz)
    p, pp = _f(xyz)
    B, m, _ = q.shape
    n = p.shape[1]
    out = torch.zeros(B, m, nsample, dtype=torch.int32)
    lib().oracle_ball_query(B, n, m, ctypes.c_float(radius), nsample, qp, pp, _p(out))
    return out


def group_points(points, idx):
    """points f32[B,C,N], idx i32[B,np,ns] -> f32[B,C,np,ns]"""
    pts, pp = _f(points)
    ix, ip = _i32(idx)
    B, C, N = pts.shape
    _, npts, ns = ix.shape
    out = torch.zeros(B, C, npts, ns)
    lib().oracle_group_points(B, C, N, npts, ns, pp, ip, _p(out))
    return out


def group_points_grad(grad_out, idx, n):
    g, gp = _f(grad_out)
    ix, ip = _i32(idx)
    B, C, npts, ns = g.shape
    out = torch.zeros(B, C, n)
    lib().oracle_group_points_grad(B, C, n, npts, ns, gp, ip, _p(out))
    return out


def three_nn(unknown, known):
    """-> (dist2 f32[B,n,3], idx i32[B,n,3])"""
    u, up = _f(unknown)
    k, kp = _f(known)
    B, n, _ = u.shape
    m = k.shape[1]
    d = torch.zeros(B, n, 3)
    ix = torch.zeros(B, n, 3, dtype=torch.int32)
    lib().oracle_three_nn(B, n, m, up, kp, _p(d), _p(ix))
    return d, ix


def three_interpolate(points, idx, weight):
    pts, pp = _f(points)
    ix, ip = _i32(idx)
    w, wp = _f(weight)
    B, C, m = pts.shape
    n = ix.shape[1]
    out = torch.zeros(B, C, n)
    lib().oracle_three_interpolate(B, C, m, n, pp, ip, wp, _p(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    g, gp = _f(grad_out)
    ix, ip = _i32(idx)
    w, wp = _f(weight)
    B, C, n = g.shape
    out = torch.zeros(B, C, m)
    lib().oracle_three_interpolate_grad(B, C, n, m, gp, ip, wp, _p(out))
    return out


FORM_DIRECT, FORM_GRAM, FORM_GRAM_KNN, FORM_SQUARE_DISTANCE, FORM_PCT_DISTS = 0, 1, 2, 3, 4  # pair_value() in pointnet2_oracle.c


def pairwise(x, y, form=FORM_DIRECT):
    """P[B,N,M] with every entry evaluated in the given form (form 1 / 2 = the reference's own fp32 arithmetic)."""
    a, ap = _f(x)
    b, bp = _f(y)
    B, N, _ = a.shape
    M = b.shape[1]
    P = torch.zeros(B, N, M)
    lib().oracle_pairwise_form(B, N, M, form, ap, bp, _p(P))
    return P


def knn_points(p1, p2, K, form=FORM_DIRECT):
    """Canonical kNN -> (dists f32[B,N,K], idx i64[B,N,K]); ascending, ties -> lower index."""
    q, qp = _f(p1)
    p, pp = _f(p2)
    B, N, _ = q.shape
    M = p.shape[1]
    d = torch.zeros(B, N, K)
    ix = torch.zeros(B, N, K, dtype=torch.int64)
    lib().oracle_knn_points_form(B, N, M, K, form, qp, pp, _p(d), _p(ix))
    return d, ix


def nn_min(x, y, form=FORM_DIRECT):
    """-> (min_j |x_i-y_j|^2 f32[B,N], argmin i32[B,N], lowest j on ties)"""
    a, ap = _f(x)
    b, bp = _f(y)
    B, N, _ = a.shape
    M = b.shape[1]
    d = torch.zeros(B, N)
    ix = torch.zeros(B, N, dtype=torch.int32)
    lib().oracle_nn_min_form(B, N, M, form, ap, bp, _p(d), _p(ix))
    return d, ix


def fps_from_start(xyz, npoint, start):
    p, pp = _f(xyz)
    s = start.detach().contiguous().to(torch.int64)
    B, N, _ = p.shape
    out = torch.zeros(B, npoint, dtype=torch.int64)
    lib().oracle_fps_from_start(B, N, npoint, pp, _p(s), _p(out))
    return out


def fps_pct(xyz, npoint, start):
    """PCT's sampler (util/other_utils.py:254-272) from given first indices -> i64[B,npoint]."""
    p, pp = _f(xyz)
    s = start.detach().contiguous().to(torch.int64)
    B, N, _ = p.shape
    out = torch.zeros(B, npoint, dtype=torch.int64)
    lib().oracle_fps_pct(B, N, npoint, pp, _p(s), _p(out))
    return out


def radius_squared(radius):
    """``sqrdists > radius ** 2`` compares an fp32 tensor with a Python double: torch rounds the scalar to fp32."""
    return float(np.float32(float(radius) ** 2))


def query_ball_point(radius, nsample, xyz, new_xyz, form=FORM_SQUARE_DISTANCE):
    """The victims' query_ball_point (model/pointnet2_utils.py:87-107), argument order as there -> i64[B,S,nsample]."""
    q, qp = _f(new_xyz)
    p, pp = _f(xyz)
    B, m, _ = q.shape
    n = p.shape[1]
    out = torch.zeros(B, m, nsample, dtype=torch.int64)
    lib().oracle_query_ball_form(B, n, m, ctypes.c_float(radius_squared(radius)), nsample, form, qp, pp, _p(out))
    return out
