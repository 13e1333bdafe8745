"""Replacement for the two pytorch3d.ops functions the reference imports
(ShapeAttack/HiT_ADV.py:9, util/dist_utils.py:12, FGM/GeoA3_args.py): ``knn_points`` and
``knn_gather``, same call signature and return type for the arguments the reference uses.

pytorch3d (==0.7.2, requirements.txt:10) is not vendored by the reference, so its exact tie
and rounding behaviour is parity-unpinned; this module's rule is: fp32 direct-difference squared
distance ((dx*dx+dy*dy)+dz*dz), K smallest in ascending order, ties -> lower index.
"""
from collections import namedtuple

from . import ops

_KNN = namedtuple("KNN", "dists idx knn")


def knn_points(p1, p2, lengths1=None, lengths2=None, norm=2, K=1, version=-1, return_nn=False,
               return_sorted=True):
    """p1[B,N,3], p2[B,M,3] -> KNN(dists[B,N,K], idx[B,N,K] int64, knn[B,N,K,3] or None).
    Differentiable w.r.t. both point sets through ``dists`` (as in pytorch3d)."""
    if lengths1 is not None or lengths2 is not None:
        raise NotImplementedError("ragged batches (lengths1/lengths2) are not used by the reference path")
    if norm != 2:
        raise NotImplementedError("only the squared-L2 metric (norm=2) is implemented")
    dists, idx = ops.KnnPoints.apply(p1, p2, int(K))
    nn = knn_gather(p2, idx) if return_nn else None
    return _KNN(dists=dists, idx=idx, knn=nn)


def knn_gather(x, idx, lengths=None):
    """x[B,M,U], idx[B,N,K] -> [B,N,K,U] with out[b,n,k] = x[b, idx[b,n,k]]."""
    if lengths is not None:
        raise NotImplementedError("ragged batches are not used by the reference path")
    B, M, U = x.shape
    _, N, K = idx.shape
    flat = idx.reshape(B, N * K, 1).expand(B, N * K, U)
    return x.gather(1, flat).reshape(B, N, K, U)
