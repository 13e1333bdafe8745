"""Import the *unmodified* Python reference (/root/reference) on a CPU-only box.

This file is fixture tooling, not product code and not a test: it only runs in
the build container (where /root/reference is mounted) and is used by
``make_golden.py`` to capture input/output vectors from the reference itself.
Nothing of the reference's source is copied; its modules are imported in place
after the third-party packages it needs but that are absent here have been
replaced by stand-ins in ``sys.modules``:

* ``pytorch3d.ops.knn_points / knn_gather`` -- pytorch3d==0.7.2 is pinned in the
  reference's requirements.txt:10 but is not vendored, so its arithmetic is
  **parity-unpinned**.  The stand-in below *defines* the canonical semantics
  used throughout this project: fp32 direct difference, evaluated as
  ``((dx*dx + dy*dy) + dz*dz)`` with one rounding per operation (no FMA),
  K smallest in ascending order, ties -> lower index (stable sort).
  Fixtures whose values passed through this stand-in (everything that calls ``knn_points`` in the reference):
  g2 (kNN tables, kappa, kappa-std, CurvStdDist), g4 (centre selection), g5 / g5b (HiT-ADV trajectories: scoring and
  centre selection), g22 (CurvDist).  g1, g3, g6-g21, g23, g24 never reach it.  The rule is stated independently in
  tests/test_gpu_kernels.py::test_knn_points_vs_independent_float64_top_k (float64 brute force on well-separated
  neighbours + the explicit tie policy).
* ``mayavi``, ``open3d``, ``torchvision``, ``seaborn``: GUI / unused imports.
* ``pointnet2_ops_lib...pointnet2_utils``: the CUDA extension (cannot be built
  without nvcc / run without a GPU) -- placeholder only so ``util.other_utils``
  imports; nothing captured here calls it.
* ``Tensor.cuda`` / ``Module.cuda`` become identities (no GPU here); ``Tensor.cpu`` returns a COPY, as a
  device->host transfer does -- on a CPU-only box ``t.cpu().numpy()`` would otherwise alias ``t`` and the
  reference's ``input_val`` snapshots (CW/Perturb.py:125) would silently follow later in-place Adam steps.
"""
import collections
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("HITADV_REFERENCE_ROOT", "/root/reference")


def canonical_knn_points(p1, p2, lengths1=None, lengths2=None, norm=2, K=1,
                         version=-1, return_nn=False, return_sorted=True):
    """Stand-in for pytorch3d.ops.knn_points with this project's canonical rule."""
    assert lengths1 is None and lengths2 is None and norm == 2
    p1 = p1.float()
    p2 = p2.float()
    D = p1.shape[-1]
    acc = None
    for d in range(D):
        diff = p1[:, :, None, d] - p2[:, None, :, d]
        sq = diff * diff
        acc = sq if acc is None else acc + sq
    order = torch.sort(acc, dim=-1, stable=True)
    dists = order.values[..., :K].contiguous()
    idx = order.indices[..., :K].contiguous()
    nn = canonical_knn_gather(p2, idx) if return_nn else None
    return _KNN(dists=dists, idx=idx, knn=nn)


def canonical_knn_gather(x, idx, lengths=None):
    B, M, U = x.shape
    _, N, K = idx.shape
    flat = idx.reshape(B, N * K, 1).expand(B, N * K, U)
    return torch.gather(x, 1, flat).reshape(B, N, K, U)


_KNN = collections.namedtuple("KNN", "dists idx knn")


def _module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    """Install the stand-ins and put the reference on sys.path. Idempotent."""
    if getattr(install, "_done", False):
        return
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True

    p3d = _module("pytorch3d")
    p3d.ops = _module("pytorch3d.ops", knn_points=canonical_knn_points,
                      knn_gather=canonical_knn_gather)
    p3d.loss = _module("pytorch3d.loss", chamfer_distance=None)
    may = _module("mayavi")
    may.mlab = _module("mayavi.mlab")
    _module("open3d")
    tv = _module("torchvision")
    tv.models = _module("torchvision.models")
    _module("seaborn", set=lambda *a, **k: None)
    _module("pointnet2_ops_lib")
    _module("pointnet2_ops_lib.pointnet2_ops")
    _module("pointnet2_ops_lib.pointnet2_ops.pointnet2_utils")
    sys.modules["pointnet2_ops_lib"].pointnet2_ops = sys.modules["pointnet2_ops_lib.pointnet2_ops"]
    sys.modules["pointnet2_ops_lib.pointnet2_ops"].pointnet2_utils = \
        sys.modules["pointnet2_ops_lib.pointnet2_ops.pointnet2_utils"]

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.Tensor.cpu = lambda self, *a, **k: self.detach().clone() if not self.requires_grad else self.clone()
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None

    sys.path.insert(0, REFERENCE_ROOT)
    install._done = True
