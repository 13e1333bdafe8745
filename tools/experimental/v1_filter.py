"""Python wrappers of libhitadv_experimental.so (include/hitadv_experimental.h; `make -C hit_adv_amd/csrc experimental`): the filtered form
of PointNet's 128 -> 1024 layer + max over the points.  Nothing in hit_adv_amd/ imports this; tests/test_gpu_kernels.py::
test_filtered_linear_max_equals_the_full_evaluation and tools/v1_filter_{probe,ablate}.py do, and skip / stop when `available()` is False."""
import ctypes
import os

import torch

from hit_adv_amd.ops import _dev, _p, _stream

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB_PATH = os.path.join(ROOT, "hit_adv_amd", "libhitadv_experimental.so")
_P, _I = ctypes.c_void_p, ctypes.c_int
_lib = None


def available():
    return os.path.exists(LIB_PATH)


def load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(LIB_PATH)
        lib.hitadv_linear_max_filter_supported.argtypes = [_I] * 5
        lib.hitadv_linear_max_filter_scratch_words.argtypes = [_I, _I]
        lib.hitadv_linear_max_filter_scratch_words.restype = ctypes.c_int64
        lib.hitadv_linear_max_fwd_f16x2_filtered.argtypes = [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]
        _lib = lib
    return _lib


def linear_max_filter_supported(B, N, Cin, Cout, blocks=0):
    """Whether ``linear_max_fwd_f16x2_filtered`` takes this shape (include/hitadv.h)."""
    return bool(load().hitadv_linear_max_filter_supported(int(B), int(N), int(Cin), int(Cout), int(blocks)))


def weight_row_norms(Wr):
    """|Wr[c,:]|_2 per output channel with a 1e-5 allowance for its own rounding: the ``wnorm`` of the filtered layer."""
    return (Wr.double().norm(dim=1) * (1. + 1e-5)).float().contiguous()


def linear_max_fwd_f16x2_filtered(x, W2, wnorm, B, N, seed, bias=None, relu=False, blocks=0, range_flag=None, scratch=None):
    """``linear_max_fwd_f16x2(..., packed=True)`` with one fp16 product per value instead of three and the exact evaluation of
    the few points that can be the maximum (csrc/experimental/victim_filter.hip).  ``seed`` int64 [B,Cout] in / out: last call's arg-max
    table (any content is valid; the result does not depend on it, the time does); ``range_flag`` is REQUIRED: it is raised
    when a candidate list overflows, and the result is then invalid."""
    x = _dev(x, "x")
    _, Cout, Cin = W2.shape
    if range_flag is None:
        raise ValueError("linear_max_fwd_f16x2_filtered needs a range_flag: an overfull candidate list is reported there")
    n = load().hitadv_linear_max_filter_scratch_words(B, Cout)
    if scratch is None or scratch.numel() < n:
        scratch = torch.empty(n, device=x.device, dtype=torch.int32)
    out = torch.empty(B, Cout, device=x.device)
    idx = torch.empty(B, Cout, device=x.device, dtype=torch.int64)
    rc = load().hitadv_linear_max_fwd_f16x2_filtered(_p(x), _p(W2), _p(wnorm), _p(bias), B, N, Cin, Cout, 1 if relu else 0,
              int(blocks), _p(seed), _p(scratch), _p(out), _p(idx), _p(range_flag), _stream())
    if rc != 0:
        raise RuntimeError("hitadv_linear_max_fwd_f16x2_filtered failed: %d" % rc)
    return out, idx
