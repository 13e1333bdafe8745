"""pytest plugin: run `-m gpu` parity tests of the NON-matrix kernels on the CPU wave emulator, test bodies unchanged.

    python -m pytest -p emu_plugin --emulate tests/test_gpu_kernels.py -k "pairwise or nn_min or knn_points or fps ..."   (PYTHONPATH=tests/native)

What it does, all of it test-side (the product is untouched and still refuses CPU tensors on its own):
  * builds hit_adv_amd/csrc/{pairwise,knn,sampling,grouping,deform,regulariser,attack_state,iteration}.hip for the emulator
    -- and victim_bf3.hip, whose 16x16x32 matrix instructions are emulated with a summation order of the emulator's own --
    (tests/native/emu_build.py) and puts a dispatcher over those libraries where hit_adv_amd._lib keeps the loaded libhitadv_hip.so;
  * lets `ops._dev` / `_ext._chk` accept CPU tensors and hands the kernels a null stream;
  * maps what the tests say about devices onto the CPU: `.cuda()`, `.to('cuda')`, `device='cuda'` (a TorchFunctionMode), a few
    `torch.cuda.*` calls (current_stream, synchronize, is_available).
A test that reaches a matrix-instruction kernel aborts the process (MFMA is not emulated): select tests with -k."""
import contextlib
import ctypes
import os
import sys
import tempfile

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests", "native"))
STEMS = ["pairwise", "knn", "sampling", "grouping", "deform", "regulariser", "attack_state", "iteration", "victim_bf3", "victim", "pointnet",
         "bmm", "attention", "rows_linear", "group_mlp", "gemm16"]  # (all of csrc/; gemm16's LDS-DMA ring kernel is switched off below)
if os.environ.get("HITADV_EMU_STEMS"):  # a subset builds faster (tests/test_emulated_gpu_subset.py: the non-matrix files only)
    STEMS = [s_ for s_ in os.environ["HITADV_EMU_STEMS"].split(",") if s_]
os.environ["HITADV_EMULATE"] = "1"  # (tests may pick emulator-sized shapes: a matrix instruction costs milliseconds here)


def pytest_addoption(parser):
    parser.addoption("--emulate", action="store_true", help="run gpu-marked tests on the CPU wave emulator (non-matrix kernels only)")


class EmuLib:
    def __init__(self, libs, prototypes, restype):
        self._libs, self._proto, self._res = libs, prototypes, restype

    def __getattr__(self, name):
        for lib in self._libs:
            try:
                fn = getattr(lib, name)
            except AttributeError:
                continue
            if name in self._proto:
                fn.argtypes = self._proto[name]
                fn.restype = self._res.get(name, ctypes.c_int)
            setattr(self, name, fn)
            return fn
        raise AttributeError("%s is not among the emulated kernel files (%s): a matrix-instruction kernel?" % (name, ", ".join(STEMS)))


def _is_cuda(d):
    return d is not None and str(d).startswith("cuda")


_POISON = os.environ.get("HITADV_EMU_POISON") == "1"
_IS_CUDA = torch.Tensor.is_cuda  # (the getset descriptor: attribute reads arrive as its __get__)


class CpuForCuda(torch.overrides.TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if _is_cuda(kwargs.get("device")):
            kwargs["device"] = "cpu"
        if getattr(func, "__self__", None) is _IS_CUDA:  # the product's model code picks its engine path by `x.is_cuda`
            return True
        name = getattr(func, "__name__", "")
        if _POISON and name in ("empty", "empty_like", "new_empty", "empty_strided"):
            # HITADV_EMU_POISON=1: uninitialised allocations come back full of NaNs (floats) / a large pattern (integers), so that a kernel
            # whose result depends on memory nobody wrote shows it deterministically (malloc's garbage is usually finite: garbage x 0 = 0)
            t = func(*args, **kwargs)
            if t.is_floating_point():
                t.fill_(float("nan"))
            elif t.dtype in (torch.int32, torch.int64, torch.int16, torch.uint8, torch.int8):
                t.fill_(0x5a5a5a5a if t.dtype in (torch.int32, torch.int64) else 0x5a)
            return t
        if name == "cuda" and args and torch.is_tensor(args[0]):
            return args[0].clone()  # a host-to-device copy is a NEW tensor (tests rely on it: x.cuda().requires_grad_() must not touch x)
        if name == "to" and args and torch.is_tensor(args[0]) and any(isinstance(a, (str, torch.device)) and _is_cuda(a) for a in args):
            args = tuple("cpu" if (isinstance(a, (str, torch.device)) and _is_cuda(a)) else a for a in args)
            return func(*args, **kwargs).clone()
        return func(*args, **kwargs)


class _Stream:
    cuda_stream = 0

    def synchronize(self):
        pass

    def wait_stream(self, other):
        pass


def pytest_configure(config):
    if not config.getoption("--emulate"):
        return
    import emu_build
    from hit_adv_amd import _lib, ops
    out = tempfile.mkdtemp(prefix="hitadv_emu_")
    libs = [ctypes.CDLL(emu_build.build(s, out)) for s in STEMS]
    emu = EmuLib(libs, _lib.PROTOTYPES, _lib._RESTYPE)
    _lib._lib = emu
    if "gemm16" in STEMS:
        emu.hitadv_debug_g16_ring(0)  # the staged G16 kernel only: the ring kernel's LDS-DMA is not emulated
    _lib.load = lambda: emu

    def dev(t, name, dtype=torch.float32):
        if not torch.is_tensor(t):
            raise TypeError("%s must be a tensor" % name)
        return (t.to(dtype) if t.dtype != dtype else t).contiguous()
    ops._dev = dev
    ops._stream = lambda: ctypes.c_void_p(0)
    # `_p(g.contiguous())` hands a kernel the address of a TEMPORARY: fine on the GPU, where the caching allocator's reuse of a freed block
    # is ordered behind the launch on the same stream; on the host the block goes back to malloc before the "kernel" has run.  Keep the
    # last few hundred tensors whose addresses were taken alive (round 6 found this as NaNs that changed from run to run in PointNet++'s
    # gradient under emulation, and nowhere when the node was replayed alone).
    keep = []

    def p_keepalive(t):
        if t is None:
            return ctypes.c_void_p(0)
        keep.append(t)
        if len(keep) > 512:
            del keep[:256]
        return ctypes.c_void_p(t.data_ptr())
    ops._p = p_keepalive
    from hit_adv_amd.pointnet2_ops import _ext
    _ext._stream = ops._stream
    _ext._p = ops._p

    def chk(t, name, dtype):
        if not t.is_contiguous():
            raise RuntimeError("%s must be a contiguous tensor" % name)
        if t.dtype != dtype:
            raise RuntimeError("%s must be a %s tensor" % (name, "float" if dtype == torch.float32 else "int"))
    _ext._chk = chk
    torch.cuda.is_available = lambda: True
    # (torch.Generator('cuda') is NOT mapped: torch's own modules use the class in type expressions; the one test that draws on the device
    #  -- test_dgcnn_attack_view_and_edge_max_kernels -- needs a hipGraph further down anyway)
    torch.cuda.current_stream = lambda *a, **k: _Stream()
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.is_current_stream_capturing = lambda: False  # (torch.optim asks once it believes a GPU is there)
    torch.cuda.current_device = lambda: 0
    torch.cuda.device_count = lambda: 1
    torch.cuda.get_sync_debug_mode = lambda: 0
    torch.cuda.set_sync_debug_mode = lambda mode: None
    torch.cuda.empty_cache = lambda: None
    torch.cuda.Stream = lambda *a, **k: _Stream()
    torch.cuda.stream = lambda s: contextlib.nullcontext()     # one OS thread: "streams" run in program order
    config._emu_mode = CpuForCuda()
    config._emu_mode.__enter__()


def pytest_unconfigure(config):
    mode = getattr(config, "_emu_mode", None)
    if mode is not None:
        mode.__exit__(None, None, None)
