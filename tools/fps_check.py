#!/usr/bin/env python3
"""Diagnostic (round 5): `install()` replaces ops.fps_from_start by a version that computes every table with BOTH sampling kernels
-- fps_lean first (or the 64-bit-key kernel first with order='key64_first'), fps_lean again right behind it, the 64-bit-key
kernel, fps_lean once more -- and counts, on the device, the clouds whose tables differ.  `counts()` / `captures()` read the
results.  tools/fps_check_modes.py and tools/fps_check_cfg4.py drive it; docs/kernels/round5.md section 8 has what it found.
With HITADV_FPS_DIAG_LIB=<path of a libfps_diag_<n>.so built from tools/tune/fps_diag_lib.hip> the "fps_lean" launches run THAT library's
instrumented kernel (the packed-f32 build is -DHITADV_FPS_DIAG=2) instead of the product's: tools/fps_packed_repro.sh."""
import ctypes
import os

import torch

from hit_adv_amd import _lib, ops

_state = {}


def install(order='lean_first', sync=None):
    lib = _lib.load()
    _p, _dev, _stream = ops._p, ops._dev, ops._stream
    shipped = lib.hitadv_debug_fps_form(-1)  # (an invalid value changes nothing and returns the current form)
    diag = ctypes.CDLL(os.environ["HITADV_FPS_DIAG_LIB"]) if os.environ.get("HITADV_FPS_DIAG_LIB") else None
    _state['diag'] = diag

    def fps_from_start(xyz, npoint, start):
        xyz = _dev(xyz.detach(), "xyz")
        start = _dev(start, "start", torch.int64)
        B, N, _ = xyz.shape
        dev = xyz.device
        if sync == 'stream':
            torch.cuda.current_stream().synchronize()
        elif sync == 'device':
            torch.cuda.synchronize()
        snap_xyz, snap_start = xyz.clone(), start.clone()

        def launch(form):
            out = torch.empty(B, npoint, device=dev, dtype=torch.int64)
            if os.environ.get("HITADV_FPS_PREFILL"):  # what a wrong table is made of: values the kernel computed, or what the buffer held before
                out.fill_(-7)
            if form == 1 and diag is not None and 256 < N <= 4080:
                rc = diag.fpsdiag_fps_from_start(_p(xyz), _p(start), B, N, npoint, _p(out), _stream())
                assert rc == 0, rc
                return out
            lib.hitadv_debug_fps_form(form)
            _lib.call("hitadv_fps_from_start", _p(xyz), _p(start), B, N, npoint, _p(out), _stream())
            return out
        logging = diag is not None and bool(os.environ.get("HITADV_FPS_LOG")) and N == 2048  # (a HITADV_FPS_DIAG=9 build) [B, npoint, 8 waves] keys of the first launch
        if logging:
            log = torch.zeros(2, B, npoint, 8, dtype=torch.int64, device=dev)  # [keys | (centre x, active lanes, winner) of lane 63]
            diag.fpsdiag_log(ctypes.c_void_p(log.data_ptr()))
        idx = launch(0 if order == 'key64_first' else 1)
        if logging:
            diag.fpsdiag_log(None)
        second, chk, again = launch(1), launch(0), launch(1)
        lib.hitadv_debug_fps_form(shipped)
        c = _state.setdefault('tables', torch.zeros(2, dtype=torch.int64, device=dev))
        more = _state.setdefault('more', torch.zeros(3, dtype=torch.int64, device=dev))
        chg = _state.setdefault('inputs_changed', torch.zeros(2, dtype=torch.int64, device=dev))
        more[0] += (again != idx).any(dim=1).sum()   # first launch vs the last fps_lean launch
        more[1] += (again != chk).any(dim=1).sum()   # the last fps_lean launch vs the 64-bit-key kernel
        more[2] += (second != chk).any(dim=1).sum()  # the launch right behind the first vs the 64-bit-key kernel
        chg[0] += (xyz != snap_xyz).any().to(torch.int64)
        chg[1] += (start != snap_start).any().to(torch.int64)
        bad = (idx != chk).any(dim=1)
        cap = _state.setdefault((N, npoint), dict(xyz=torch.zeros(N, 3, device=dev), start=torch.zeros((), dtype=torch.int64, device=dev),
                                                  first=torch.zeros(npoint, dtype=torch.int64, device=dev),
                                                  key64=torch.zeros(npoint, dtype=torch.int64, device=dev),
                                                  have=torch.zeros((), dtype=torch.bool, device=dev)))
        at = bad.to(torch.int64).argmax()
        take = bad.any() & ~cap['have']
        cap['xyz'].copy_(torch.where(take, xyz[at], cap['xyz']))
        cap['start'].copy_(torch.where(take, start[at], cap['start']))
        cap['first'].copy_(torch.where(take, idx[at], cap['first']))
        cap['key64'].copy_(torch.where(take, chk[at], cap['key64']))
        if logging:
            cl = cap.setdefault('log', torch.zeros(2, npoint, 8, dtype=torch.int64, device=dev))
            cl.copy_(torch.where(take, log[:, at], cl))
        cap['have'].logical_or_(take)
        c[0] += bad.sum()
        c[1] += B
        # is a wrong table the RIGHT table of the same stream's previous pass (the same buffer address, the cloud one Adam step older)?
        hist = _state.setdefault(('hist', torch.cuda.current_stream().cuda_stream, N, npoint), dict(prev=torch.full((B, npoint), -1, dtype=torch.int64, device=dev)))
        old = _state.setdefault('wrong_equals_previous_pass', torch.zeros(2, dtype=torch.int64, device=dev))
        same_as_prev = (idx == hist['prev']).all(dim=1)
        old[0] += (bad & same_as_prev).sum()
        old[1] += bad.sum()
        hist['prev'].copy_(chk)
        pre = _state.setdefault('prefill_seen', torch.zeros(1, dtype=torch.int64, device=dev))
        pre[0] += (idx == -7).sum()
        return idx
    ops.fps_from_start = fps_from_start


def reset():
    for k, v in _state.items():
        if isinstance(v, torch.Tensor):
            v.zero_()
        elif isinstance(k, tuple) and k[0] == 'hist':
            v['prev'].fill_(-1)


def diag_counters():
    """The instrumented kernel's own counters (HITADV_FPS_DIAG 4 / 5 builds), or None without a diagnostic library."""
    if _state.get('diag') is None:
        return None
    buf = (ctypes.c_uint * 8)()
    _state['diag'].fpsdiag_counters(buf)
    return list(buf)


def counts():
    """{'tables': [first launch != 64-bit-key kernel, clouds], 'more': [first != last lean, last lean != key64, second != key64],
    'inputs_changed': [calls during which the cloud changed, ... the start indices]}"""
    return {k: v.tolist() for k, v in _state.items() if isinstance(v, torch.Tensor)}


def captures():
    return {k: {n: t.cpu() for n, t in v.items()} for k, v in _state.items() if isinstance(k, tuple) and k[0] != 'hist'}
