#!/usr/bin/env python3
"""Diagnostic (round 5): `install()` replaces ops.fps_from_start by a version that computes every table with BOTH sampling kernels
-- fps_lean first (or the 64-bit-key kernel first with order='key64_first'), fps_lean again right behind it, the 64-bit-key
kernel, fps_lean once more -- and counts, on the device, the clouds whose tables differ.  `counts()` / `captures()` read the
results.  tools/fps_check_modes.py and tools/fps_check_cfg4.py drive it; docs/kernels/round5.md section 8 has what it found."""
import os

import torch

from hit_adv_amd import _lib, ops

_state = {}


def install(order='lean_first', sync=None):
    lib = _lib.load()
    _p, _dev, _stream = ops._p, ops._dev, ops._stream

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
            lib.hitadv_debug_fps_form(form)
            _lib.call("hitadv_fps_from_start", _p(xyz), _p(start), B, N, npoint, _p(out), _stream())
            return out
        idx = launch(0 if order == 'key64_first' else 1)
        second, chk, again = launch(1), launch(0), launch(1)
        lib.hitadv_debug_fps_form(0)  # the shipped form
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
        cap['have'].logical_or_(take)
        c[0] += bad.sum()
        c[1] += B
        return idx
    ops.fps_from_start = fps_from_start


def reset():
    for v in _state.values():
        if isinstance(v, torch.Tensor):
            v.zero_()


def counts():
    """{'tables': [first launch != 64-bit-key kernel, clouds], 'more': [first != last lean, last lean != key64, second != key64],
    'inputs_changed': [calls during which the cloud changed, ... the start indices]}"""
    return {k: v.tolist() for k, v in _state.items() if isinstance(v, torch.Tensor)}


def captures():
    return {k: {n: t.cpu() for n, t in v.items()} for k, v in _state.items() if isinstance(k, tuple)}
