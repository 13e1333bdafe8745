"""Stress test of the in-launch split hand-off (csrc/common.hpp::handoff_last_arriver): the K chunks of fc_layer and the
point splits of linear_max_fwd / linear_max_fwd_bf16x3 publish partials with write-through stores and the LAST workgroup
to draw a ticket merges them.  The protocol is the measured gfx950 form of MI355X_MICROARCH.md ("Valid forms", first
row), not something the HIP memory model promises, so it is exercised the way that guide asks: many launches, several
streams at once, uneven load from an HBM-bound kernel on a fourth stream, ragged shapes, every output word compared.
A stale partial or a lost ticket shows up as a bit that differs from the same operator run alone (fc_layer: the result
is a fixed-order sum, bitwise reproducible; linear_max_fwd: additionally against the two-launch merge)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _p(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def test_split_handoff_under_uneven_load_on_three_streams():
    from hit_adv_amd import _lib, ops
    lib = _lib.load()
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(0)
    B, Np, Cin, Cout = 32, 1000, 128, 1024  # 1000 points: the second split of every cloud is ragged
    x = torch.randn(B * Np, Cin, generator=g).relu().to(dev)
    Wt = (torch.randn(Cin, Cout, generator=g) * 0.1).to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    W3 = ops.split_weights_bf16x3(Wt.t().contiguous())
    fx = torch.randn(B, 1024, generator=g).to(dev)          # fc_layer: K = 1024 -> 8 chunks per tile
    fw = (torch.randn(1024, 512, generator=g) / 32).to(dev)
    fb = torch.randn(512, generator=g).to(dev)

    # references, each operator alone on an idle GPU
    ref_v, ref_i = ops.linear_max_fwd(x, Wt, B, Np, bias=bias, relu=True)
    ref3_v, ref3_i = ops.linear_max_fwd_bf16x3(x, W3, B, Np, bias=bias, relu=True)
    ref_fc = ops.fc_layer(fx, fw, fb, relu=True)
    n = lib.hitadv_linear_max_fwd_scratch(B, Np, Cout)
    pv, pi = torch.empty(n, device=dev), torch.empty(n, device=dev, dtype=torch.int32)
    two_v, two_i = torch.empty(B, Cout, device=dev), torch.empty(B, Cout, device=dev, dtype=torch.int64)
    rc = lib.hitadv_linear_max_fwd(_p(x), _p(Wt), _p(bias), B, Np, Cin, Cout, 1, _p(pv), _p(pi), _p(two_v), _p(two_i),
                                   _p(None), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(two_v, ref_v) and torch.equal(two_i, ref_i)  # in-launch merge == separate merge launch

    streams = [torch.cuda.Stream() for _ in range(3)]
    noise = torch.cuda.Stream()
    bad = [torch.zeros((), device=dev, dtype=torch.int64) for _ in streams]
    px, py = torch.randn(32, 1024, 3, device=dev), torch.randn(32, 1024, 3, device=dev)
    rounds = 1200
    torch.cuda.synchronize()
    for r in range(rounds):
        if r % 3 == 0:
            with torch.cuda.stream(noise):  # 135 MB of stores per launch: the memory system is busy, unevenly
                ops.pairwise_sqdist(px, py)
        for k, s in enumerate(streams):
            with torch.cuda.stream(s):
                order = (r + k) % 3
                for j in range(3):
                    which = (order + j) % 3
                    if which == 0:
                        v, i = ops.linear_max_fwd(x, Wt, B, Np, bias=bias, relu=True)
                        bad[k] += (v != ref_v).sum() + (i != ref_i).sum()
                    elif which == 1:
                        v, i = ops.linear_max_fwd_bf16x3(x, W3, B, Np, bias=bias, relu=True)
                        bad[k] += (v != ref3_v).sum() + (i != ref3_i).sum()
                    else:
                        o = ops.fc_layer(fx, fw, fb, relu=True)
                        bad[k] += (o != ref_fc).sum()
    torch.cuda.synchronize()
    assert [int(b) for b in bad] == [0, 0, 0]  # 3 x 1200 x 3 launches, every output word equal to the idle-GPU result
