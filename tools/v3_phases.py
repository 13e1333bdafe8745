#!/usr/bin/env python3
"""Where the time of the PointNet engine's backward kernels (rowmlp_bwd16_k in the default fp16x2 mode, rowmlp_bwd_k otherwise; stages 2 / 1 / 0) goes: a diagnostic build of the
library with wall-clock stamps at the phase boundaries of every block.

    tools/v3_phases.py --build      # here (cross-compiles tools/build/libhitadv_hip_stamps.so with -DHITADV_STAMPS)
    gpurun -- python tools/v3_phases.py [B]     # B clouds in the pass (default 32; 128 = one stack of four attacks)
"""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'tools', 'build', 'libhitadv_hip_stamps.so')
PHASES = ['arg-max table, ballot', 'list build', 'gather (MFMA)', 'operand requests, tile store', 'ReLU mask, 128->64 product',
          '64->64 (stage 1), barrier', '64->3 and outputs']


def build():
    src = os.path.join(ROOT, 'hit_adv_amd', 'csrc')
    objs = [os.path.join(src, f) for f in os.listdir(src) if f.endswith('.o') and f != 'pointnet.o']
    subprocess.check_call(['make', '-C', src])
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-slp-vectorize',
                           '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'), '-DHITADV_STAMPS', '-c',
                           os.path.join(src, 'pointnet.hip'), '-o', '/tmp/pointnet_stamps.o'])
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, '/tmp/pointnet_stamps.o'] + objs)
    print('built', LIB)


def main():
    if '--build' in sys.argv:
        return build()
    os.environ['HITADV_LIBRARY'] = LIB
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from hit_adv_amd import _lib
    from hit_adv_amd.Dataset.synthetic import synth_batch
    from hit_adv_amd.model.pointnet import PointNetFeatureModel
    torch.manual_seed(0)
    model = PointNetFeatureModel(40, normal_channel=False).cuda().eval()
    view = model.attack_view()
    B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 32
    data, _ = synth_batch(B, 1024)
    x = data[:, :, :3].transpose(1, 2).contiguous().cuda().requires_grad_()
    for _ in range(3):
        out = view(x)
        logits = out[0] if isinstance(out, tuple) else out
        g, = torch.autograd.grad(logits.logsumexp(1).sum(), x)
    torch.cuda.synchronize()
    n = 3 * 1024 * 8
    host = (ctypes.c_ulonglong * n)()
    lib = _lib.load()
    lib.hitadv_debug_v3_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.hitadv_debug_v3_stamps(host, n) == 0
    st = np.frombuffer(host, dtype=np.uint64).reshape(3, 1024, 8).astype(np.int64)[:, :512]
    res = {}
    for stage in (2, 1, 0):
        s = st[stage]
        live = s[:, 7] > 0
        s = s[live]
        t0 = s[:, 0].min()
        d = {'blocks': int(live.sum()), 'first_start_to_last_end_us': round(float(s[:, 7].max() - t0) * 0.01, 2),
             'block_start_after_first_us': {'median': round(float(np.median(s[:, 0] - t0)) * 0.01, 2), 'max': round(float((s[:, 0] - t0).max()) * 0.01, 2)},
             'block_lifetime_us': {'median': round(float(np.median(s[:, 7] - s[:, 0])) * 0.01, 2), 'max': round(float((s[:, 7] - s[:, 0]).max()) * 0.01, 2)}}
        full = s[(s[:, 1:6] > 0).all(1)]  # blocks that had work (D > 0)
        if len(full):
            if stage == 2:  # no stamp 6: the stage returns from inside its last branch
                full = full.copy()
                full[:, 6] = full[:, 5]
            ph = np.diff(full[:, :8], axis=1) * 0.01
            d['phase_median_us'] = {PHASES[i]: round(float(np.median(ph[:, i])), 2) for i in range(7)}
            slow = full[np.argsort(full[:, 7] - full[:, 0])[-16:]]
            d['phase_median_us_of_the_16_slowest_blocks'] = {PHASES[i]: round(float(np.median(np.diff(slow, axis=1)[:, i] * 0.01)), 2) for i in range(7)}
        res['stage%d' % stage] = d
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
