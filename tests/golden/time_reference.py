"""Build-container only (needs /root/reference): wall time of the UNMODIFIED reference's HiT_ADV.attack on the CPU next to
the CPU oracle's, same inputs, same thread count -- the check SURVEY.md 8(d) asks for before the oracle is used as the
bench's cpu_baseline ("kind": "port").  Prints one JSON line; the numbers are quoted in DESIGN.md section 7.

    python tests/golden/time_reference.py [B] [iterations]
"""
import io
import json
import os
import sys
import time
from contextlib import redirect_stdout

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ref_harness  # noqa: E402

ref_harness.install()
from ShapeAttack.HiT_ADV import HiT_ADV  # noqa: E402  (the reference)
from util import adv_utils  # noqa: E402
from model import feature_models  # noqa: E402

from helpers import synth_batch  # noqa: E402
from oracle import hitadv_oracle as O  # noqa: E402

HP = dict(attack_lr=1e-2, central_num=192, total_central_num=256, init_weight=10., max_weight=80., cd_weight=1e-4,
          ker_weight=1., hide_weight=1., curv_loss_knn=16, max_sigm=1.2, min_sigm=0.1, budget=0.55)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    torch.set_num_threads(8)
    torch.manual_seed(0)
    model = feature_models.PointNetFeatureModel(40, normal_channel=False)
    model.eval()  # (the reference's FeatureModel.eval() returns None)
    data, _ = synth_batch(B, 1024)
    with torch.no_grad():
        label = model(data[:, :, :3].transpose(1, 2).contiguous())[0].argmax(1)
    out = {}
    for n in (iters, 2 * iters):  # two lengths -> per-iteration time without the setup
        att = HiT_ADV(model, adv_func=adv_utils.UntargetedLogitsAdvLoss(kappa=30.), binary_step=1, num_iter=max(n, 5), **HP)
        torch.manual_seed(1)
        t0 = time.perf_counter()
        with redirect_stdout(io.StringIO()):
            ref_best, _ = att.attack(data, label)
        out['reference_%d' % n] = time.perf_counter() - t0
        orc = O.HiTADVOracle(model, lambda l, t: O.untargeted_logits_adv_loss(l, t, 30.), binary_step=1,
                             num_iter=max(n, 5), **HP)
        torch.manual_seed(1)
        t0 = time.perf_counter()
        with redirect_stdout(io.StringIO()):
            orc_best, _ = orc.attack(data, label)
        out['oracle_%d' % n] = time.perf_counter() - t0
        out['max_abs_diff_%d' % n] = float(abs(ref_best - orc_best).max())
    a, b = max(iters, 5), max(2 * iters, 5)
    ref_it = (out['reference_%d' % (2 * iters)] - out['reference_%d' % iters]) / (b - a)
    orc_it = (out['oracle_%d' % (2 * iters)] - out['oracle_%d' % iters]) / (b - a)
    print(json.dumps(dict(B=B, threads=8, s_per_iteration_reference=round(ref_it, 3), s_per_iteration_oracle=round(orc_it, 3),
                          ratio=round(orc_it / ref_it, 3), **{k: round(v, 6) for k, v in out.items()})))


if __name__ == '__main__':
    main()
