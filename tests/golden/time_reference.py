"""Build-container only (needs /root/reference): wall time of the UNMODIFIED reference's HiT_ADV.attack on the CPU next to
the CPU oracle's, same inputs, same thread count -- the check SURVEY.md 8(d) asks for before the oracle is used as the
bench's cpu_baseline ("kind": "port").  Prints one JSON line; the numbers are quoted in DESIGN.md section 7.

    python tests/golden/time_reference.py [B] [iterations]

Per-iteration time = the interval between consecutive ``kernel_density`` calls (one per inner iteration, at its start, in
both), after one untimed run of each, the two alternating.  (Round 3's version subtracted a 6- from a 12-iteration call of
each, reference first: the reference's first call also paid the process's one-time costs, which made its difference look
small -- the "1.44x slower port" of VERDICT r03 was that artefact; the op profiles of the two are the same, tools note in
DESIGN.md section 7.)
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


def _stamped(owner, name, stamps):
    """Wrap ``owner.name`` (called exactly once per inner iteration, at its start, by the reference and by the oracle alike)
    so that every call leaves a time stamp: consecutive stamps are whole iterations, setup and one-time costs excluded."""
    inner = getattr(owner, name)

    def f(*a, **k):
        stamps.append(time.perf_counter())
        return inner(*a, **k)
    setattr(owner, name, f)
    return inner


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    torch.set_num_threads(8)
    torch.manual_seed(0)
    model = feature_models.PointNetFeatureModel(40, normal_channel=False)
    model.eval()  # (the reference's FeatureModel.eval() returns None)
    data, _ = synth_batch(B, 1024)
    with torch.no_grad():
        label = model(data[:, :, :3].transpose(1, 2).contiguous())[0].argmax(1)

    def reference(n, stamps):
        att = HiT_ADV(model, adv_func=adv_utils.UntargetedLogitsAdvLoss(kappa=30.), binary_step=1, num_iter=max(n, 5), **HP)
        _stamped(att, 'kernel_density', stamps)
        torch.manual_seed(1)
        with redirect_stdout(io.StringIO()):
            return att.attack(data, label)[0]

    def oracle(n, stamps):
        orc = O.HiTADVOracle(model, lambda l, t: O.untargeted_logits_adv_loss(l, t, 30.), binary_step=1, num_iter=max(n, 5), **HP)
        real = _stamped(O, 'kernel_density', stamps)
        try:
            torch.manual_seed(1)
            with redirect_stdout(io.StringIO()):
                return orc.attack(data, label)[0]
        finally:
            O.kernel_density = real

    # one-time costs (thread pool, oneDNN primitives, allocator) are paid by whoever runs first: both run once untimed,
    # then the two alternate, so that neither is measured on a colder machine than the other
    reference(5, [])
    oracle(5, [])
    per = dict(reference=[], oracle=[])
    diff = 0.
    for _ in range(2):
        sr, so = [], []
        rb = reference(iters, sr)
        ob = oracle(iters, so)
        diff = max(diff, float(abs(rb - ob).max()))
        per['reference'] += [b - a for a, b in zip(sr, sr[1:])]
        per['oracle'] += [b - a for a, b in zip(so, so[1:])]
    med = {k: sorted(v)[len(v) // 2] for k, v in per.items()}
    print(json.dumps(dict(B=B, threads=8, timed_iterations_each=len(per['oracle']),
                          s_per_iteration_reference=round(med['reference'], 4), s_per_iteration_oracle=round(med['oracle'], 4),
                          ratio=round(med['oracle'] / med['reference'], 3), max_abs_diff=diff,
                          method="median of the intervals between consecutive kernel_density calls (one per iteration in both); "
                                 "both warmed up, runs alternated")))


if __name__ == '__main__':
    main()
