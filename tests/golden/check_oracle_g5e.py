#!/usr/bin/env python3
"""The Python oracle against the reference over the headline's FULL inner horizon.

Fixture g5e (make_golden.py g5e) = the imported reference's HiT_ADV.attack on cfg2's shape, real PointNet, binary_step = 1 x
num_iter = 500.  This script runs oracle/hitadv_oracle.py::HiTADVOracle on the same clouds, victim, seed and hyper-parameters --
~25 min on this container's 8 cores, which is why it is a script with a committed report (tests/golden/g5e_oracle_report.json) and
not a test of the CPU suite; tests/test_oracle_golden.py holds the first iterations of the same fixture in seconds.  Needs nothing
of /root/reference (the fixture is data); run from the repo root:  python tests/golden/check_oracle_g5e.py"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import T, golden, hp_from_fixture, pointnet_from_fixture, synth_batch  # noqa: E402
from oracle import hitadv_oracle as O  # noqa: E402


class Reduce(list):
    """The oracle's `trace` sink: keeps per iteration only what the report needs (500 full records would be ~250 MB)."""

    def __init__(self, kept, iters):
        super().__init__()
        self.kept, self.iters = set(int(k) for k in kept), iters
        self.rows, self.P, self.sigma, self.adv = [], {}, {}, {}

    def append(self, rec):
        it = rec['it']
        self.rows.append(dict(pred=rec['pred'].copy(), adv_loss=rec['adv_loss'], dist_val=rec['dist_val'].copy()))
        if it + 1 in self.kept or it == self.iters - 1:  # the fixture's row k = the CLAMPED parameters iteration k STARTS from
            self.P[it + 1], self.sigma[it + 1] = rec['P'].copy(), rec['sigma'].copy()
        if it in (0, self.iters - 1):
            self.adv[it] = rec['adv'].copy()


def stats(a, b):
    e = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).ravel()
    return dict(p50=float(np.quantile(e, 0.5)), p999=float(np.quantile(e, 0.999)), max=float(e.max()))


def main():
    fx = golden('g5e_attack_pointnet_500.npz')
    hp = hp_from_fixture(fx)
    prefix = int(sys.argv[1]) if len(sys.argv) > 1 else None  # a dry run over the first n iterations (no report written)
    if prefix:
        hp['num_iter'] = prefix
        for k in ('pred', 'adv_loss', 'dist_val', 'margin'):
            fx[k] = fx[k][:prefix]
    iters = hp['num_iter']
    model = pointnet_from_fixture(fx)
    data, _ = synth_batch(32, 1024, first=int(fx['first']))
    oracle = O.HiTADVOracle(model, lambda l, t: O.untargeted_logits_adv_loss(l, t, 30.), **hp)
    sink = Reduce(fx['kept_iterations'], iters)
    torch.manual_seed(int(fx['seed']))
    t0 = time.time()
    best, succ = oracle.attack(data, T(fx['target']), trace=sink)
    secs = time.time() - t0
    st = oracle.state
    pred = np.stack([r['pred'] for r in sink.rows])
    adv_loss = np.array([r['adv_loss'] for r in sink.rows])
    dist_val = np.stack([r['dist_val'] for r in sink.rows])
    sure = fx['margin'] > 1e-5
    rep = dict(
        what="HiTADVOracle vs fixture g5e (the reference's own 1 x 500 run at cfg2's shape)", seconds=round(secs, 1),
        threads=torch.get_num_threads(), centres_bit_equal=bool(torch.equal(st['central'], T(fx['central']))),
        prediction_agreement=float((pred == fx['pred']).mean()),
        predictions_equal_where_reference_margin_above_1e5=bool((pred[sure] == fx['pred'][sure]).all()),
        share_of_predictions_with_margin_above_1e5=float(sure.mean()),
        success_num=[int(succ), int(fx['success_num'])],
        lower_bound_equal=bool(np.array_equal(st['steps'][0]['lower'], fx['step_lower'][0])),
        scale_const_equal=bool(np.array_equal(st['steps'][0]['scale_const'], fx['step_scale_const'][0])),
        bestscore_equal=bool(np.array_equal(st['steps'][0]['o_bestscore'], fx['step_o_bestscore'][0])),
        taken_iteration_equal=bool(np.array_equal(st['taken'][:, 1][fx['step_lower'][0] > 0], fx['taken_iter'][fx['step_lower'][0] > 0])),
        adv_loss_max_rel=float(np.max(np.abs(adv_loss - fx['adv_loss']) / np.abs(fx['adv_loss']))),
        dist_val_max_rel_by_iteration={str(i): float(np.max(np.abs(dist_val[i] - fx['dist_val'][i]) / np.abs(fx['dist_val'][i])))
                                       for i in (0, 10, 25, 50, 100, 200, 300, 400, 499) if i < iters},
        dist_val_max_rel=float(np.max(np.abs(dist_val - fx['dist_val']) / np.abs(fx['dist_val']))),
        P_drift_by_iteration={}, sigma_drift_by_iteration={},
        adv_first=stats(sink.adv[0], fx['adv'][0]), adv_last=stats(sink.adv[iters - 1], fx['adv'][1]),
        final_o_bestdist_max_rel=float(np.max(np.abs(st['o_bestdist'] - fx['final_o_bestdist']) / np.abs(fx['final_o_bestdist']))),
        returned_clouds=stats(best, fx['best']))
    for k, it in enumerate(int(i) for i in fx['kept_iterations']):
        if it == 0 or it not in sink.P:
            continue
        rep['P_drift_by_iteration'][str(it)] = stats(np.clip(sink.P[it], -hp['budget'], hp['budget']), fx['P'][k])
        rep['sigma_drift_by_iteration'][str(it)] = stats(np.clip(sink.sigma[it], hp['min_sigm'], hp['max_sigm']), fx['sigma'][k])
    if not prefix:
        with open(os.path.join(HERE, 'g5e_oracle_report.json'), 'w') as f:
            json.dump(rep, f, indent=1, sort_keys=True)
    print(json.dumps(rep, indent=1, sort_keys=True))


if __name__ == '__main__':
    main()
