"""The headline's own inner horizon against the REFERENCE ITSELF (VERDICT r05 missing #4 / next-round #5).

Fixture ``g5e_attack_pointnet_500.npz`` (tests/golden/make_golden.py g5e; 1,929 s of this container's CPU): the imported reference's
``HiT_ADV.attack`` on cfg2's shape -- the victim, clouds and seed of g5d -- with ``binary_step = 1 x num_iter = 500``, i.e. one of the
ten steps eval.py:126-133 runs per batch.  Per iteration: the prediction, the top-two margin, the adversarial loss and the distance
the bookkeeping compares; the logits every 25th; (P, sigma) every 50th; the deformed clouds at iterations 0 and 499; the reference's
bookkeeping after the step and what it returns.  Its first 50 iterations are bit-identical to g5d's first step (two separate runs of
the reference: tests/test_oracle_golden.py), and the Python oracle reproduces all 500 (tests/golden/g5e_oracle_report.json).

Held here on the DEFAULT engine (fp16x2), eager loop, every iteration watched: predictions (>= 99 %, and all where the reference's
margin exceeds 1e-4), the loss and the distance at every 25th iteration, the discrete bookkeeping exactly, the returned clouds -- the
successes' clouds were taken within the first ten iterations, the others are the LAST iterate (failure fill, HiT_ADV.py:277-281), so
they carry what 500 Adam steps of fp32 re-association cost: reported as p50 / p99.9 / max and bounded.

STATUS: written in round 6, which had no GPU access -- NOT RUN ON HARDWARE YET; the float bounds below are round 5's measured 2 x 50
figures widened for ten times the horizon, to be replaced by pins (tools/make_parity_pins.py) on the first run.  The file sorts last on
purpose (`pytest -x`)."""
import numpy as np
import pytest
import torch

from helpers import T, close, golden, hp_from_fixture, note, synth_batch
from test_gpu_headline_parity import _attacker, _watch

pytestmark = pytest.mark.gpu


def _drift(a, b, what, p999_bound, max_bound):
    err = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).ravel()
    p50, p999, mx = float(np.quantile(err, 0.5)), float(np.quantile(err, 0.999)), float(err.max())
    note(what + ' p50', p50)
    note(what + ' p99.9', p999)
    note(what + ' max', mx)
    assert p999 <= p999_bound and mx <= max_bound, (what, p50, p999, mx)


def test_headline_victim_one_step_of_500_iterations_vs_the_reference():
    fx = golden('g5e_attack_pointnet_500.npz')
    data, _ = synth_batch(32, 1024, first=int(fx['first']))
    hp = hp_from_fixture(fx)
    iters = hp['num_iter']
    assert iters == 500 and hp['binary_step'] == 1
    att = _attacker(fx, use_graph=False)
    seen = _watch(att, 32, per_iteration=True)
    torch.manual_seed(int(fx['seed']))
    best, succ = att.attack(data, T(fx['target']))
    assert att._view is not None and att._view.hip_engine and att._view.matrix_mode == 'fp16x2'
    ws = next(iter(att._ws.values()))
    assert torch.equal(ws.central.cpu(), T(fx['central']))
    rows = seen['rows']
    assert len(rows) == iters and len(seen['steps']) == 1

    pred = np.stack([r['pred'] for r in rows])
    agree = float((pred == fx['pred']).mean())
    note('prediction agreement over 500 iterations x 32 clouds', agree)
    assert agree >= 0.99
    sure = fx['margin'] > 1e-4
    note('share of (iteration, cloud) pairs with a reference margin above 1e-4', float(sure.mean()))
    assert (pred[sure] == fx['pred'][sure]).all()
    ok, ref_ok = (pred != fx['target'][None]).sum(1), (fx['pred'] != fx['target'][None]).sum(1)
    note('largest difference of the per-iteration success count', float(np.abs(ok - ref_ok).max()))
    assert (ok[25:] == 0).all() and (ref_ok[25:] == 0).all()  # the reference's curve: 14 successes at iteration 0, none after iteration ~20

    kept = [int(k) for k in fx['kept_iterations']]
    for i in list(range(10)) + list(range(25, iters, 25)) + [iters - 1]:
        close(rows[i]['adv_loss'], fx['adv_loss'][i], rtol=1e-4, atol=1e-5, what='adv_loss_i%03d' % i)
        close(rows[i]['dist_val'], fx['dist_val'][i], rtol=1e-3, atol=1e-6, what='dist_val_i%03d' % i)
    for k, i in enumerate(kept):
        if i == 0:
            continue
        prev = rows[i - 1]  # the fixture's row = the clamped parameters the iteration STARTS from
        _drift(np.clip(prev['P'], -hp['budget'], hp['budget']), fx['P'][k], 'P_i%03d' % i, 1e-3, 0.1)
        _drift(np.clip(prev['sigma'], hp['min_sigm'], hp['max_sigm']), fx['sigma'][k], 'sigma_i%03d' % i, 1e-3, 0.1)
    close(rows[0]['adv'], fx['adv'][0], rtol=0, atol=1e-5, what='adv_first')
    _drift(rows[iters - 1]['adv'], fx['adv'][1], 'adv_last (iteration 499)', 1e-3, 5e-2)

    # the bookkeeping after the step: discrete parts exactly (every best was taken within the first eleven iterations)
    rec = seen['steps'][0]
    for name in ('lower', 'upper', 'scale_const', 'o_bestscore', 'bestscore'):
        np.testing.assert_array_equal(rec[name], fx['step_' + name][0], err_msg=name)
    never = fx['step_lower'][0] == 0.
    assert int(never.sum()) == 13 and int(succ) == int(fx['success_num']) == 19
    np.testing.assert_array_equal(seen['taken'][~never, 1], fx['taken_iter'][~never])
    assert fx['taken_iter'][~never].max() <= 10 and (seen['taken'][never] == -1).all()
    close(rec['bestdist'][~never], fx['step_bestdist'][0][~never], rtol=1e-4, atol=0, what='bestdist')
    # what comes back: the successes' clouds are iterates of the first eleven iterations ...
    assert best.dtype == np.float64 and best.shape == fx['best'].shape
    close(best[~never], fx['best'][~never], rtol=0, atol=1e-4, what='best (successes)')
    # ... the others are iteration 499's clouds: the cloud-level cost of the whole horizon
    _drift(best[never], fx['best'][never], 'returned clouds of the 13 failures (the last iterate)', 1e-3, 5e-2)
    close(att.last_bestdist[never], fx['final_o_bestdist'][never], rtol=1e-3, atol=0, what='final_o_bestdist (failures)')
