"""The headline configuration against the REFERENCE ITSELF at a horizon that means something.

Fixture ``g5d_attack_pointnet.npz`` (tests/golden/make_golden.py g5d): the imported reference's ``HiT_ADV.attack`` on cfg2's
own shape -- seeded ``PointNetFeatureModel`` with shaken BatchNorm statistics, B = 32, N = 1024, C = 192, T = 256, eval.py's
hyper-parameters, ``binary_step = 2 x num_iter = 50`` -- with, per iteration, its logits / adversarial loss / bookkeeping
distance, its parameters every tenth iteration, its deformed clouds at the ends of each step, and its own bookkeeping
variables after each step (read through ``sys.settrace``).  19 of the 32 clouds succeed: 9 in both steps (lower bound 45),
10 in the first only (10), 13 never (0) -- every branch of the best tracking and of the bisection fires.

Held here, on the DEFAULT engine (fp16x2 matrix mode): every one of the 100 iterations in the eager loop, then the captured
graphs (ten iterations per replay) and the attack inside a stack of ``attack_many`` -- both bitwise equal to the eager run.
Tolerances: what 50 Adam steps of fp32 re-association cost was MEASURED on MI355X and is pinned through
tests/golden/parity_pins.json (``close()`` asserts 4x the pinned figure)."""
import contextlib
import io

import numpy as np
import pytest
import torch

from helpers import T, close, golden, hp_from_fixture, note, pointnet_from_fixture, synth_batch

pytestmark = pytest.mark.gpu


def _attacker(fx, **kw):
    from hit_adv_amd.ShapeAttack.HiT_ADV import HiT_ADV
    from hit_adv_amd.util.adv_utils import UntargetedLogitsAdvLoss
    return HiT_ADV(pointnet_from_fixture(fx), adv_func=UntargetedLogitsAdvLoss(kappa=30.), verbose=False,
                   **hp_from_fixture(fx), **kw)


def _cpu(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _watch(att, B, per_iteration):
    """Read the device-resident bookkeeping after every binary step (and, in the eager loop, after every iteration)."""
    out = dict(steps=[], rows=[], taken=-np.ones((B, 2), dtype=np.int64))
    last = dict(obd=np.full(B, 1e10), at=(-1, -1))
    end_step, begin_step, iteration = att._end_step, att._begin_step, att._iteration

    def watched_begin(ws, binary_step):
        last['at'] = (binary_step, -1)
        begin_step(ws, binary_step)

    def watched_iteration(ws):
        last['at'] = (last['at'][0], last['at'][1] + 1)
        iteration(ws)
        torch.cuda.synchronize()
        st = ws.state
        obd = _cpu(st['o_bestdist'])
        out['taken'][obd != last['obd']] = last['at']
        last['obd'] = obd
        out['rows'].append(dict(P=ws.P.detach().cpu().numpy().copy(), sigma=ws.sigma.detach().cpu().numpy().copy(),
                                adv=ws.adv.cpu().numpy().copy(), pred=st['pred'].cpu().numpy().copy(),
                                adv_loss=ws.adv_loss.item(), dist_val=st['dist_val'].cpu().numpy().copy()))

    def watched_end(ws):
        end_step(ws)
        st = ws.state
        out['steps'].append(dict(lower=_cpu(ws.lower), upper=_cpu(ws.upper), scale_const=_cpu(ws.scale_const),
                                 o_bestdist=_cpu(st['o_bestdist']), o_bestscore=_cpu(st['o_bestscore']),
                                 bestdist=_cpu(st['bestdist']), bestscore=_cpu(st['bestscore'])))

    att._begin_step, att._end_step = watched_begin, watched_end
    if per_iteration:
        att._iteration = watched_iteration
    return out


def _mostly_close(a, b, what, typical=1e-4, worst=5e-2):
    """Parameters / clouds after tens of Adam steps.  Adam's update is lr * m / (sqrt(v) + eps): a coordinate whose gradient is
    smaller than its own evaluation error (a centre whose kernel no point feels: |g| ~ 1e-7 with either sign) moves by a
    normalised step of up to lr = 0.05 per iteration in whichever direction the rounding fell, in the reference as well; such
    a coordinate does not move the cloud (that is what a vanishing gradient means).  MEASURED on MI355X against the
    reference over 2 x 50 iterations: parameters p99.9 1.2e-5 / 3.4e-5 (end of step 0 / 1), largest 2.0e-4 / 1.4e-2; deformed
    clouds: see the bounds at the call sites; distances and losses agree to 1.4e-6 at every iteration.  So: the 99.9th
    percentile of |gpu - reference| is held to `typical`, the largest to `worst` (one Adam step); both recorded."""
    err = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).ravel()
    p999 = float(np.quantile(err, 0.999))
    note(what + ' p99.9', p999)
    note(what + ' max', float(err.max()))
    note(what + ' share beyond 1e-5', float((err > 1e-5).mean()))
    assert p999 <= typical and err.max() <= worst, (what, p999, float(err.max()))


def _check_steps(steps, fx):
    """The fixture's rows were read from the running reference's variables when each step's bisection was done -- the LAST row at
    the function's return, i.e. after the failure fill (:277-281) had overwritten o_bestdist of the samples that never
    succeeded; before the fill those still hold the initial 1e10, which is what the device holds after the last step."""
    never = fx['step_lower'][-1] == 0.
    for i, rec in enumerate(steps):
        for name in ('lower', 'upper', 'scale_const', 'o_bestscore', 'bestscore'):  # discrete: exact
            np.testing.assert_array_equal(rec[name], fx['step_' + name][i], err_msg="%s after step %d" % (name, i))
        for name in ('o_bestdist', 'bestdist'):
            want = fx['step_' + name][i].copy()
            if name == 'o_bestdist' and i == len(steps) - 1:
                want[never] = 1e10
            close(rec[name], want, rtol=1e-4, atol=0, what='%s_step%d' % (name, i))


def _check_result(att, best, succ, fx):
    assert best.dtype == np.float64 and best.shape == fx['best'].shape
    assert int(succ) == int(fx['success_num']) == 19
    np.testing.assert_array_equal(att.last_lower_bound.numpy().astype(np.float64), fx['step_lower'][-1])
    close(att.last_bestdist, fx['final_o_bestdist'], rtol=1e-4, atol=0, what='final_o_bestdist')
    close(best, fx['best'], rtol=0, atol=1e-4, what='best')


def test_headline_victim_every_iteration_of_2x50_vs_the_reference():
    fx = golden('g5d_attack_pointnet.npz')
    data, _ = synth_batch(32, 1024, first=int(fx['first']))
    hp = hp_from_fixture(fx)
    iters = hp['num_iter']
    att = _attacker(fx, use_graph=False)
    seen = _watch(att, 32, per_iteration=True)
    torch.manual_seed(int(fx['seed']))
    best, succ = att.attack(data, T(fx['target']))
    assert att._view is not None and att._view.hip_engine and att._view.matrix_mode == 'fp16x2'  # the engine the bench times
    ws = next(iter(att._ws.values()))
    assert torch.equal(ws.central.cpu(), T(fx['central']))  # the same 192 centres in all 32 clouds
    rows = seen['rows']
    assert len(rows) == 2 * iters and len(seen['steps']) == 2

    ref_pred = fx['logits'].argmax(-1)
    top2 = np.sort(fx['logits'], -1)
    margin = top2[..., -1] - top2[..., -2]
    pred = np.stack([r['pred'] for r in rows])
    agree = float((pred == ref_pred).mean())
    note('prediction agreement over 100 iterations x 32 clouds', agree)
    assert agree >= 0.99
    sure = margin > 1e-4  # where the reference's own top-2 logits are further apart than fp32 re-association moves them
    assert (pred[sure] == ref_pred[sure]).all()
    # successes per iteration: the curve the reference printed (14 -> 0 within 20 iterations of each step)
    ok = (pred != fx['target'][None]).sum(1)
    note('largest difference of the per-iteration success count', float(np.abs(ok - (ref_pred != fx['target'][None]).sum(1)).max()))

    kept = [int(k) for k in fx['kept_iterations']]
    for step in range(2):
        for i in range(iters):
            r, j = rows[step * iters + i], step * iters + i
            tag = 's%d_i%02d' % (step, i)
            # every iteration of the first ten, then every fifth: the comparisons are recorded one by one
            if i < 10 or i % 5 == 0 or i == iters - 1:
                close(r['adv_loss'], fx['adv_loss'][j], rtol=1e-4, atol=1e-5, what='adv_loss_' + tag)
                close(r['dist_val'], fx['dist_val'][j], rtol=1e-4, atol=1e-6, what='dist_val_' + tag)
            if i in kept and i > 0:  # the fixture's row = the clamped parameters the iteration STARTS from
                k = step * len(kept) + kept.index(i)
                prev = rows[j - 1]
                _mostly_close(np.clip(prev['P'], -hp['budget'], hp['budget']), fx['P'][k], 'P_' + tag)
                _mostly_close(np.clip(prev['sigma'], hp['min_sigm'], hp['max_sigm']), fx['sigma'][k], 'sigma_' + tag)
        close(rows[step * iters]['adv'], fx['adv'][2 * step], rtol=0, atol=1e-5, what='adv_s%d_first' % step)
        _mostly_close(rows[step * iters + iters - 1]['adv'], fx['adv'][2 * step + 1], 'adv_s%d_last' % step, typical=3e-5, worst=1e-3)
    _check_steps(seen['steps'], fx)
    # (step, iteration) of every sample's last replacement of its overall best.  The samples that never succeed have none on
    # the device; the fixture's watcher saw the failure fill (:277-281) rewrite their o_bestdist at the very end: (1, 49)
    never = fx['step_lower'][-1] == 0.
    assert (fx['taken_step'][never] == 1).all() and (fx['taken_iter'][never] == iters - 1).all() and (seen['taken'][never] == -1).all()
    np.testing.assert_array_equal(seen['taken'][~never, 0], fx['taken_step'][~never])
    np.testing.assert_array_equal(seen['taken'][~never, 1], fx['taken_iter'][~never])
    _check_result(att, best, succ, fx)


def test_headline_victim_graphs_and_stack_reproduce_the_reference_run():
    """The default path (captured graphs, ten iterations per replay) and the attack as the FIRST member of a stack of four
    (attack_many: one victim pass over 128 clouds per iteration): the reference's bookkeeping after each step, its returned
    clouds and its success count; and both runs equal each other bit for bit."""
    fx = golden('g5d_attack_pointnet.npz')
    data, _ = synth_batch(32, 1024, first=int(fx['first']))
    target = T(fx['target'])
    att = _attacker(fx)
    seen = _watch(att, 32, per_iteration=False)
    torch.manual_seed(int(fx['seed']))
    with contextlib.redirect_stdout(io.StringIO()):
        best, succ = att.attack(data, target)
    assert att.last_graph_used and att._view.matrix_mode == 'fp16x2'
    assert len(seen['steps']) == 2
    _check_steps(seen['steps'], fx)
    _check_result(att, best, succ, fx)

    others = []
    cpu_model = pointnet_from_fixture(fx)
    for k in range(3):
        d, _ = synth_batch(32, 1024, first=9000 + 32 * k)
        with torch.no_grad():
            lab = cpu_model(d[:, :, :3].transpose(1, 2).contiguous())[0].argmax(1)
        others.append((d, lab))
    stacked = _attacker(fx)
    assert stacked.stacks()
    torch.manual_seed(int(fx['seed']))  # the first batch of the group takes the draws the single call took
    res = stacked.attack_many([(data, target)] + others)
    assert stacked.last_graph_used
    assert any(isinstance(k[3], str) for k in stacked._ws), "attack_many did not stack the victim passes"
    sbest, ssucc = res[0]
    assert np.array_equal(sbest, best) and int(ssucc) == int(succ)  # stacked == alone, bitwise
    close(sbest, fx['best'], rtol=0, atol=1e-4, what='best (stacked)')
    note('successes of the three other batches of the stack', float(sum(int(k) for _, k in res[1:])))
