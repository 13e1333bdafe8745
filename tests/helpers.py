"""Shared test helpers: fixture loading, synthetic inputs, the toy victim."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


PARITY = {}  # test id -> worst achieved errors of its float comparisons (dumped by conftest at session end)
PIN_FACTOR = 4.0


def _pins():
    """tests/golden/parity_pins.json (tools/make_parity_pins.py): the error ACHIEVED on MI355X for every comparison whose
    written tolerance was more than 10x looser than that; on GPU runs close() also asserts 4x the pinned figure."""
    global _PINS
    if _PINS is None:
        path = os.path.join(GOLDEN, "parity_pins.json")
        _PINS = {}
        if torch.cuda.is_available() and os.path.exists(path) and os.environ.get('HITADV_NO_PARITY_PINS') != '1':
            with open(path) as f:
                _PINS = json.load(f)
    return _PINS


_PINS = None


def close(a, b, rtol=1e-5, atol=1e-6, what=None):
    """assert_allclose(a, b) that also RECORDS what was achieved: the largest |a-b|, the largest |a-b|/|b| over the
    elements with |b| above the absolute tolerance, and |a-b|_max / |b|_max, per test, next to the bound asserted."""
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    a64, b64 = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a64.shape == b64.shape and a64.size:
        err = np.abs(a64 - b64)
        big = np.abs(b64) > max(atol, 1e-30)
        rec = dict(max_abs=float(err.max()), max_rel=float((err[big] / np.abs(b64[big])).max()) if big.any() else 0.0,
                   max_abs_over_scale=float(err.max() / max(np.abs(b64).max(), 1e-30)), rtol=rtol, atol=atol, n=int(a64.size))
        test = os.environ.get('PYTEST_CURRENT_TEST', 'unknown').split(' ')[0]
        rows = PARITY.setdefault(test, [])
        name = what or 'cmp%d' % len(rows)
        pin = _pins().get(test, {}).get(name)
        if pin is not None:
            rec['pinned_max_abs'] = pin['max_abs']
            rec['atol_from_pin'] = PIN_FACTOR * pin['max_abs']
        rows.append(dict(what=name, **rec))
        if pin is not None:
            assert err.max() <= PIN_FACTOR * pin['max_abs'], (
                "%s / %s: max |a - b| = %.3g exceeds %g x the %.3g achieved on MI355X (tests/golden/parity_pins.json)"
                % (test, name, err.max(), PIN_FACTOR, pin['max_abs']))
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def note(what, value):
    """A recorded statistic (no bound asserted, none implied): lands in the parity report next to the comparisons."""
    test = os.environ.get('PYTEST_CURRENT_TEST', 'unknown').split(' ')[0]
    PARITY.setdefault(test, []).append(dict(what=what + ' [statistic only]', max_abs=float(value), max_rel=0.0, max_abs_over_scale=0.0,
                                            rtol=0.0, atol=0.0, n=1, statistic_only=True))


def golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def golden_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def T(a):
    t = torch.from_numpy(np.asarray(a))
    return t


from hit_adv_amd.Dataset.synthetic import ToyVictim, synth_batch, synth_cloud  # noqa: E402,F401


def toy_from_fixture(fx):
    m = ToyVictim()
    m.load_state_dict({k[2:]: T(v) for k, v in fx.items() if k.startswith('w_')})
    return m.eval()


def hp_from_fixture(fx):
    hp = {}
    for k, v in fx.items():
        if k.startswith('hp_'):
            v = v.item()
            hp[k[3:]] = int(v) if float(v).is_integer() and k[3:] in (
                'central_num', 'total_central_num', 'binary_step', 'num_iter', 'curv_loss_knn') else float(v)
    return hp


def gradient_close(g, gd, what, frac_bound=2e-3, l2_bound=3e-2):
    """An input gradient of a deep victim against a reference evaluation (float64 module, or the reference's own fp32
    gradient in a fixture).  The gradient is piecewise: where two evaluations disagree on a ReLU sign or on the winner of a
    max whose two best candidates are an fp32 rounding apart, a whole path's contribution moves.  So: all but a few elements
    agree to 1e-3, and what the few that do not carry is small against the gradient as a whole.  Records what was achieved."""
    g = torch.as_tensor(np.asarray(g.detach().cpu() if torch.is_tensor(g) else g)).double()
    gd = torch.as_tensor(np.asarray(gd.detach().cpu() if torch.is_tensor(gd) else gd)).double()
    scale = float(gd.abs().max())
    bad = float(((g - gd).abs() > 1e-3 * gd.abs() + 1e-5 * scale).double().mean())
    l2 = float((g - gd).norm() / gd.norm())
    PARITY.setdefault(os.environ.get('PYTEST_CURRENT_TEST', 'unknown').split(' ')[0], []).append(
        dict(what=what + ' (fraction of elements off by > 1e-3, relative L2 error)', max_abs=bad, max_rel=l2,
             max_abs_over_scale=l2, rtol=l2_bound, atol=frac_bound, n=int(g.numel())))
    assert bad <= frac_bound and l2 <= l2_bound, (what, bad, l2)


def pointnet_from_fixture(fx):
    """The PointNet victim of fixture g5d: NOT stored (14 MB) -- it is ``torch.manual_seed(model_seed)`` + the default
    initialisation + ``shake_bn``; the fixture carries a checksum of every tensor of the reference's own instance."""
    from hit_adv_amd.Dataset.synthetic import shake_bn
    from hit_adv_amd.model.pointnet import PointNetFeatureModel
    torch.manual_seed(int(fx['model_seed']))
    m = PointNetFeatureModel(40, normal_channel=False).eval()
    shake_bn(m, seed=int(fx['shake_seed']), mean_std=float(fx['shake_mean_std']), var_spread=float(fx['shake_var_spread']))
    sums = np.array([float(v.double().abs().sum()) for v in m.state_dict().values()])
    np.testing.assert_allclose(sums, fx['weight_checksum'], rtol=1e-12, atol=0,
                               err_msg="the seeded PointNet is not the one the fixture was made with")
    return m
