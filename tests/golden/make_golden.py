"""Generate the golden fixtures in this directory from the reference itself.

Runs ONLY in the build container (needs /root/reference, CPU torch).  Imports the
reference unmodified through ``ref_harness`` (stand-ins for absent third-party
packages only), feeds it seeded synthetic inputs and stores inputs + outputs as
small .npz files.  The fixtures are data; no reference source text is stored.

    python tests/golden/make_golden.py

Fixture index:
  g1_set_distance.npz  chamfer / hausdorff / ChamferDist / HausdorffDist (+Q1 call)
  g2_knn_dist.npz      KNNDist values, CurvStdDist, kappa / kappa-std
  g3_deform.npz        kernel_density, 192-step deformation, grads wrt (P, sigma)
  g4_fps.npz           farthest_point_sample with recorded start indices
  g5_attack.npz        full HiT_ADV.attack trajectory with a toy victim
  g5b_attack_wide.npz  short trajectory at eval.py sizes (N=1024, C=192, T=256)
  g5c_attack_long.npz  ten binary steps x 20 iterations: the bisection bounds after every step, the best-so-far records
                       and the (step, iteration) each record was taken at, read from the running reference's own variables
  g6_adv_clip.npz      adversarial losses and clip/projection operators
  g7_cwknn.npz         CWKNN.attack trajectory with the toy victim
  g8_state_dicts.json  state_dict key/shape lists of the victims
  g9_cwperturb.npz     CWPerturb.attack (L2Dist + ClipPointsLinf) trajectory with the toy victim
  g11_pointnet2.npz    PointNet++ SSG (seeded init + seeded FPS starts): FPS / ball-query tables, logits, input grad
  g12_pct.npz          PCT (seeded init + seeded FPS starts): logits, input gradient, first FPS table
  g13_aof.npz          CWAOF.attack trajectory (torch.symeig served by torch.linalg.eigh) with the toy victim
  g14..g18             CWPerturbT / CWAdvPC / CWUAdvPC / CWTAOF / CWUAEAOF trajectories (toy victim, toy auto-encoder)
  g19..g21             CWAdd (Chamfer and Hausdorff) / CWAddClusters / CWAddObjects results
  g22_dist_more.npz    LaplacianDist, FarthestDist, FarChamferDist, L2ChamferDist, CurvDist values and gradients
  g23_datasets.npz     ModelNetDataLoader / PartNormalDataset items read from the tiny tree in g23_dataset_tree.json
  g24_cwuknn.npz       CWUKNN.attack trajectories (ProjectInnerClipLinf, pre_head): L2Dist and ChamferkNNDist
  g10_dgcnn.npz        DGCNN_cls (seeded init, eval mode): logits, input gradient, first-layer kNN table
"""
import io
import json
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

ref_harness.install()

from ShapeAttack.HiT_ADV import HiT_ADV  # noqa: E402
from util import dist_utils, set_distance, adv_utils, clip_utils  # noqa: E402
from CW.kNN import CWKNN  # noqa: E402
from CW.Perturb import CWPerturb  # noqa: E402

torch.set_num_threads(8)


def synth_cloud(cloud_id, n):
    """BASELINE.md section 3 synthetic input: unit-ball gaussian cloud + unit normals."""
    g = torch.Generator('cpu').manual_seed(1234 + cloud_id)
    xyz = torch.randn(n, 3, generator=g)
    xyz = xyz - xyz.mean(0, keepdim=True)
    xyz = xyz / xyz.norm(dim=1).max()
    nrm = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1)
    label = torch.randint(0, 40, (1,), generator=g)
    return torch.cat([xyz, nrm], 1), label


def synth_batch(b, n, first=0):
    cl = [synth_cloud(first + i, n) for i in range(b)]
    return torch.stack([c[0] for c in cl]), torch.cat([c[1] for c in cl])


class ToyVictim(torch.nn.Module):
    """< 1K parameters; weights are stored in the fixture."""

    def __init__(self, classes=40, width=16):
        super().__init__()
        self.conv = torch.nn.Conv1d(3, width, 1)
        self.fc = torch.nn.Linear(width, classes)

    def forward(self, x):
        h = torch.relu(self.conv(x))
        return self.fc(torch.max(h, 2)[0])


def toy_victim(seed):
    torch.manual_seed(seed)
    m = ToyVictim()
    with torch.no_grad():
        m.conv.weight.mul_(3.0)
        m.fc.weight.mul_(4.0)
    return m.eval()


def npify(d):
    out = {}
    for k, v in d.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    return out


def save(name, d):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **npify(d))
    print("wrote %-24s %7.1f KB" % (name, os.path.getsize(path) / 1024))


# ------------------------------------------------------------------ G1
def g1():
    data, _ = synth_batch(2, 1024)
    ori = data[:, :, :3].contiguous()
    g = torch.Generator('cpu').manual_seed(7)
    adv = ori + 0.02 * torch.randn(ori.shape, generator=g)
    other, _ = synth_batch(2, 256, first=10)
    small = other[:, :, :3].contiguous()
    w = torch.tensor([0.3, 1.7])
    out = dict(adv=adv, ori=ori, small=small, weights=w)
    out['chamfer_l1'], out['chamfer_l2'] = set_distance.chamfer(adv, ori)
    out['hausdorff_l1'], out['hausdorff_l2'] = set_distance.hausdorff(adv, ori)
    out['chamfer_small_l1'], out['chamfer_small_l2'] = set_distance.chamfer(small, ori)
    out['hausdorff_small_l1'], out['hausdorff_small_l2'] = set_distance.hausdorff(small, ori)
    for m in ('adv2ori', 'ori2adv', 'both'):
        out['ChamferDist_%s' % m] = dist_utils.ChamferDist(m)(adv, ori, w, batch_avg=False)
        out['HausdorffDist_%s' % m] = dist_utils.HausdorffDist(m)(adv, ori, w, batch_avg=False)
    out['ChamferDist_avg'] = dist_utils.ChamferDist()(adv, ori)
    out['HausdorffDist_avg'] = dist_utils.HausdorffDist()(adv, ori)
    # quirk Q1: the HiT-ADV call site passes [B,3,N] tensors (HiT_ADV.py:230)
    out['ChamferDist_q1'] = dist_utils.ChamferDist()(adv.transpose(1, 2).contiguous(),
                                                     ori.transpose(1, 2).contiguous(),
                                                     torch.from_numpy(np.ones(2) * 1e-4),
                                                     batch_avg=False)
    # autograd through the Chamfer operator (what CW/kNN.py:104-111 back-propagates)
    a = adv.clone().requires_grad_()
    dist_utils.ChamferDist('both')(a, ori, w).backward()
    out['ChamferDist_both_grad'] = a.grad
    a = adv.clone().requires_grad_()
    dist_utils.HausdorffDist('both')(a, ori, w).backward()
    out['HausdorffDist_both_grad'] = a.grad
    save('g1_set_distance.npz', out)


# ------------------------------------------------------------------ G2
def g2():
    data, _ = synth_batch(2, 1024, first=2)
    ori = data[:, :, :3].contiguous()
    nrm = data[:, :, 3:].contiguous()
    g = torch.Generator('cpu').manual_seed(8)
    adv = ori + 0.01 * torch.randn(ori.shape, generator=g)
    out = dict(ori=ori, normal=nrm, adv=adv)
    for k in (4, 5):
        out['KNNDist_k%d' % k] = dist_utils.KNNDist(k=k)(adv, batch_avg=False)
        out['KNNDist_k%d_chfirst' % k] = dist_utils.KNNDist(k=k)(adv.transpose(1, 2).contiguous(),
                                                                 batch_avg=False)
    a = adv.clone().requires_grad_()
    dist_utils.KNNDist(k=5)(a, torch.tensor([0.5, 2.0])).backward()
    out['KNNDist_k5_grad'] = a.grad
    a = adv.clone().requires_grad_()
    dist_utils.ChamferkNNDist()(a, ori).backward()
    out['ChamferkNNDist'] = dist_utils.ChamferkNNDist()(adv, ori, batch_avg=False)
    out['ChamferkNNDist_grad'] = a.grad
    ori_t, adv_t, nrm_t = (t.transpose(1, 2).contiguous() for t in (ori, adv, nrm))
    out['CurvStdDist_k4'] = dist_utils.CurvStdDist(k=4)(ori_t, adv_t, nrm_t)
    att = HiT_ADV.__new__(HiT_ADV)
    out['kappa_k16'] = att._get_kappa_ori(ori_t, nrm_t, k=16)
    out['kappa_std_k16'] = att._get_kappa_std_ori(ori_t, nrm_t, k=16)
    save('g2_knn_dist.npz', out)


# ------------------------------------------------------------------ G3
def g3():
    out = {}
    data, _ = synth_batch(2, 1024, first=4)
    ori = data[:, :, :3].transpose(1, 2).contiguous()  # [B,3,N]
    out['ori'] = ori
    for C in (16, 192):
        g = torch.Generator('cpu').manual_seed(100 + C)
        pick = torch.stack([torch.randperm(1024, generator=g)[:C] for _ in range(2)])
        central = torch.gather(ori, 2, pick[:, None, :].expand(2, 3, C)).contiguous()
        P = ((torch.rand(2, C, 3, generator=g) * 2 - 1) * 0.55).requires_grad_()
        sig = (0.1 + torch.rand(2, C, generator=g) * 1.1).requires_grad_()
        up = torch.randn(2, 3, 1024, generator=g)
        att = HiT_ADV.__new__(HiT_ADV)
        att.central_num = C
        ker = att.kernel_density(central, ori, sig)
        num = torch.zeros_like(ori)
        den = torch.zeros(2, 1, 1024)
        for j in range(C):  # the reference's loop body (HiT_ADV.py:170-175) driven from here
            num = num + (ori + P[:, j, :].unsqueeze(dim=2)) * ker[:, j, :].unsqueeze(dim=1)
            den = den + ker[:, j, :].unsqueeze(1)
        adv = num / den
        (adv * up).sum().backward()
        pre = 'c%d_' % C
        out.update({pre + 'central': central, pre + 'P': P, pre + 'sigma': sig, pre + 'upstream': up,
                    pre + 'adv': adv, pre + 'grad_P': P.grad, pre + 'grad_sigma': sig.grad})
        if C == 16:
            out[pre + 'ker'] = ker
        out[pre + 'tl_batch'] = att.transformation_loss(adv, P, sig, batch_avg=True)
        out[pre + 'tl_each'] = att.transformation_loss(adv, P, sig, batch_avg=False)
        kstd = torch.rand(2, C, 1, generator=g)
        out[pre + 'central_kappa'] = kstd
        out[pre + 'hide'] = att.curv_std_loss(sig, kstd, 1.2, 0.1)
    save('g3_deform.npz', out)


# ------------------------------------------------------------------ G4
def g4():
    data, _ = synth_batch(3, 1024, first=6)
    xyz = data[:, :, :3].contiguous()
    att = HiT_ADV.__new__(HiT_ADV)
    torch.manual_seed(42)
    start = torch.randint(0, 1024, (3,), dtype=torch.long)
    torch.manual_seed(42)
    idx = att.farthest_point_sample(xyz, 256)
    assert (idx[:, 0] == start).all()
    save('g4_fps.npz', dict(xyz=xyz, start=start, idx=idx))


# ------------------------------------------------------------------ G5
def run_reference_attack(model, data, target, seed, **hp):
    """Run HiT_ADV.attack while recording the arguments of its own helper calls."""
    att = HiT_ADV(model, adv_func=adv_utils.UntargetedLogitsAdvLoss(kappa=30.), **hp)
    trace = dict(P=[], sigma=[], adv=[], logits=[], adv_loss=[])
    cap = {}
    orig_tl, orig_kd, orig_hide = att.transformation_loss, att.kernel_density, att.curv_std_loss

    def tl(adv_data, perturb_mat, gauss_delta, batch_avg=True):
        if not batch_avg:  # called exactly once per iteration with batch_avg=False (:195)
            trace['P'].append(perturb_mat.detach().clone())
            trace['sigma'].append(gauss_delta.detach().clone())
            trace['adv'].append(adv_data.detach().clone())
        return orig_tl(adv_data, perturb_mat, gauss_delta, batch_avg)

    def kd(central_points, pc, delta):
        cap['central'] = central_points.detach().clone()
        return orig_kd(central_points, pc, delta)

    def hide(gauss_delta, central_kappa_std, max_delta, min_delta):
        cap['central_kappa'] = central_kappa_std.detach().clone()
        return orig_hide(gauss_delta, central_kappa_std, max_delta, min_delta)

    class Adv(torch.nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, logits, targets):
            v = self.inner(logits, targets)
            trace['logits'].append(logits.detach().clone())
            trace['adv_loss'].append(v.detach().clone())
            return v

    att.transformation_loss, att.kernel_density, att.curv_std_loss = tl, kd, hide
    att.adv_func = Adv(att.adv_func)
    torch.manual_seed(seed)
    with redirect_stdout(io.StringIO()) as log:
        best, succ = att.attack(data, target)
    out = {k: torch.stack(v) for k, v in trace.items()}
    out.update(cap)
    out['best'] = best
    out['success_num'] = int(succ)
    lb = [l for l in log.getvalue().splitlines() if l.startswith('lower_bound is')]
    out['lower_bound_text'] = np.array(lb[-1] if lb else '')
    return out


def g5():
    model = toy_victim(3)
    data, _ = synth_batch(4, 256, first=20)
    with torch.no_grad():
        target = model(data[:, :, :3].transpose(1, 2).contiguous()).argmax(1)  # clean-correct labels
    hp = dict(attack_lr=1e-2, central_num=16, total_central_num=32, init_weight=10., max_weight=80.,
              binary_step=2, num_iter=10, cd_weight=1e-4, ker_weight=1., hide_weight=1.,
              curv_loss_knn=8, max_sigm=1.2, min_sigm=0.1, budget=0.55)
    out = run_reference_attack(model, data, target, seed=11, **hp)
    out.update(data=data, target=target, seed=11,
               **{'w_' + k: v for k, v in model.state_dict().items()})
    out.update({'hp_' + k: v for k, v in hp.items()})
    save('g5_attack.npz', out)


def g5b():
    model = toy_victim(5)
    data, _ = synth_batch(2, 1024, first=30)
    with torch.no_grad():
        target = model(data[:, :, :3].transpose(1, 2).contiguous()).argmax(1)
    hp = dict(attack_lr=1e-2, central_num=192, total_central_num=256, init_weight=10.,
              max_weight=80., binary_step=1, num_iter=5, cd_weight=1e-4, ker_weight=1.,
              hide_weight=1., curv_loss_knn=16, max_sigm=1.2, min_sigm=0.1, budget=0.55)
    out = run_reference_attack(model, data, target, seed=13, **hp)
    out['adv'] = out['adv'][-1]  # keep only the last iterate of the wide run
    out.update(data=data, target=target, seed=13,
               **{'w_' + k: v for k, v in model.state_dict().items()})
    out.update({'hp_' + k: v for k, v in hp.items()})
    save('g5b_attack_wide.npz', out)


def watch_bookkeeping(run):
    """Run ``run()`` (a call of the reference's HiT_ADV.attack) while watching the local variables of its frame through
    ``sys.settrace``: nothing of the reference is edited or wrapped, its own bookkeeping is read as it runs.  Returns what
    ``run`` returned plus, per binary step, (lower_bound, upper_bound, scale_const, o_bestdist, o_bestscore, bestdist,
    bestscore) as they stand when the step's bisection is done, the same after the failure fill, and for every sample the
    (step, iteration) at which its overall best record was last replaced (-1, -1: never)."""
    rec = dict(steps=[], taken=None, final=None)
    seen = dict(at=None, obd=None)

    def snap(loc):
        f = lambda v: np.array(v.detach().cpu().numpy() if torch.is_tensor(v) else v, dtype=np.float64).copy()  # noqa: E731
        return dict(lower=f(loc['lower_bound']), upper=f(loc['upper_bound']), scale_const=f(loc['scale_const']),
                    o_bestdist=f(loc['o_bestdist']), o_bestscore=f(loc['o_bestscore']),
                    bestdist=f(loc['bestdist']), bestscore=f(loc['bestscore']))

    def local(frame, event, arg):
        loc = frame.f_locals
        if 'o_bestdist' not in loc or 'binary_step' not in loc:
            return local
        at = (loc['binary_step'], loc.get('iteration', -1))
        obd = np.asarray(loc['o_bestdist'], dtype=np.float64)
        if rec['taken'] is None:
            rec['taken'] = -np.ones((obd.shape[0], 2), dtype=np.int64)
            seen['obd'] = obd.copy()
        changed = obd != seen['obd']  # replaced while the reference was inside iteration `at`
        if changed.any() and event != 'return':
            rec['taken'][changed] = at
            seen['obd'] = obd.copy()
        if seen['at'] is not None and at[0] != seen['at'][0]:  # a new binary step begins: the previous one's bisection is done
            rec['steps'].append(snap(loc))
        seen['at'] = at
        if event == 'return':
            rec['steps'].append(snap(loc))  # the last step's bisection ...
            rec['final'] = dict(o_bestdist=np.array(loc['o_bestdist'], dtype=np.float64).copy())  # ... and the failure fill
        return local

    def tracer(frame, event, arg):
        code = frame.f_code
        if code.co_name == 'attack' and code.co_filename.endswith(os.path.join('ShapeAttack', 'HiT_ADV.py')):
            return local
        return None

    sys.settrace(tracer)
    try:
        out = run()
    finally:
        sys.settrace(None)
    return out, rec


def g5c():
    """Row a16 over a long horizon: toy victim, B=4, N=256, C=16, binary_step=10 x num_iter=20."""
    model = toy_victim(7)
    data, _ = synth_batch(4, 256, first=40)
    with torch.no_grad():
        target = model(data[:, :, :3].transpose(1, 2).contiguous()).argmax(1)
    hp = dict(attack_lr=1e-2, central_num=16, total_central_num=32, init_weight=10., max_weight=80.,
              binary_step=10, num_iter=20, cd_weight=1e-4, ker_weight=1., hide_weight=1.,
              curv_loss_knn=8, max_sigm=1.2, min_sigm=0.1, budget=0.55)
    att = HiT_ADV(model, adv_func=adv_utils.UntargetedLogitsAdvLoss(kappa=30.), **hp)

    def run():
        torch.manual_seed(17)
        with redirect_stdout(io.StringIO()):
            return att.attack(data, target)
    (best, succ), rec = watch_bookkeeping(run)
    assert len(rec['steps']) == hp['binary_step']
    out = dict(best=best, success_num=int(succ), data=data, target=target, seed=17,
               taken_step=rec['taken'][:, 0], taken_iter=rec['taken'][:, 1], final_o_bestdist=rec['final']['o_bestdist'],
               **{'w_' + k: v for k, v in model.state_dict().items()})
    for name in ('lower', 'upper', 'scale_const', 'o_bestdist', 'o_bestscore', 'bestdist', 'bestscore'):
        out['step_' + name] = np.stack([s_[name] for s_ in rec['steps']])
    out.update({'hp_' + k: v for k, v in hp.items()})
    save('g5c_attack_long.npz', out)



def g5d(steps=2, iters=50, name='g5d_attack_pointnet.npz', p_every=10, logits_every=1, tag='g5d'):
    """The headline configuration's victim at a horizon that means something: the imported reference's
    HiT_ADV.attack on cfg2's shape -- seeded PointNetFeatureModel (the reference's own class) with shaken BatchNorm
    statistics, B=32, N=1024, C=192, T=256, eval.py's hyper-parameters, binary_step=2 x num_iter=50.  The victim seed /
    shake were chosen (tools/explore_success.py, on the GPU build) so that some clouds succeed in both steps, some in the
    first only and some never: every branch of the best-tracking and of the bisection fires.  Stored: per iteration the
    logits, adv_loss and the per-sample distance the bookkeeping compares; (P, sigma) every tenth iteration; the deformed
    clouds at iterations 0 and 49 of each step; the reference's own bookkeeping variables after each step (sys.settrace);
    the returned clouds and success count.  Weights are NOT stored: they are torch.manual_seed(0) + the default
    initialisation + Dataset.synthetic.shake_bn(seed=2), a checksum of every tensor is."""
    from model.feature_models import PointNetFeatureModel
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from hit_adv_amd.Dataset.synthetic import shake_bn
    shake = dict(seed=2, mean_std=0.05, var_spread=0.2)
    torch.manual_seed(0)
    model = PointNetFeatureModel(40, normal_channel=False)
    model.eval()
    shake_bn(model, **shake)
    first, seed = 7000, 21
    data, _ = synth_batch(32, 1024, first=first)
    with torch.no_grad():
        clean = model(data[:, :, :3].transpose(1, 2).contiguous())[0]
    target = clean.argmax(1)
    hp = dict(attack_lr=1e-2, central_num=192, total_central_num=256, init_weight=10., max_weight=80.,
              binary_step=steps, num_iter=iters, cd_weight=1e-4, ker_weight=1., hide_weight=1.,
              curv_loss_knn=16, max_sigm=1.2, min_sigm=0.1, budget=0.55)
    att = HiT_ADV(model, adv_func=adv_utils.UntargetedLogitsAdvLoss(kappa=30.), **hp)
    trace = dict(P=[], sigma=[], adv=[], logits=[], adv_loss=[], dist_val=[])
    cap = {}
    orig_tl, orig_kd = att.transformation_loss, att.kernel_density
    count = dict(it=0)

    def tl(adv_data, perturb_mat, gauss_delta, batch_avg=True):
        out = orig_tl(adv_data, perturb_mat, gauss_delta, batch_avg)
        if not batch_avg:  # exactly once per iteration (:195): the distance the best-tracking compares
            it = count['it'] % iters
            trace['dist_val'].append(out.detach().clone())
            if it % p_every == 0 or it == iters - 1:
                trace['P'].append(perturb_mat.detach().clone())
                trace['sigma'].append(gauss_delta.detach().clone())
            if it in (0, iters - 1):
                trace['adv'].append(adv_data.detach().clone())
            count['it'] += 1
        return out

    def kd(central_points, pc, delta):
        cap['central'] = central_points.detach().clone()
        return orig_kd(central_points, pc, delta)

    class Adv(torch.nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, logits, targets):
            v = self.inner(logits, targets)
            it = len(trace['adv_loss']) % iters
            if it % logits_every == 0 or it == iters - 1:
                trace['logits'].append(logits.detach().clone())
            if logits_every != 1:  # the long trace keeps every iteration's prediction and margin instead of every logit
                trace.setdefault('pred', []).append(logits.argmax(1).to(torch.int16))
                top2 = logits.detach().topk(2, dim=1).values
                trace.setdefault('margin', []).append(top2[:, 0] - top2[:, 1])
            trace['adv_loss'].append(v.detach().clone())
            return v

    att.transformation_loss, att.kernel_density = tl, kd
    att.adv_func = Adv(att.adv_func)

    def run():
        torch.manual_seed(seed)
        with redirect_stdout(io.StringIO()):
            return att.attack(data, target)
    import time
    t0 = time.time()
    (best, succ), rec = watch_bookkeeping(run)
    print("%s: reference attack %d x %d at B=32 took %.0f s; success %d / 32" % (tag, steps, iters, time.time() - t0, int(succ)))
    assert len(rec['steps']) == steps and count['it'] == steps * iters
    out = {k: torch.stack(v) for k, v in trace.items()}
    out.update(cap)
    out.update(best=best, success_num=int(succ), target=target, clean_logits=clean, seed=seed, first=first, model_seed=0,
               shake_seed=shake['seed'], shake_mean_std=shake['mean_std'], shake_var_spread=shake['var_spread'],
               taken_step=rec['taken'][:, 0], taken_iter=rec['taken'][:, 1], final_o_bestdist=rec['final']['o_bestdist'],
               kept_iterations=np.array([i for i in range(iters) if i % p_every == 0 or i == iters - 1]),
               weight_checksum=np.array([float(v.double().abs().sum()) for v in model.state_dict().values()]))
    for field in ('lower', 'upper', 'scale_const', 'o_bestdist', 'o_bestscore', 'bestdist', 'bestscore'):
        out['step_' + field] = np.stack([s_[field] for s_ in rec['steps']])
    out.update({'hp_' + k: v for k, v in hp.items()})
    if logits_every != 1:
        out['kept_logit_iterations'] = np.array([i for i in range(iters) if i % logits_every == 0 or i == iters - 1])
    save(name, out)


def g5e():
    """The headline's own inner horizon: the same victim, clouds and seed as g5d, ONE binary
    step of num_iter=500 -- what eval.py:126-133 runs ten times per batch.  ~37 min of this container's 8 cores.  Stored: the
    prediction, the top-two margin, adv_loss and the compared distance at EVERY iteration; the logits every 25th; (P, sigma)
    every 50th; the deformed clouds at iterations 0 and 499; the reference's bookkeeping after the step and what it returns."""
    g5d(steps=1, iters=500, name='g5e_attack_pointnet_500.npz', p_every=50, logits_every=25, tag='g5e')


# ------------------------------------------------------------------ G6
def g6():
    g = torch.Generator('cpu').manual_seed(21)
    logits = torch.randn(6, 40, generator=g) * 5
    tgt = torch.randint(0, 40, (6,), generator=g)
    out = dict(logits=logits, target=tgt)
    for kappa in (0., 30.):
        out['untargeted_k%d' % kappa] = adv_utils.UntargetedLogitsAdvLoss(kappa)(logits, tgt)
        out['targeted_k%d' % kappa] = adv_utils.LogitsAdvLoss(kappa)(logits, tgt)
    out['cross_entropy'] = adv_utils.CrossEntropyAdvLoss()(logits, tgt)
    ori = torch.randn(2, 3, 128, generator=g)
    pc = ori + 0.2 * torch.randn(2, 3, 128, generator=g)
    nrm = torch.nn.functional.normalize(torch.randn(2, 3, 128, generator=g), dim=1)
    out.update(pc=pc, ori=ori, normal=nrm)
    out['clip_l2'] = clip_utils.ClipPointsL2(budget=1.5)(pc, ori)
    out['clip_linf'] = clip_utils.ClipPointsLinf(budget=0.18)(pc, ori)
    out['project_inner'] = clip_utils.ProjectInnerPoints()(pc.clone(), ori, nrm)
    out['project_clip'] = clip_utils.ProjectInnerClipLinf(budget=0.18)(pc.clone(), ori, nrm)
    save('g6_adv_clip.npz', out)


# ------------------------------------------------------------------ G7
def g7():
    model = toy_victim(9)
    data, _ = synth_batch(2, 256, first=40)
    xyz = data[:, :, :3].contiguous()
    with torch.no_grad():
        clean = model(xyz.transpose(1, 2).contiguous()).argmax(1)
    target = (clean + 1) % 40  # targeted attack towards another class
    advs = []
    clip = clip_utils.ClipPointsLinf(budget=0.18)

    def recording_clip(pc, ori_pc):
        r = clip(pc, ori_pc)
        advs.append(r.detach().clone())
        return r

    att = CWKNN(model, adv_utils.LogitsAdvLoss(kappa=15.), dist_utils.ChamferkNNDist(),
                recording_clip, attack_lr=1e-2, num_iter=10)
    torch.manual_seed(17)
    with redirect_stdout(io.StringIO()):
        final, succ = att.attack(xyz, target)
    out = dict(data=xyz, target=target, seed=17, adv_trace=torch.stack(advs), final=final,
               success_num=int(succ), **{'w_' + k: v for k, v in model.state_dict().items()})
    save('g7_cwknn.npz', out)


# ------------------------------------------------------------------ G9
def g9():
    model = toy_victim(12)
    data, _ = synth_batch(3, 256, first=50)
    xyz = data[:, :, :3].contiguous()
    with torch.no_grad():
        clean = model(xyz.transpose(1, 2).contiguous()).argmax(1)
    target = (clean + 3) % 40
    advs = []
    clip = clip_utils.ClipPointsLinf(budget=0.18)

    def recording_clip(pc, ori_pc):
        r = clip(pc, ori_pc)
        advs.append(r.detach().clone())
        return r

    att = CWPerturb(model, adv_utils.LogitsAdvLoss(kappa=5.), dist_utils.L2Dist(), attack_lr=1e-2, init_weight=10.,
                    max_weight=80., binary_step=3, num_iter=10, clip_func=recording_clip)
    torch.manual_seed(23)
    with redirect_stdout(io.StringIO()):
        best, succ = att.attack(xyz, target)
    save('g9_cwperturb.npz', dict(data=xyz, target=target, seed=23, adv_trace=torch.stack(advs), best=best,
                                  success_num=int(succ), **{'w_' + k: v for k, v in model.state_dict().items()}))


# ------------------------------------------------------------------ G10
def g10():
    import argparse
    import model.dgcnn_cls as ref_dgcnn

    class _TorchOnCpu:  # dgcnn_cls.py:25 hard-codes torch.device('cuda'); everything else passes through
        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def device(*a, **k):
            return torch.device('cpu')

    ref_dgcnn.torch = _TorchOnCpu()
    torch.manual_seed(31)
    m = ref_dgcnn.DGCNN_cls(argparse.Namespace(k=5, emb_dims=1024, dropout=0.2), output_channels=40).eval()
    data, _ = synth_batch(2, 256, first=60)
    x = data[:, :, :3].transpose(1, 2).contiguous().requires_grad_()
    logits = m(x)
    w = torch.randn(2, 40, generator=torch.Generator().manual_seed(4))
    (logits * w).sum().backward()
    save('g10_dgcnn.npz', dict(x=x.detach(), logits=logits, grad_w=w, grad_x=x.grad, seed=31,
                               knn_layer1=ref_dgcnn.knn(x.detach(), 5),
                               edge_layer1=ref_dgcnn.get_graph_feature(x.detach(), k=5)))


# ------------------------------------------------------------------ G11
def g11():
    from model import pointnet2_utils as pu
    from model.pointnet2_cls_ssg import get_model
    torch.manual_seed(37)
    m = get_model(40, normal_channel=False).eval()
    data, _ = synth_batch(2, 1024, first=70)
    x = data[:, :, :3].transpose(1, 2).contiguous().requires_grad_()
    torch.manual_seed(41)  # pins the two randint draws of the forward (sa1, sa2 FPS starts)
    logits, _ = m(x)
    w = torch.randn(2, 40, generator=torch.Generator().manual_seed(4))
    (logits * w).sum().backward()
    pts = x.detach().transpose(1, 2).contiguous()
    torch.manual_seed(41)
    fps1 = pu.farthest_point_sample(pts, 512)
    new_xyz = pu.index_points(pts, fps1)
    ball1 = pu.query_ball_point(0.2, 32, pts, new_xyz)
    save('g11_pointnet2.npz', dict(x=x.detach(), logits=logits, grad_w=w, grad_x=x.grad, init_seed=37, fwd_seed=41,
                                   fps1=fps1, ball1=ball1))


# ------------------------------------------------------------------ G12
def g12():
    import argparse
    from model.pct_cls import Pct
    from util.other_utils import fps
    torch.manual_seed(43)
    m = Pct(argparse.Namespace(dropout=0.2), output_channels=40).eval()
    data, _ = synth_batch(2, 1024, first=80)
    x = data[:, :, :3].transpose(1, 2).contiguous().requires_grad_()
    torch.manual_seed(47)
    logits = m(x)
    w = torch.randn(2, 40, generator=torch.Generator().manual_seed(4))
    (logits * w).sum().backward()
    torch.manual_seed(47)
    fps1 = fps(x.detach().transpose(1, 2).contiguous(), 512)
    save('g12_pct.npz', dict(x=x.detach(), logits=logits, grad_w=w, grad_x=x.grad, init_seed=43, fwd_seed=47, fps1=fps1))


# ------------------------------------------------------------------ G13
def g13():
    from CW.AOF import CWAOF
    # torch.symeig only raises in torch 2.x; torch.linalg.eigh is the same decomposition (ascending eigenvalues)
    torch.symeig = lambda L, eigenvectors=True: torch.linalg.eigh(L)
    model = toy_victim(14)
    data, _ = synth_batch(2, 256, first=90)
    xyz = data[:, :, :3].contiguous()
    with torch.no_grad():
        label = model(xyz.transpose(1, 2).contiguous()).argmax(1)
    advs = []
    clip = clip_utils.ClipPointsLinf(budget=0.18)

    def recording_clip(pc, ori_pc):
        r = clip(pc, ori_pc)
        advs.append(r.detach().clone())
        return r

    att = CWAOF(model, adv_utils.UntargetedLogitsAdvLoss(kappa=30.), dist_utils.L2Dist(), attack_lr=1e-2,
                binary_step=2, num_iter=5, GAMMA=0.25, low_pass=40, clip_func=recording_clip)
    torch.manual_seed(29)
    with redirect_stdout(io.StringIO()):
        final, succ = att.attack(xyz, label)
    save('g13_aof.npz', dict(data=xyz, target=label, seed=29, adv_trace=torch.stack(advs[:10]), final=final,
                             success_num=int(succ), **{'w_' + k: v for k, v in model.state_dict().items()}))


# ------------------------------------------------------------------ G8
def g8():
    shapes = {}
    from model import feature_models
    m = feature_models.PointNetFeatureModel(40, normal_channel=False)
    shapes['pointnet'] = {k: list(v.shape) for k, v in m.state_dict().items()}
    shapes['pointnet_param_count'] = sum(p.numel() for p in m.parameters())
    try:
        from model.pointnet2_cls_ssg import get_model
        m = get_model(40, normal_channel=False)
        shapes['pointnet++'] = {k: list(v.shape) for k, v in m.state_dict().items()}
    except Exception as e:  # noqa: BLE001
        shapes['pointnet++_error'] = repr(e)
    try:
        import argparse
        from model.dgcnn_cls import DGCNN_cls
        m = DGCNN_cls(argparse.Namespace(k=5, emb_dims=1024, dropout=0.2), output_channels=40)
        shapes['dgcnn'] = {k: list(v.shape) for k, v in m.state_dict().items()}
    except Exception as e:  # noqa: BLE001
        shapes['dgcnn_error'] = repr(e)
    try:
        import argparse
        from model.pct_cls import Pct
        m = Pct(argparse.Namespace(dropout=0.2), output_channels=40)
        shapes['pct'] = {k: list(v.shape) for k, v in m.state_dict().items()}
    except Exception as e:  # noqa: BLE001
        shapes['pct_error'] = repr(e)
    with open(os.path.join(HERE, 'g8_state_dicts.json'), 'w') as f:
        json.dump(shapes, f, indent=0, sort_keys=True)
    print('wrote g8_state_dicts.json', {k: (len(v) if isinstance(v, dict) else v)
                                        for k, v in shapes.items()})

# ------------------------------------------------------------------ G14-G18: the remaining un-weighted CW variants
class ToyAE(torch.nn.Module):
    """Stand-in auto-encoder ([B,3,K] -> [B,3,K]); the reference ships none.  Weights are stored in the fixture."""

    def __init__(self):
        super().__init__()
        self.enc = torch.nn.Conv1d(3, 8, 1)
        self.dec = torch.nn.Conv1d(8, 3, 1)

    def forward(self, x):
        return x + 0.1 * self.dec(torch.tanh(self.enc(x)))


def toy_ae(seed):
    torch.manual_seed(seed)
    return ToyAE().eval()


def _runner_up(model, xyz):
    """(clean prediction, second-best class): an easy target so that the few iterations of a fixture reach it."""
    with torch.no_grad():
        top2 = model(xyz.transpose(1, 2).contiguous()).topk(2, dim=1).indices
    return top2[:, 0].contiguous(), top2[:, 1].contiguous()


def _recording_clip(budget, sink):
    clip = clip_utils.ClipPointsLinf(budget=budget)

    def rec(pc, ori_pc):
        r = clip(pc, ori_pc)
        sink.append(r.detach().clone())
        return r
    return rec


def _weights(model, prefix):
    return {prefix + k: v for k, v in model.state_dict().items()}


def g14():
    from CW.PerturbT import CWPerturbT
    model = toy_victim(21)
    data, _ = synth_batch(3, 256, first=110)
    xyz = data[:, :, :3].contiguous()
    clean, target = _runner_up(model, xyz)
    advs = []
    att = CWPerturbT(model, adv_utils.LogitsAdvLoss(kappa=0.), dist_utils.L2Dist(), attack_lr=3e-2, init_weight=10.,
                     max_weight=80., binary_step=3, num_iter=10, clip_func=_recording_clip(0.3, advs))
    torch.manual_seed(31)
    with redirect_stdout(io.StringIO()):
        best, succ = att.attack(xyz, target)
    save('g14_cwperturbt.npz', dict(data=xyz, target=target, seed=31, adv_trace=torch.stack(advs), best=best,
                                    success_num=int(succ), **_weights(model, 'w_')))


def _family_fixture(name, cls_path, seed, spectral, ae, targeted, gamma, num_iter=5, lr=3e-2):
    import importlib
    torch.symeig = lambda L, eigenvectors=True: torch.linalg.eigh(L)
    mod_name, cls_name = cls_path.rsplit('.', 1)
    cls = getattr(importlib.import_module(mod_name), cls_name)
    model, aem = toy_victim(seed), toy_ae(seed + 1)
    data, _ = synth_batch(2, 256, first=120 + seed)
    xyz = data[:, :, :3].contiguous()
    clean, second = _runner_up(model, xyz)
    target = second if targeted else clean
    advs = []
    clipf = _recording_clip(0.3, advs)
    adv_f = adv_utils.LogitsAdvLoss(kappa=0.) if targeted else adv_utils.UntargetedLogitsAdvLoss(kappa=30.)
    kw = dict(attack_lr=lr, binary_step=2, num_iter=num_iter, GAMMA=gamma, clip_func=clipf)
    if spectral:
        kw['low_pass'] = 40
    args = (model, aem, adv_f, dist_utils.L2Dist()) if ae else (model, adv_f, dist_utils.L2Dist())
    att = cls(*args, **kw)
    torch.manual_seed(seed + 2)
    with redirect_stdout(io.StringIO()):
        out = att.attack(xyz, target, clean) if targeted else att.attack(xyz, target)
    bestdist, final, succ = out
    for p in model.parameters():
        p.requires_grad = True
    save(name, dict(data=xyz, target=target, y_truth=clean, seed=seed + 2, adv_trace=torch.stack(advs[:10]), num_iter=num_iter, lr=lr,
                    bestdist=bestdist, final=final, success_num=int(succ), gamma=gamma,
                    **_weights(model, 'w_'), **_weights(aem, 'ae_')))


def g15():
    _family_fixture('g15_advpc.npz', 'CW.AdvPC.CWAdvPC', 40, spectral=False, ae=True, targeted=True, gamma=0.5, num_iter=10,
                    lr=6e-2)


def g16():
    _family_fixture('g16_uadvpc.npz', 'CW.UAdvPC.CWUAdvPC', 50, spectral=False, ae=True, targeted=False, gamma=0.5)


def g17():
    _family_fixture('g17_taof.npz', 'CW.TAOF.CWTAOF', 60, spectral=True, ae=False, targeted=True, gamma=0.25)


def g18():
    _family_fixture('g18_uaeaof.npz', 'CW.UAEAOF.CWUAEAOF', 70, spectral=True, ae=True, targeted=False, gamma=0.25)


# ------------------------------------------------------------------ G19-G21: the point / cluster / object adding attacks
def _add_setup(seed, first):
    model = toy_victim(seed)
    data, _ = synth_batch(2, 256, first=first)
    xyz = data[:, :, :3].contiguous()
    return model, xyz, _runner_up(model, xyz)[1]


def g19():
    from CW.Add import CWAdd, get_critical_points
    model, xyz, target = _add_setup(80, 200)
    cri = get_critical_points(model, xyz.transpose(1, 2).contiguous(), target, 32)
    out = {}
    for tag, dist in (('chamfer', dist_utils.ChamferDist(method='adv2ori')), ('hausdorff', dist_utils.HausdorffDist(method='adv2ori'))):
        att = CWAdd(model, adv_utils.LogitsAdvLoss(kappa=0.), dist, attack_lr=6e-2, init_weight=5., max_weight=40.,
                    binary_step=3, num_iter=12, num_add=32)
        torch.manual_seed(83)
        with redirect_stdout(io.StringIO()):
            bestdist, final, succ = att.attack(xyz, target)
        out.update({tag + '_bestdist': bestdist, tag + '_final': final, tag + '_success_num': int(succ)})
    save('g19_cwadd.npz', dict(data=xyz, target=target, seed=83, critical=cri, **out, **_weights(model, 'w_')))


def g20():
    from CW.Add_Cluster import CWAddClusters
    model, xyz, target = _add_setup(90, 210)
    att = CWAddClusters(model, adv_utils.LogitsAdvLoss(kappa=0.), dist_utils.FarChamferDist(num_add=3, chamfer_weight=0.1),
                        attack_lr=3e-2, init_weight=5., max_weight=30., binary_step=3, num_iter=8, num_add=3, cl_num_p=16)
    torch.manual_seed(93)
    np.random.seed(94)
    with redirect_stdout(io.StringIO()):
        centers = att._init_centers(xyz.transpose(1, 2).contiguous(), target)
    torch.manual_seed(93)
    np.random.seed(94)
    with redirect_stdout(io.StringIO()):
        bestdist, final, succ = att.attack(xyz, target)
    save('g20_cwaddclusters.npz', dict(data=xyz, target=target, seed=93, np_seed=94, centers=centers, bestdist=bestdist,
                                       final=final, success_num=int(succ), **_weights(model, 'w_')))


def g21():
    from CW.Add_Objects import CWAddObjects
    model, xyz, target = _add_setup(100, 220)
    obj = synth_batch(1, 128, first=230)[0][0, :, :3].numpy().astype(np.float64)
    np.random.seed(104)
    att = CWAddObjects(model, adv_utils.LogitsAdvLoss(kappa=0.), dist_utils.L2ChamferDist(num_add=2, chamfer_weight=0.2),
                       obj.copy(), attack_lr=6e-2, init_weight=1., max_weight=40., binary_step=3, num_iter=14, num_add=2,
                       obj_num_p=24, scaling=0.3)
    object_pc = att.object_pc.copy()
    np.random.seed(105)
    centers = att._init_centers(xyz.transpose(1, 2).contiguous(), target)
    torch.manual_seed(103)
    np.random.seed(105)
    with redirect_stdout(io.StringIO()):
        bestdist, final, succ = att.attack(xyz, target)
    save('g21_cwaddobjects.npz', dict(data=xyz, target=target, seed=103, np_seed=104, obj=obj, object_pc=object_pc,
                                      centers=centers,
                                      bestdist=bestdist, final=final, success_num=int(succ), **_weights(model, 'w_')))


# ------------------------------------------------------------------ G22: the remaining distance operators
def g22():
    data, _ = synth_batch(3, 256, first=300)
    ori = data[:, :, :3].transpose(1, 2).contiguous()
    normal = data[:, :, 3:].transpose(1, 2).contiguous()
    g = torch.Generator().manual_seed(5)
    adv = (ori + 0.03 * torch.randn(ori.shape, generator=g)).requires_grad_()
    w = torch.tensor([0.5, 2.0, 1.25])
    out = dict(ori=ori, normal=normal, adv=adv.detach(), weights=w)
    lap = dist_utils.LaplacianDist(k=6)
    val, idx = lap.KNN_indices(ori)
    out['lap_knn_value'], out['lap_knn_idx'] = val, idx
    d = lap(adv, ori, idx, weights=w, batch_avg=False)
    out['lap'] = d
    out['lap_grad'], = torch.autograd.grad(d.sum(), adv)
    clusters = (0.2 * torch.randn(3, 4, 16, 3, generator=g)).requires_grad_()
    d = dist_utils.FarthestDist()(clusters, weights=w, batch_avg=False)
    out['clusters'], out['far'] = clusters.detach(), d
    out['far_grad'], = torch.autograd.grad(d.sum(), clusters)
    added = (ori[:, :, :64].transpose(1, 2) + 0.05 * torch.randn(3, 64, 3, generator=g)).contiguous().requires_grad_()
    d = dist_utils.FarChamferDist(num_add=4, chamfer_weight=0.1)(added, ori.transpose(1, 2).contiguous(), weights=w, batch_avg=False)
    out['added'], out['farchamfer'] = added.detach(), d
    out['farchamfer_grad'], = torch.autograd.grad(d.sum(), added)
    obj0 = 0.1 * torch.randn(3, 4, 16, 3, generator=g)
    obj1 = (obj0 + 0.01 * torch.randn(3, 4, 16, 3, generator=g)).requires_grad_()
    d = dist_utils.L2ChamferDist(num_add=4, chamfer_weight=0.2)(added, ori.transpose(1, 2).contiguous(), obj1, obj0,
                                                                 weights=w, batch_avg=False)
    out['obj0'], out['obj1'], out['l2chamfer'] = obj0, obj1.detach(), d
    out['curv'] = dist_utils.CurvDist(curv_loss_knn=2)(ori, adv.detach(), normal)
    save('g22_dist_more.npz', out)

# ------------------------------------------------------------------ G23: dataset readers on a tiny synthetic tree
def g23():
    import argparse
    import tempfile
    from Dataset.ModelNet import ModelNetDataLoader
    from Dataset.ShapeNetDataLoader import PartNormalDataset
    rng = np.random.RandomState(7)
    files = {}
    shapes = [('airplane', 'airplane_0001'), ('airplane', 'airplane_0002'), ('night_stand', 'night_stand_0001')]
    files['modelnet40_shape_names.txt'] = 'airplane\nnight_stand\n'
    files['modelnet40_train.txt'] = 'airplane_0001\n'
    files['modelnet40_test.txt'] = 'airplane_0002\nnight_stand_0001\n'
    for cls, sid in shapes:
        pts = rng.randn(40, 6)
        files['%s/%s.txt' % (cls, sid)] = ''.join(','.join('%.6f' % v for v in row) + '\n' for row in pts)
    files['synsetoffset2category.txt'] = 'Airplane\t02691156\nBag\t02773838\n'
    tokens = {'02691156': ['aaa1', 'aaa2', 'aaa3'], '02773838': ['bbb1', 'bbb2']}
    split = {'train': ['shape_data/02691156/aaa1', 'shape_data/02773838/bbb1'], 'val': ['shape_data/02691156/aaa2'],
             'test': ['shape_data/02691156/aaa3', 'shape_data/02773838/bbb2']}
    for k, v in split.items():
        files['train_test_split/shuffled_%s_file_list.json' % k] = json.dumps(v)
    for syn, toks in tokens.items():
        for t in toks:
            pts = np.concatenate([rng.randn(30, 6), rng.randint(0, 4, (30, 1))], axis=1)
            files['%s/%s.txt' % (syn, t)] = ''.join(' '.join('%.6f' % v for v in row) + '\n' for row in pts)
    out = {}
    with tempfile.TemporaryDirectory() as root:
        for rel, text in files.items():
            path = os.path.join(root, rel)
            os.makedirs(os.path.dirname(path), exist_ok=True)
            open(path, 'w').write(text)
        with redirect_stdout(io.StringIO()):
            for tag, uniform, normals, process in (('plain', False, True, False), ('fps', True, False, False),
                                                   ('cached', False, True, True)):
                args = argparse.Namespace(num_point=16, use_uniform_sample=uniform, use_normals=normals, num_category=40)
                np.random.seed(11)
                ds = ModelNetDataLoader(root, args, split='test', process_data=process)
                items = [ds[i] for i in range(len(ds))]
                out['modelnet_%s_points' % tag] = np.stack([p for p, _ in items])
                out['modelnet_%s_labels' % tag] = np.array([l for _, l in items])
            for tag, sp, normals in (('test', 'test', True), ('trainval', 'trainval', False)):
                np.random.seed(13)
                ds = PartNormalDataset(root=root, npoints=12, split=sp, normal_channel=normals)
                items = [ds[i] for i in range(len(ds))]
                out['shapenet_%s_points' % tag] = np.stack([p for p, _ in items])
                out['shapenet_%s_labels' % tag] = np.array([l for _, l in items])
    with open(os.path.join(HERE, 'g23_dataset_tree.json'), 'w') as f:
        json.dump(files, f, indent=0, sort_keys=True)
    save('g23_datasets.npz', out)


# ------------------------------------------------------------------ G24
class CentreHead(torch.nn.Module):
    """The pre_head of fixture g24: subtracts every cloud's centroid (a parameter-free stand-in for the defence
    modules the reference passes as ``pre_head``; the test rebuilds it from this description)."""

    def forward(self, x):
        return x - x.mean(dim=2, keepdim=True)


def g24():
    from CW.UKNN import CWUKNN
    model = toy_victim(30)
    data, _ = synth_batch(3, 256, first=90)  # [B,N,6]: CWUKNN hands the normals to its clip (UKNN.py:120-122)
    with torch.no_grad():
        label = model(CentreHead()(data[:, :, :3].transpose(1, 2).contiguous())).argmax(1)
    out = dict(data=data, target=label, **{'w_' + k: v for k, v in model.state_dict().items()})
    for tag, dist, seed in (('l2', dist_utils.L2Dist(), 31), ('cham', dist_utils.ChamferkNNDist(), 33)):
        advs = []
        clip = clip_utils.ProjectInnerClipLinf(budget=0.3)

        def recording_clip(pc, ori_pc, normal):
            r = clip(pc, ori_pc, normal)
            advs.append(r.detach().clone())
            return r

        att = CWUKNN(model, adv_utils.UntargetedLogitsAdvLoss(kappa=15.), dist, recording_clip, attack_lr=3e-2,
                     num_iter=10, pre_head=CentreHead())
        torch.manual_seed(seed)
        with redirect_stdout(io.StringIO()) as log:
            final, succ = att.attack(data, label)
        out.update({tag + '_seed': seed, tag + '_adv_trace': torch.stack(advs), tag + '_final': final,
                    tag + '_success_num': int(succ),
                    tag + '_last_line': np.array(log.getvalue().strip().splitlines()[-1])})
    save('g24_cwuknn.npz', out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['g1', 'g2', 'g3', 'g4', 'g5', 'g5b', 'g5c', 'g6', 'g7', 'g8', 'g9', 'g10', 'g11', 'g12', 'g13',
                             'g14', 'g15', 'g16', 'g17', 'g18', 'g19', 'g20', 'g21', 'g22', 'g23', 'g24']
    for name in which:
        globals()[name]()
