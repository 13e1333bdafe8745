#!/usr/bin/env python3
"""Which seeded victims make SHORT attacks end with 0 < successes < B (both branches of the best-tracking /
bisection bookkeeping must fire in the parity fixtures and in the bench's other configurations).  Runs on the GPU box:

    python tools/explore_success.py pointnet|dgcnn|pointnet++|pct [...]

Prints one JSON line per (victim, shake) with the success counts; nothing here is a test or a measurement."""
import argparse
import json
import os
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from hit_adv_amd.Dataset.synthetic import shake_bn, sharpen, synth_batch  # noqa: E402


def cw_sweep(model, data, label, cfg, dev, gain, shake, logits, top2):
    from hit_adv_amd import CW
    from hit_adv_amd.util.adv_utils import LogitsAdvLoss, UntargetedLogitsAdvLoss
    from hit_adv_amd.util.clip_utils import ClipPointsLinf
    from hit_adv_amd.util.dist_utils import ChamferkNNDist, L2Dist
    torch.manual_seed(2)
    ae = bench.ToyAE().eval().to(dev)
    clip = ClipPointsLinf(budget=0.18)
    xyz = data[:, :, :3].contiguous()
    target = (label + 1) % cfg['classes']
    kw = dict(verbose=False)
    a = CW.CWAdvPC(model, ae, LogitsAdvLoss(kappa=0.), L2Dist(), clip_func=clip, binary_step=2, num_iter=60, **kw)
    k = CW.CWKNN(model, LogitsAdvLoss(kappa=15.), ChamferkNNDist(), clip, num_iter=300, **kw)
    f = CW.CWAOF(model, UntargetedLogitsAdvLoss(kappa=30.), L2Dist(), clip_func=clip, binary_step=2, num_iter=60, **kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        (_, _, s1), (_, s2), (_, s3) = CW.attack_concurrently([(a, (xyz, target, label)), (k, (xyz, target)), (f, (xyz, label))])
    print(json.dumps(dict(victim=cfg['victim'], gain=gain, shake=shake, advpc=int(s1), knn=int(s2), aof=int(s3), B=cfg['B'],
                          classes_predicted=int(label.unique().numel()), clean_margin_median=float((top2[:, 0] - top2[:, 1]).median()),
                          logit_scale=float(logits.abs().max()))), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("victims", nargs="+")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--first", type=int, default=7000)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--points", type=int, default=None)
    ap.add_argument("--gains", type=float, nargs="+", default=[1.0])
    ap.add_argument("--shake-seeds", type=lambda v: None if v == "none" else int(v), nargs="+", default=[None])
    ap.add_argument("--cw", action="store_true", help="pct: the short CW sweep of bench.py instead of HiT-ADV")
    args = ap.parse_args()
    from hit_adv_amd.ShapeAttack.HiT_ADV import HiT_ADV
    from hit_adv_amd.util.adv_utils import UntargetedLogitsAdvLoss
    dev = torch.device("cuda", 0)
    shakes = [(g, None if s is None else dict(seed=s, mean_std=0.05, var_spread=0.2)) for g in args.gains for s in args.shake_seeds]
    for name in args.victims:
        cfg = dict(next(c for c in bench.CONFIGS.values() if c['victim'] == name))
        if args.batch:
            cfg['B'] = args.batch
        if args.points:
            cfg['N'] = args.points
        for gain, shake in shakes:
            tuning = bench.VICTIM_TUNING.pop(name, None)  # (this tool applies its own)
            bench.VICTIM_TUNING[name] = dict(gain=1.0, shake=None)
            model = bench.build_victim(cfg)
            if tuning is not None:
                bench.VICTIM_TUNING[name] = tuning
            if gain != 1.0:
                sharpen(model, gain)
            if shake is not None:
                shake_bn(model, **shake)
            model = model.to(dev)
            data, _ = synth_batch(cfg['B'], cfg['N'], first=args.first)
            data = data.to(dev)
            with torch.no_grad():
                logits = bench.logits_of(model, data[:, :, :3].transpose(1, 2).contiguous())
            label = logits.argmax(1)
            top2 = logits.topk(2, dim=1).values
            if args.cw:
                cw_sweep(model, data, label, cfg, dev, gain, shake, logits, top2)
                continue
            att = HiT_ADV(model, adv_func=UntargetedLogitsAdvLoss(kappa=30.), binary_step=args.steps, num_iter=args.iters,
                          verbose=False, **bench.HP)
            torch.manual_seed(21)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                best, succ = att.attack(data, label)
            print(json.dumps(dict(victim=name, gain=gain, shake=shake, success=int(succ), B=cfg['B'], classes_predicted=int(label.unique().numel()),
                                  clean_margin_median=float((top2[:, 0] - top2[:, 1]).median()), logit_scale=float(logits.abs().max()),
                                  lower=[round(float(v), 2) for v in att.last_lower_bound])), flush=True)


if __name__ == "__main__":
    main()
