"""The evaluation driver (eval.py at the repo root; the reference's eval.py:22-135) and the synthetic split it can run on."""
import json
import os
import subprocess
import sys

import pytest
import torch

from helpers import synth_cloud

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_synthetic_split_is_the_bench_generator():
    from hit_adv_amd.Dataset.synthetic import SyntheticClouds
    ds = SyntheticClouds(5, 256, first=3)
    assert len(ds) == 5
    for i in (0, 4):
        pts, label = ds[i]
        ref_pts, ref_label = synth_cloud(3 + i, 256)
        assert torch.equal(pts, ref_pts) and int(label) == int(ref_label)
    with pytest.raises(IndexError):
        ds[5]
    with pytest.raises(ValueError):
        SyntheticClouds(1, kind='cube')


def test_sphere_split_is_normalised_and_its_normals_point_outwards():
    from hit_adv_amd.Dataset.synthetic import SyntheticClouds
    pts, label = SyntheticClouds(2, 512, kind='sphere')[1]
    xyz, normal = pts[:, :3], pts[:, 3:]
    assert pts.shape == (512, 6) and 0 <= int(label) < 40
    assert abs(float(xyz.norm(dim=1).max()) - 1.) < 1e-6 and float(xyz.mean(0).abs().max()) < 1e-6
    assert torch.allclose(normal.norm(dim=1), torch.ones(512), atol=1e-6)
    assert float((torch.nn.functional.normalize(xyz, dim=1) * normal).sum(1).min()) > 0.9


def test_flags_keep_the_reference_names_and_defaults():
    import eval as driver
    a = driver.parse_args([])
    ref = dict(num_class=40, budget=0.55, num_iter=100, num_point=1024, model='pointnet', emb_dims=1024, dropout=0.2, k=5,
               curv_loss_knn=16, cd_weight=0.0001, ker_weight=1., hide_weight=1., max_sigm=1.2, min_sigm=0.1,
               central_num=192, total_central_num=256, dataset='ModelNet', kappa=30., attack_lr=1e-2, binary_step=10)
    for name, value in ref.items():  # eval.py:24-66 and FGM/CWPert_args.py:39-44 of the reference
        assert getattr(a, name) == value, name
    a = driver.parse_args(['--model', 'pointnet++', '--synthetic', '2', '--in_flight', '1'])
    assert (a.model, a.synthetic, a.in_flight) == ('pointnet++', 2, 1)
    with pytest.raises(SystemExit):
        driver.build_loader(driver.parse_args([]))  # neither a dataset root nor --synthetic


SMALL = ['--synthetic', '3', '--batch_size', '4', '--num_point', '1024', '--num_iter', '3', '--binary_step', '2',
         '--central_num', '16', '--total_central_num', '32', '--checkpoint', '/nonexistent']


def _last_json(text):
    return json.loads([l for l in text.strip().splitlines() if l.startswith('{')][-1])


@pytest.mark.gpu
def test_driver_runs_on_synthetic_clouds(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'eval.py'), '--log_dir', str(tmp_path)] + SMALL,
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    r = _last_json(out.stdout)
    assert r['batches'] == 3 and r['world'] == 1 and 0. <= r['ASR'] <= 1. and r['weights'].startswith('random init')
    assert r['clean_correct'] == 12  # self-labelled: every synthetic cloud counts
    assert all(r[k] == r[k] for k in ('knn', 'uniform', 'curv_std'))  # no NaN
    log = open(os.path.join(str(tmp_path), 'eval_last_log.txt')).read()
    for line in ('Overall attack success rate', 'Overall KNN dist', 'Overall Uniform dist', 'Overall CurvStd dist'):
        assert line in log


@pytest.mark.gpu
@pytest.mark.parametrize("model,extra", [("dgcnn", ["--k", "5"]), ("dgcnn", ["--k", "20", "--metric_k", "5"]),
                                         ("pointnet++", []), ("pct", ["--synthetic_kind", "sphere"])])
def test_driver_runs_every_victim(tmp_path, model, extra):
    """The four victims of eval.py:109-121 through the driver (short attack, synthetic clouds, random-init weights)."""
    args = ['--log_dir', str(tmp_path), '--model', model, '--synthetic', '2', '--batch_size', '4', '--num_iter', '3',
            '--binary_step', '2', '--central_num', '16', '--total_central_num', '32', '--checkpoint', '/nonexistent'] + extra
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'eval.py')] + args, capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    r = _last_json(out.stdout)
    assert r['model'] == model and r['batches'] == 2 and r['clean_correct'] == 8 and 0. <= r['ASR'] <= 1.
    assert all(r[k] == r[k] for k in ('knn', 'uniform', 'curv_std'))
    bad = subprocess.run([sys.executable, os.path.join(ROOT, 'eval.py'), '--model', model, '--synthetic', '1', '--k', '20',
                          '--checkpoint', '/nonexistent', '--log_dir', str(tmp_path)], capture_output=True, text=True,
                         timeout=600, cwd=ROOT) if model == 'dgcnn' and '--metric_k' in extra else None
    if bad is not None:  # k + 1 neighbours do not fit the Uniform metric's smallest ball: a clear message, no traceback
        assert bad.returncode != 0 and 'metric_k' in (bad.stderr + bad.stdout)


@pytest.mark.gpu
def test_driver_under_torchrun_matches_the_plain_run(tmp_path):
    """One rank through the launcher the 8-GPU run uses (RCCL process group, rank-sharded batches, one all-reduce)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    plain = subprocess.run([sys.executable, os.path.join(ROOT, 'eval.py'), '--log_dir', str(tmp_path), '--in_flight', '1']
                           + SMALL, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    launched = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
                               '--master-addr', '127.0.0.1', '--master-port', '29731', os.path.join(ROOT, 'eval.py'),
                               '--log_dir', str(tmp_path), '--in_flight', '1'] + SMALL,
                              capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert plain.returncode == 0, plain.stderr[-2000:]
    assert launched.returncode == 0, launched.stderr[-2000:]
    a, b = _last_json(plain.stdout), _last_json(launched.stdout)
    for k in ('ASR', 'knn', 'uniform', 'curv_std', 'clean_correct', 'batches'):
        assert a[k] == b[k], k


def test_groups_and_stacks_are_balanced():
    """How pending batches are cut into groups of attacks in flight and a group into stacks (hit_adv_amd/__init__.py): balanced
    where the victim passes stack (a short last group leaves streams idle), round 2's rule where every attack has a stream."""
    import hit_adv_amd as H
    assert H.hardware_queues() >= 8  # (importing the package puts GPU_MAX_HW_QUEUES=8 in place on a process that has not started the runtime)
    assert H.groups_in_flight(20, 12) == [10, 10] and H.groups_in_flight(24, 12) == [12, 12] and H.groups_in_flight(13, 12) == [7, 6]
    assert H.groups_in_flight(5, 12) == [5] and H.groups_in_flight(0, 12) == []
    assert H.groups_in_flight(7, 4, stacked=False) == [4, 2, 1] and H.groups_in_flight(8, 4, stacked=False) == [4, 4]
    for n in range(1, 30):
        for inf in (1, 4, 12):
            for st in (True, False):
                g = H.groups_in_flight(n, inf, stacked=st)
                assert sum(g) == n and max(g) <= inf and min(g) >= 1
        for per in (1, 3, 4, 8):
            s = H.stack_sizes(n, per)
            assert sum(s) == n and max(s) - min(s) <= 1 and max(s) <= max(per, -(-n // 3))
    assert H.stack_sizes(12, 4) == [4, 4, 4] and H.stack_sizes(10, 4) == [4, 3, 3] and H.stack_sizes(8, 4) == [3, 3, 2]
    assert H.stack_sizes(4, 4) == [4] and H.stack_sizes(3, 4) == [3] and H.stack_sizes(3, 2) == [2, 1]
