"""bench.py's own N > 1 bookkeeping on CPU (gloo, world_size 2): MAX of the elapsed time, SUM of the success counters,
whole-job throughput = clouds of all ranks / slowest rank's time, and the count of collective calls a rank makes
(what RCCL will see on the GPU box: 1 MAX + 1 SUM all-reduce, plus the timing barriers)."""
import json
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import bench
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    collectives = dict(barrier=0, all_reduce_max=0, all_reduce_sum=0)
    elapsed, succ, att = bench.reduce_over_ranks(10.0 + 2.0 * rank, 5 + rank, 64, 'cpu', world, collectives)
    cfg = bench.CONFIGS['cfg2']
    line = bench.headline(cfg, 2, 1, world, elapsed, succ, att, 5000, 2, dict(hip_graph=True), collectives)
    out[rank] = json.dumps(line)
    dist.destroy_process_group()


def test_two_rank_reduction_and_line():
    ctx = mp.get_context('spawn')
    out = ctx.Manager().dict()
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    a, b = json.loads(out[0]), json.loads(out[1])
    assert a == b  # every rank reduces to the same line
    assert a['n_gpus'] == 2 and a['ms_per_step'] == 12.0 / 2 * 1e3          # slowest rank: 12 s for 2 steps
    assert a['value'] == 2 * 32 * 2 / 12.0 and a['scaling'] == 'weak'        # clouds of BOTH ranks / that time
    assert a['attack_success'] == {'succeeded': 11.0, 'attacked': 128.0}
    assert a['collectives_per_rank'] == {'barrier': 0, 'all_reduce_max': 1, 'all_reduce_sum': 1, 'world': 2,
                                         'backend': 'nccl (RCCL)'}
    assert abs(a['cloud_iterations_per_s'] - a['value'] * 5000) < 1e-6


def test_single_process_line_has_no_collectives():
    sys.path.insert(0, ROOT)
    import bench
    c = dict(barrier=0, all_reduce_max=0, all_reduce_sum=0)
    e, s, n = bench.reduce_over_ranks(3.0, 7, 96, 'cpu', 1, c)
    assert (e, s, n) == (3.0, 7.0, 96.0) and c == dict(barrier=0, all_reduce_max=0, all_reduce_sum=0)
    line = bench.headline(bench.CONFIGS['cfg3'], 3, 0, 1, e, s, n, 5000, 1, {}, c)
    assert line['value'] == 3 * 32 / 3.0 and line['vs_baseline'] is None and line['collectives_per_rank']['backend'] is None
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config'):
        assert key in line
    assert 'workload' in line['config'] and 'model' not in line['config']


def test_bench_flop_model_of_the_pointnet_forward():
    sys.path.insert(0, ROOT)
    import bench
    from hit_adv_amd.model.pointnet import PointNetFeatureModel
    # count multiply-adds from the module's own layer shapes: every Conv1d(k=1) acts per point, every Linear per cloud;
    # plus the two learned transforms applied per point (3x3 and 64x64)
    m = PointNetFeatureModel(40, normal_channel=False)
    per_point = sum(2 * c.in_channels * c.out_channels for c in m.modules() if isinstance(c, torch.nn.Conv1d))
    per_cloud = sum(2 * l.in_features * l.out_features for l in m.modules() if isinstance(l, torch.nn.Linear))
    per_point += 2 * (3 * 3 + 64 * 64)
    assert bench.pointnet_forward_flops(32, 1024) == 32.0 * (1024 * per_point + per_cloud)


def test_bench_help_runs_without_a_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--help'], capture_output=True, text=True)
    assert out.returncode == 0 and '--config' in out.stdout


def _bench(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *argv], capture_output=True, text=True, env=e,
                          timeout=300)


def test_gpus_2_without_a_launcher_starts_two_ranks():
    """`python bench.py --gpus 2` (no torchrun around it): bench.py starts the two ranks itself before any GPU call and
    relays rank 0's line -- exactly one line on stdout, n_gpus = 2, the collectives of a world of 2 (gloo stands in for
    RCCL; --mock-cpu runs no attack and says so in `data`)."""
    out = _bench('--gpus', '2', '--steps', '2', '--warmup', '1', '--mock-cpu')
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['steps'] == 2 and line['warmup'] == 1
    assert line['collectives_per_rank'] == {'barrier': 2, 'all_reduce_max': 1, 'all_reduce_sum': 1, 'world': 2, 'backend': 'gloo'}
    assert line['attack_success']['attacked'] == 2 * 2 * 32          # both ranks' clouds
    assert line['value'] == 2 * 2 * 32 / (line['ms_per_step'] * 2 / 1e3)
    assert line['ms_per_step'] >= 2 * 10.                           # the slower rank (rank 1 sleeps twice as long)
    assert line['data'].startswith('MOCK')                          # never to be read as a measurement


def test_gpus_must_equal_the_world_that_runs():
    """A launcher that started a different number of ranks than --gpus says is an error, not a line with the wrong
    n_gpus; and so is a rank that dies (exit code of the whole job is non-zero, no line)."""
    out = _bench('--gpus', '2', '--mock-cpu', env={'WORLD_SIZE': '1'})
    assert out.returncode != 0 and '"n_gpus"' not in out.stdout and 'WORLD_SIZE=1' in out.stderr
    out = _bench('--gpus', '1', '--mock-cpu', env={'WORLD_SIZE': '2', 'RANK': '0'})
    assert out.returncode != 0 and '"n_gpus"' not in out.stdout
    out = _bench('--gpus', '2', '--steps', '1')                     # the real job on a box without two GPUs
    assert out.returncode != 0 and '"n_gpus"' not in out.stdout
