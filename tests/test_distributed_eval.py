"""The N>1 path on CPU: eval_ASR shards batches over ranks and combines its six counters with one
all-reduce.  world_size=2 over gloo must reproduce the single-process result on the same batches.
The attacker here is a deterministic stand-in and the metrics come from the CPU oracle, so this runs
without a GPU; the GPU path is the same code with RCCL as the backend.
"""
import argparse
import logging
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ToyVictim, synth_batch


class _ShiftAttack:
    """Moves every point by a fixed, label-dependent offset: cheap, deterministic, sometimes flips the label."""

    def attack(self, data, target):
        xyz = data[:, :, :3]
        shift = 0.15 * torch.sin(target.float())[:, None, None] * torch.ones_like(xyz)
        return (xyz + shift).double().numpy(), torch.tensor(0)


def _metrics():
    from oracle import c_oracle as N
    from oracle import hitadv_oracle as O
    return dict(knn=lambda adv: O.knn_dist(adv, None, True, 4),
                uniform=lambda adv, k: O.uniform_loss(adv, N, k=k),
                curv_std=lambda ori, adv, normal: O.curv_std_dist(ori, adv, normal, k=4))


def _loader(n_batches=5, bs=3, n=256):
    torch.manual_seed(0)
    model = ToyVictim()
    with torch.no_grad():
        model.conv.weight.mul_(3.0)
        model.fc.weight.mul_(4.0)
    batches = []
    for i in range(n_batches):
        data, _ = synth_batch(bs, n, first=900 + i * bs)
        with torch.no_grad():
            label = model(data[:, :, :3].transpose(1, 2).contiguous()).argmax(1)
        label[0] = (label[0] + 1) % 40  # one clean-misclassified sample per batch exercises at_denom
        batches.append((data, label))
    return model.eval(), batches


def _run(rank, world, port, out):
    from hit_adv_amd.util.other_utils import eval_ASR
    if world > 1:
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    model, batches = _loader()
    args = argparse.Namespace(k=5, model='toy', budget=0.55)
    log = logging.getLogger('quiet')
    log.addHandler(logging.NullHandler())
    log.propagate = False
    asr = eval_ASR(model, batches, args, _ShiftAttack(), device='cpu', metrics=_metrics(), logger=log)
    res = dict(eval_ASR.last, asr=asr)
    if world > 1:
        dist.destroy_process_group()
        out[rank] = res
    return res


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_equals_single_process():
    single = _run(0, 1, 0, None)
    assert single['at_denom'] == 5 * 2 and 0.0 <= single['asr'] <= 1.0 and single['batches'] == 5
    ctx = mp.get_context('spawn')
    out = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert out[0] == out[1]  # every rank holds the same global result
    for k in ('asr', 'at_num', 'at_denom', 'batches'):
        assert out[0][k] == single[k]
    for k in ('knn', 'uniform', 'curv_std'):
        np.testing.assert_allclose(out[0][k], single[k], rtol=1e-12)  # same per-batch values, summed in another order
    assert out[0]['world'] == 2


def test_shard_indices_partition():
    from hit_adv_amd.util.other_utils import shard_indices
    for n in (0, 1, 7, 8, 9):
        for world in (1, 2, 3, 8):
            parts = [shard_indices(n, r, world) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))


def test_rank_loader_reads_only_its_own_batches():
    """rank_loader (batch_sampler level sharding): the union over ranks is the shuffle=False batch sequence, a rank never
    touches another rank's samples, and eval_ASR does not filter a pre-sharded loader a second time."""
    from hit_adv_amd.util.other_utils import RankBatchSampler, rank_loader, shard_indices

    class Counting(torch.utils.data.Dataset):
        def __init__(self):
            self.seen = []

        def __len__(self):
            return 11

        def __getitem__(self, i):
            self.seen.append(i)
            return torch.full((4, 6), float(i)), torch.tensor(i % 40)

    plain = [b[1].tolist() for b in torch.utils.data.DataLoader(Counting(), batch_size=3, shuffle=False)]
    for world in (1, 2, 3):
        got = {}
        for rank in range(world):
            ds = Counting()
            loader = rank_loader(ds, 3, rank=rank, world=world)
            assert loader.rank_sharded and len(loader) == len(shard_indices(4, rank, world))
            mine = [b[1].tolist() for b in loader]
            assert sorted(ds.seen) == sorted(sum(mine, []))  # nothing outside this rank's batches was read
            for j, b in zip(shard_indices(4, rank, world), mine):
                got[j] = b
        assert [got[j] for j in range(4)] == plain
    assert list(RankBatchSampler(0, 3, 0, 2)) == []
