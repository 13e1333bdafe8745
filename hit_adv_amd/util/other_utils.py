"""Evaluation harness: ``eval_ASR(model, test_loader, args, val_attack)`` with the reference's call
signature and metric definitions (util/other_utils.py:15-101), plus the one thing the reference lacks:
data-parallel execution.  With ``torch.distributed`` initialised, rank r attacks batches r, r+R, ...
(each ``attack()`` call keeps the reference's per-batch semantics) and the six
running sums are combined by ONE all-reduce (RCCL over xGMI on GPUs, gloo in the CPU tests).
"""
import logging
import os
import time
from datetime import datetime

import torch
import torch.distributed as dist


def create_logger(save_path='', file_type='', level='debug'):
    """Root logger with a stream handler and ./<save_path>/<file_type>_log.txt (other_utils.py:150-170).
    Unlike the reference it creates the directory and does not stack duplicate handlers."""
    level = {'debug': logging.DEBUG, 'info': logging.INFO}.get(level, logging.INFO)
    logger = logging.getLogger()
    logger.setLevel(level)
    for h in list(logger.handlers):
        if getattr(h, '_hitadv', False):
            logger.removeHandler(h)
    cs = logging.StreamHandler()
    cs.setLevel(level)
    cs._hitadv = True
    logger.addHandler(cs)
    if file_type != '':
        os.makedirs(save_path or '.', exist_ok=True)
        fh = logging.FileHandler(os.path.join(save_path, file_type + '_log.txt'), mode='w')
        fh.setLevel(level)
        fh._hitadv = True
        logger.addHandler(fh)
    return logger


def load_checkpoint(model, checkpoint, strict=True):
    """Load a reference-style checkpoint into ``model``: a path or an already loaded object; the weights sit under
    'model_state_dict' (eval.py:79,123), 'state_dict' (save_checkpoint, other_utils.py:173-184) or at top level, with an
    optional DataParallel 'module.' prefix.  Returns the loaded object."""
    obj = torch.load(checkpoint, map_location='cpu') if isinstance(checkpoint, (str, bytes, os.PathLike)) else checkpoint
    sd = obj
    for key in ('model_state_dict', 'state_dict'):
        if isinstance(obj, dict) and key in obj:
            sd = obj[key]
            break
    sd = {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}
    model.load_state_dict(sd, strict=strict)
    return obj


def shard_indices(n_batches, rank, world):
    """Batches owned by ``rank``: r, r+world, ... (round-robin keeps shuffle=False order per rank)."""
    return list(range(rank, n_batches, world))


class RankBatchSampler(torch.utils.data.Sampler):
    """batch_sampler that yields only the batches ``shard_indices`` gives this rank (in-order batches of ``batch_size``
    consecutive samples, as ``shuffle=False`` forms them): a rank's DataLoader workers then read and parse only that
    rank's files, instead of every rank loading the whole split and dropping the other ranks' batches."""

    def __init__(self, n_samples, batch_size, rank, world):
        self.n, self.bs, self.rank, self.world = n_samples, batch_size, rank, world

    def __iter__(self):
        for b in shard_indices(-(-self.n // self.bs), self.rank, self.world):
            yield list(range(b * self.bs, min(self.n, (b + 1) * self.bs)))

    def __len__(self):
        return len(shard_indices(-(-self.n // self.bs), self.rank, self.world))


def rank_loader(dataset, batch_size, num_workers=0, rank=None, world=None):
    """DataLoader over this rank's batches only; ``eval_ASR`` recognises it (``rank_sharded``) and does not filter again."""
    if rank is None or world is None:
        on = dist.is_available() and dist.is_initialized()
        rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
    loader = torch.utils.data.DataLoader(dataset, batch_sampler=RankBatchSampler(len(dataset), batch_size, rank, world),
                                         num_workers=num_workers)
    loader.rank_sharded = True
    return loader


def all_reduce_sums(values, device):
    """SUM-all-reduce a short list of python floats; identity when not distributed."""
    t = torch.tensor(values, dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()


def _default_metrics():
    from . import dist_utils
    from ..FGM.GeoA3_args import uniform_loss
    knn = dist_utils.KNNDist(k=4)
    curv = dist_utils.CurvStdDist(k=4)
    return dict(knn=lambda adv: knn.forward(pc=adv, weights=None, batch_avg=True),
                uniform=lambda adv, k: uniform_loss(adv_pc=adv, k=k),
                curv_std=lambda ori, adv, normal: curv.forward(ori_data=ori, adv_data=adv, ori_normal=normal))


def _logits(model, x):
    out = model(x)
    return out[0] if isinstance(out, tuple) else out  # every victim, not only 'pointnet' (other_utils.py:77-82)


def eval_ASR(model, test_loader, args, val_attack, device=None, metrics=None, logger=None, in_flight=1):
    """Evaluate Attack Success Rate: ASR = (clean-correct - clean-correct-and-still-correct) /
    clean-correct, and the batch means of KNN / Uniform / CurvStd distances of the adversarial clouds.
    Returns the (global) ASR as a float; every rank returns the same value.

    ``in_flight`` > 1 hands that many of this rank's batches at a time to ``val_attack.attack_many`` (HiT_ADV: results and
    RNG draws of back-to-back ``attack`` calls; on the PointNet engine the victim passes of four attacks at a time are merged
    into one pass over 128 clouds and the stacks run on separate HIP streams -- 12 in flight give 2.1x the throughput of 1
    on one MI355X with the eight hardware queues the package asks the HIP runtime for, ``hit_adv_amd.attacks_in_flight``
    caps the count at 8 when they cannot be had -- and no per-iteration progress lines)."""
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    metrics = metrics or _default_metrics()
    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if distributed else 0
    world = dist.get_world_size() if distributed else 1
    if logger is None:
        logger = create_logger('./log', datetime.now().strftime("%Y%m%d%H%M%S") + '_r%d' % rank, 'info')
    for name in ('ker_weight', 'hide_weight', 'budget', 'max_sigm', 'min_sigm', 'central_num', 'attack_type'):
        if hasattr(args, name):
            logger.info('%s: %s', name, getattr(args, name))

    model.eval()
    at_num = at_denom = knn_sum = uni_sum = curv_sum = 0.0
    n_batches = 0
    def score(ori_data, label, adv_data):
        nonlocal at_num, at_denom, knn_sum, uni_sum, curv_sum
        if isinstance(adv_data, tuple):
            adv_data = adv_data[0]
        if not torch.is_tensor(adv_data):
            adv_data = torch.Tensor(adv_data)
        adv_data = adv_data.float().to(device).transpose(1, 2).contiguous()  # [B,3,N]
        ori = ori_data.transpose(1, 2).contiguous()
        normal = ori[:, 3:, :].contiguous() if ori.shape[1] == 6 else None
        ori = ori[:, :3, :].contiguous()
        with torch.no_grad():
            knn_sum += float(metrics['knn'](adv_data))
            uni_sum += float(metrics['uniform'](adv_data, args.k))
            if normal is not None:
                curv_sum += float(metrics['curv_std'](ori, adv_data, normal))
            ok_ori = _logits(model, ori).argmax(dim=-1) == label
            ok_adv = _logits(model, adv_data).argmax(dim=-1) == label
            at_denom += ok_ori.sum().float().item()
            at_num += ok_ori.sum().float().item() - (ok_ori & ok_adv).sum().float().item()

    seconds = dict(attack=0., metrics=0.)  # both phases end in a D2H copy, so host clocks see the device time

    def flush(pending):
        t0 = time.perf_counter()
        if len(pending) > 1:
            results = val_attack.attack_many(pending)
        else:
            results = [val_attack.attack(*pending[0])]
        t1 = time.perf_counter()
        for (ori_data, label), res in zip(pending, results):
            score(ori_data, label, res[0])
        seconds['attack'] += t1 - t0
        seconds['metrics'] += time.perf_counter() - t1

    from .. import groups_in_flight
    group = max(1, int(in_flight)) if hasattr(val_attack, 'attack_many') else 1
    if hasattr(val_attack, 'in_flight'):  # un-stacked victims: capped (a stream, a workspace and activations per attack)
        group = val_attack.in_flight(group)
    stacked = val_attack.stacks() if hasattr(val_attack, 'stacks') else False
    pending = []
    presharded = getattr(test_loader, 'rank_sharded', False)  # rank_loader(): only this rank's batches arrive
    for i, (ori_data, label) in enumerate(test_loader):
        if not presharded and i % world != rank:
            continue
        n_batches += 1
        pending.append((ori_data.float().to(device), label.long().to(device)))
        if len(pending) == group:
            flush(pending)
            pending = []
    for n in groups_in_flight(len(pending), group, stacked=stacked):  # the tail, as bench.py's runner
        flush(pending[:n])
        pending = pending[n:]

    at_num, at_denom, knn_sum, uni_sum, curv_sum, total_batches = all_reduce_sums(
        [at_num, at_denom, knn_sum, uni_sum, curv_sum, float(n_batches)], device)
    ASR = at_num / (at_denom + 1e-9)
    total_batches = max(total_batches, 1.0)
    eval_ASR.last = dict(ASR=ASR, knn=knn_sum / total_batches, uniform=uni_sum / total_batches,
                         curv_std=curv_sum / total_batches, at_num=at_num, at_denom=at_denom,
                         batches=total_batches, world=world)
    eval_ASR.last_seconds = seconds  # this rank's own clocks, not part of the (rank-independent) result
    if rank == 0:
        logger.info('Overall attack success rate: %s', ASR)
        logger.info('Overall KNN dist: %s', knn_sum / total_batches)
        logger.info('Overall Uniform dist: %s', uni_sum / total_batches)
        logger.info('Overall CurvStd dist: %s', curv_sum / total_batches)
    return ASR
