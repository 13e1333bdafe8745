"""hit_adv_amd -- MI355X-native HiT-ADV hot path.

Module names mirror the reference tree so that its callers switch with an import prefix:
    from hit_adv_amd.ShapeAttack.HiT_ADV import HiT_ADV
    from hit_adv_amd.util.dist_utils import ChamferDist, HausdorffDist, KNNDist, ChamferkNNDist
    from hit_adv_amd.util.set_distance import chamfer, hausdorff
    from hit_adv_amd.util.adv_utils import UntargetedLogitsAdvLoss
    from hit_adv_amd.pytorch3d_ops import knn_points, knn_gather
    from hit_adv_amd.pointnet2_ops import pointnet2_utils
All compute runs in libhitadv_hip.so (hand-written HIP for gfx950); there is no CPU path.
"""
import os as _os
import sys as _sys
import warnings as _warnings

# attack_many() runs stacks of independent attacks on their own HIP streams; with the runtime's default of 4 hardware queues
# three or more of them serialise again (bench.py: 30.3 vs 37.4 clouds/s at twelve attacks in three stacks).  The variable is read ONCE, when the HIP runtime starts: setting
# it here only helps if nothing has touched the GPU yet (importing torch is fine; torch.cuda.is_available(), set_device(),
# init_process_group('nccl') and a profiler's preloaded library are not).  This is the ONE thing the package changes in the
# importing process's environment; HITADV_KEEP_ENVIRONMENT=1 switches it off (attacks in flight are then capped for 4 queues).
_HW_QUEUES_WANTED = 8


def _runtime_started():
    t = _sys.modules.get("torch")
    return bool(t is not None and t.cuda.is_initialized())


_preset = _os.environ.get("GPU_MAX_HW_QUEUES")
if _os.environ.get("HITADV_KEEP_ENVIRONMENT") == "1":
    # the importing process does not want its environment touched: whatever the runtime's queue count is, it stays
    _QUEUES_IN_TIME = _preset is not None
elif _preset is None:
    _os.environ["GPU_MAX_HW_QUEUES"] = str(_HW_QUEUES_WANTED)
    _QUEUES_IN_TIME = not _runtime_started()
    if not _QUEUES_IN_TIME:
        _warnings.warn("hit_adv_amd: the HIP runtime was already initialised when the package was imported, so "
                       "GPU_MAX_HW_QUEUES=8 cannot take effect in this process (4 hardware queues): attacks in flight are "
                       "capped at 8.  Export GPU_MAX_HW_QUEUES=8, or import hit_adv_amd, before the first torch.cuda call.",
                       RuntimeWarning, stacklevel=2)
else:
    _QUEUES_IN_TIME = True  # the environment carried it into the process: whatever started the runtime saw it


def hardware_queues():
    """HIP hardware queues this process's streams are multiplexed onto, as far as the package can know: the value of
    GPU_MAX_HW_QUEUES if it was in the environment before the runtime started, else the runtime's default of 4."""
    if not _QUEUES_IN_TIME:
        return 4
    try:
        return int(_os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        return 4


def attacks_in_flight(requested):
    """How many independent attacks to hand to ``attack_many`` at a time: ``requested``, capped at 16 (two stacks of eight on
    two streams, ``stack_sizes``) when the process has only the runtime's 4 hardware queues -- three streams on four shared
    queues serialise again (round 3, 12 in flight: 30.3 clouds/s on 4 queues, 37.4 on 8; 8 in flight on 4 queues: 35.5)."""
    requested = max(1, int(requested))
    return requested if hardware_queues() >= 8 else min(requested, 16)


def groups_in_flight(pending, in_flight, stacked=True):
    """Split ``pending`` batches into the group sizes attacked together: at most ``in_flight`` at a time.  With stacked victim
    passes the groups are BALANCED (20 batches at 12 in flight go as 10 + 10, not 12 + 8: 47.9 against 47.0 clouds/s -- a
    short last group leaves streams idle).  Without stacking (one stream per attack) full groups first, and a remainder of
    three goes as two and one: three streams measured slower than two.  bench.py and eval_ASR share this."""
    in_flight = max(1, int(in_flight))
    pending = max(0, int(pending))
    if stacked and pending > 0:
        groups = -(-pending // in_flight)
        return [pending // groups + (1 if i < pending % groups else 0) for i in range(groups)]
    sizes = []
    while pending > 0:
        n = min(in_flight, pending)
        if n == 3 and in_flight != 3 and not stacked:
            n = 2
        sizes.append(n)
        pending -= n
    return sizes


def stack_sizes(attacks, per_stack):
    """How ``attacks`` stacked attacks of one group are cut into stacks (one stream each): as few stacks as ``per_stack``
    allows, but three from six attacks on (eight as 3 + 3 + 2 on three streams: 46.9 clouds/s, as 4 + 4 on two: 45.9) where the
    runtime has the hardware queues for three streams, and balanced (ten as 4 + 3 + 3, twenty as 7 + 7 + 6)."""
    attacks, per_stack = int(attacks), max(1, int(per_stack))
    stacks = -(-attacks // per_stack)
    if attacks >= 6 and hardware_queues() >= 8:
        stacks = max(stacks, 3)
    return [attacks // stacks + (1 if i < attacks % stacks else 0) for i in range(stacks)]


__version__ = "0.1.0"
