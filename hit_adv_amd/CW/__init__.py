"""CW attack family.  Exports the class names of the reference's CW package (CW/__init__.py:1-14); the sub-modules are
imported on first use, so ``from hit_adv_amd.CW import CWKNN`` does not pull in scikit-learn for the cluster attacks."""
import importlib

_EXPORTS = {
    'CWPerturb': 'Perturb', 'CWPerturbT': 'PerturbT', 'CWAdd': 'Add', 'CWKNN': 'kNN', 'CWUKNN': 'UKNN', 'CWAOF': 'AOF',
    'CWTAOF': 'TAOF', 'CWUAdvPC': 'UAdvPC', 'CWAdvPC': 'AdvPC', 'CWUAEAOF': 'UAEAOF', 'CWAddClusters': 'Add_Cluster',
    'CWAddObjects': 'Add_Objects',
}
__all__ = sorted(_EXPORTS) + ['attack_concurrently']


def attack_concurrently(calls):
    """Several independent CW attacks on one GPU at once: ``[(attacker, args), ...] -> [attacker.attack(*args), ...]``.

    The results are those of calling the attacks one after the other, in the order given: every attack takes ALL its random
    numbers when its turn comes in that order (``steps()`` draws them up front, in the reference's order within the attack)
    and captures its iteration; only then do the loops start, each on a stream of its own -- the host queues one attack's
    replays and goes on to the next while the GPU works through all of them.  A PCT / DGCNN / PointNet++ pass at batch 32
    is hundreds of kernels of a few microseconds that leave most of the chip idle: three attacks in flight fill it.
    Attackers without ``steps()`` (CWPerturb, the Add family) fall back to the plain sequence.  No progress lines are printed
    in a meaningful order: construct the attackers with ``verbose=False``.

    fp16 range: the flag a fused fp16x2 layer raises is ONE PER DEVICE, so an overflow in any of the attacks invalidates all
    of them: all are thrown away and run again one after the other, each degrading on its own (model/_pointwise.py).  On
    that path and on any other exception (a failed capture under ``use_graph=True``, out of memory, KeyboardInterrupt)
    the GPU is drained FIRST and only then are the generators closed -- closing one drops its captured graph and the graph's
    private memory pool, and doing that under a replay still in flight on another stream is a memory access fault (found in
    round 5 by a sharpened PCT victim whose first attack overflowed while the other two were still running) -- the victims'
    feeds are closed, the side streams joined to the caller's, and the exception (other than the range one) re-raised."""
    import os
    import time

    import torch
    from ..model._pointwise import Fp16RangeExceeded
    calls = [(a, tuple(args)) for a, args in calls]
    if len(calls) < 2 or not all(hasattr(a, 'steps') for a, _ in calls):
        return [a.attack(*args) for a, args in calls]
    rng = torch.get_rng_state()
    here = torch.cuda.current_stream()
    streams = [torch.cuda.Stream() for _ in calls]
    gens = [a.steps(*args) for a, args in calls]
    results = [None] * len(calls)
    def advance(i, stops):
        """Generator i to its next stop in ``stops`` (None: to its end); True once it has returned its result."""
        with torch.cuda.stream(streams[i]):
            try:
                while True:
                    if next(gens[i]) in stops:
                        return False
            except StopIteration as done:
                results[i] = done.value
                return True

    finished = False
    try:
        for i, st in enumerate(streams):  # setups first, one after the other: draws in sequence order ...
            st.wait_stream(here)
            advance(i, ('probed',))
        torch.cuda.synchronize()
        for i in range(len(gens)):        # ... then the captures, with no eager victim work after any of them
            advance(i, ('ready',))
        # the loops: the host goes round the attacks, a few iterations of each per turn (CW/_family.py::TURN), the longer
        # loops taking proportionally more turns per round so that all of them end together; launching a captured PCT
        # iteration costs the host 0.85 ms against ~4 ms of GPU time (tools/graph_launch_cost.py), so one thread keeps
        # every stream's queue full and waits only where the runtime's own limit on queued work makes it wait
        totals = [max(1, int(getattr(a, 'total_iterations', 1))) for a, _ in calls]
        share = [max(1, round(t / min(totals))) for t in totals]
        queued = [False] * len(gens)
        trace = os.environ.get("HITADV_CW_TIMELINE") == "1"  # diagnostic: when the host got each attack fully queued
        t0 = time.perf_counter()
        while not all(queued):
            for i in range(len(gens)):
                with torch.cuda.stream(streams[i]):
                    for _ in range(share[i]):
                        if queued[i]:
                            break
                        queued[i] = next(gens[i]) == 'enqueued'
                        if queued[i] and trace:
                            print("hitadv cw timeline: attack %d (%s) fully queued at %.3f s" % (
                                i, type(calls[i][0]).__name__, time.perf_counter() - t0))
        if trace:
            torch.cuda.synchronize()
            print("hitadv cw timeline: GPU drained at %.3f s" % (time.perf_counter() - t0))
        for i in range(len(gens)):  # results are read back in sequence order
            advance(i, ())
        finished = True
        return results
    except Fp16RangeExceeded:
        pass  # cleaned up below, then the plain sequence
    finally:
        if not finished:
            torch.cuda.synchronize()  # FIRST: no replay of any attack may still be in flight when its graph is dropped
            for g in gens:
                g.close()
            for a, _ in calls:
                victim = getattr(a, '_victim', None)
                if victim is not None:
                    victim.feed = None
            from ..model import _pointwise
            for flag in _pointwise._RANGE_FLAGS.values():  # what the attacks that were still running raised meanwhile
                flag.zero_()
            for a, _ in calls:
                view = getattr(getattr(a, '_victim', None), 'view', None)
                if hasattr(view, 'range_flag'):
                    view.range_flag.zero_()
            torch.cuda.synchronize()
        for st in streams:
            here.wait_stream(st)
    torch.set_rng_state(rng)
    return [a.attack(*args) for a, args in calls]  # each degrades on its own (model/_pointwise.py)


def __getattr__(name):
    if name in _EXPORTS:
        return getattr(importlib.import_module('.' + _EXPORTS[name], __name__), name)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
