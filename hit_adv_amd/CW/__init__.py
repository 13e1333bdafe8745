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
    Attackers without ``steps()`` (CWPerturb, the Add family) and a pass that left fp16's range fall back to the plain
    sequence.  No progress lines are printed in a meaningful order: construct the attackers with ``verbose=False``."""
    import os
    import threading

    import torch
    from ..model._pointwise import Fp16RangeExceeded
    THREADS = os.environ.get("HITADV_CW_THREADS", "1") != "0"  # 0: one host thread queues all loops (A/B switch)
    calls = [(a, tuple(args)) for a, args in calls]
    if len(calls) < 2 or not all(hasattr(a, 'steps') for a, _ in calls):
        return [a.attack(*args) for a, args in calls]
    rng = torch.get_rng_state()
    here = torch.cuda.current_stream()
    streams = [torch.cuda.Stream() for _ in calls]
    gens = [a.steps(*args) for a, args in calls]
    results = [None] * len(calls)
    def advance(i, stop):
        with torch.cuda.stream(streams[i]):
            try:
                while True:
                    if next(gens[i]) == stop:
                        return
            except StopIteration as done:
                results[i] = done.value

    try:
        for i, st in enumerate(streams):  # setups first, one after the other: draws in sequence order, captures undisturbed
            st.wait_stream(here)
            advance(i, 'ready')
        # the loops: one host thread per attack.  Launching a captured PCT iteration costs the host about what it costs the
        # GPU to run it (a thousand-odd graph nodes), so ONE thread feeding three streams feeds none of them fast enough
        # (measured: 18.97 s for cfg5's sweep against 19.61 s in sequence); graph launches release the interpreter lock.
        failures = []

        def loop(i):
            try:
                advance(i, 'enqueued')
            except BaseException as e:  # noqa: BLE001  (re-raised in the caller's thread)
                failures.append(e)
        if THREADS:
            workers = [threading.Thread(target=loop, args=(i,), name='hitadv-cw-%d' % i) for i in range(len(gens))]
            for w in workers:
                w.start()
            for w in workers:
                w.join()
        else:
            for i in range(len(gens)):
                loop(i)
        if failures:
            raise failures[0]
        for i in range(len(gens)):  # results are read back in sequence order
            advance(i, None)
        for st in streams:
            here.wait_stream(st)
        return results
    except Fp16RangeExceeded:
        for g in gens:
            g.close()
        torch.cuda.synchronize()
        torch.set_rng_state(rng)
        return [a.attack(*args) for a, args in calls]  # each degrades on its own (model/_pointwise.py)


def __getattr__(name):
    if name in _EXPORTS:
        return getattr(importlib.import_module('.' + _EXPORTS[name], __name__), name)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
