"""The synchronisation of gemm_f16x2_ring_k's three-stage LDS ring (hit_adv_amd/csrc/gemm16.hip), as a model: every wave requests
(by LDS-DMA, completing asynchronously and in order) its share of stage s + 3 into the buffer stage s has left, reads step s + 1's
fragments from LDS (asynchronously too) while step s multiplies, and meets the other waves at ONE barrier per step, in front of which it
waits for its own DMA of stage s + 1 (vmcnt) and for its own outstanding LDS reads (lgkmcnt(0)).  A random scheduler decides when each
wave runs and when each asynchronous operation lands.  Checked: a fragment read never sees a stage that has not fully landed, and a DMA
never lands in a buffer a wave is still reading.  Without the lgkmcnt(0) -- the first build waited only for the A rows -- the scheduler
finds the second kind of violation."""
import random


def run(nw, nk, seed, wait_reads_before_barrier=True):
    rng = random.Random(seed)
    landed = [[False] * nw for _ in range(nk)]       # landed[stage][wave]: that wave's share of the stage is in LDS
    reading = [set() for _ in range(3)]              # waves with an LDS read of this BUFFER still in flight
    dma_q = [[] for _ in range(nw)]                  # per wave: requested, not yet landed (in order)
    arrived = [0] * (nk + 1)
    bad = []

    def land_one(w):
        if dma_q[w]:
            st = dma_q[w].pop(0)
            if reading[st % 3]:
                bad.append(('dma lands in a buffer being read', st, w, set(reading[st % 3])))
            landed[st][w] = True

    def wave(w):
        pend = []                                    # this wave's LDS reads in flight: (buffer)
        for st in range(min(3, nk)):
            dma_q[w].append(st)
        yield
        while not landed[0][w]:                      # prologue: vmcnt for stage 0, barrier, read step 0's fragments
            yield
        arrived[nk] += 1
        while arrived[nk] < nw:
            yield
        if not all(landed[0]):
            bad.append(('read before landed', 0, w))
        reading[0].add(w); pend.append(0)
        for s in range(nk):
            # split_a(): waits for the A rows only -- the B fragments (requested later) may still be in flight
            yield
            if s + 1 < nk:
                while not landed[s + 1][w]:          # s_waitcnt vmcnt: this wave's share of stage s + 1
                    yield
                if wait_reads_before_barrier:        # ... lgkmcnt(0): every fragment of stage s is in registers
                    for b in pend:
                        reading[b].discard(w)
                    pend.clear()
                arrived[s] += 1
                while arrived[s] < nw:               # s_barrier
                    yield
                if s + 3 < nk:
                    dma_q[w].append(s + 3)           # into buffer (s + 3) % 3 = s % 3
                if not all(landed[s + 1]):
                    bad.append(('read before landed', s + 1, w))
                reading[(s + 1) % 3].add(w); pend.append((s + 1) % 3)
            yield                                    # the products; outstanding reads complete whenever the scheduler says
            for b in list(pend):
                if rng.random() < 0.5 or not wait_reads_before_barrier and rng.random() < 0.2:
                    reading[b].discard(w); pend.remove(b)
        for b in pend:
            reading[b].discard(w)

    live = {w: wave(w) for w in range(nw)}
    while live:
        w = rng.choice(list(live))
        if rng.random() < 0.4:
            land_one(rng.randrange(nw))
            continue
        try:
            next(live[w])
        except StopIteration:
            del live[w]
        if all(not q for q in dma_q) is False and rng.random() < 0.3:
            land_one(rng.randrange(nw))
    return bad


def test_ring_protocol_holds_under_a_random_scheduler():
    for nk in (2, 3, 4, 16):
        for seed in range(150):
            assert run(8, nk, seed) == []


def test_without_the_lgkmcnt_wait_a_dma_can_land_in_a_buffer_still_being_read():
    assert any(run(8, 16, seed, wait_reads_before_barrier=False) for seed in range(300))
