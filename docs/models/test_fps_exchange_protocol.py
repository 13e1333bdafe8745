"""fps_lean's exchange of the waves' keys (hit_adv_amd/csrc/sampling.hip): per step every wave posts its key into LDS word
j % 3 by an atomic maximum, a barrier, every wave reads the word, and wave 0 clears word (j + 2) % 3 -- no second barrier.  The
claim is that three words in rotation make this safe for EVERY interleaving the barrier allows.  The model below runs the waves as
coroutines under a random scheduler (a wave blocked at the barrier waits until all have arrived) and checks that each wave reads, at
each step, exactly the maximum of that step's keys; with only two words in rotation the same scheduler finds the race."""
import random


def run(nw, steps, words, seed):
    rng = random.Random(seed)
    slot = [0] * words
    arrived = [0] * (steps + 1)
    keys = [[rng.randrange(1, 1 << 20) for _ in range(nw)] for _ in range(steps)]
    seen = [[None] * steps for _ in range(nw)]

    def wave(w):
        for j in range(steps):
            yield  # the step's arithmetic
            slot[j % words] = max(slot[j % words], keys[j][w])   # ds_max_rtn_u64 + s_waitcnt: performed before the barrier
            yield
            arrived[j] += 1
            while arrived[j] < nw:                                # s_barrier
                yield
            seen[w][j] = slot[j % words]                          # the read behind the barrier
            yield
            if w == 0:
                slot[(j + 2) % words] = 0                         # cleared by wave 0, no barrier of its own

    live = {w: wave(w) for w in range(nw)}
    while live:
        w = rng.choice(list(live))
        try:
            next(live[w])
        except StopIteration:
            del live[w]
    return all(seen[w][j] == max(keys[j]) for w in range(nw) for j in range(steps))


def test_three_words_in_rotation_are_enough_for_every_interleaving():
    assert all(run(nw, 40, 3, seed) for nw in (4, 8) for seed in range(300))


def test_two_words_in_rotation_are_not():
    # (j + 2) % 2 == j % 2: wave 0 would clear the word the slower waves are still reading / the faster ones already post into
    assert not all(run(4, 40, 2, seed) for seed in range(300))
