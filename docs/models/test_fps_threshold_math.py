"""The exact tie threshold of PCT's sampler in hit_adv_amd/csrc/sampling.hip::sqrt_preimage_floor, restated in numpy: for a
correctly rounded fp32 square root s, the smallest float t with sqrt_rn(t) == s is the smallest float >= ((s + pred(s)) / 2)^2,
evaluated in fp64 (exactly: a 25-bit number squared has 50 bits).  fps_lean keeps the running distance squared and takes ONE sqrt per
wave and step; points whose squared distance is >= t are the ones whose sqrt rounds to the maximum (util/other_utils.py:237-272
compares the rounded square roots).  numpy's float32 sqrt is correctly rounded, so it can check the claim value by value."""
import numpy as np


def preimage_floor(s):
    s = np.asarray(s, dtype=np.float32)
    sp = (s.view(np.uint32) - np.uint32(1)).view(np.float32)
    mid = (s.astype(np.float64) + sp.astype(np.float64)) * 0.5
    m2 = mid * mid
    t = m2.astype(np.float32)  # round to nearest
    up = (t.view(np.uint32) + np.uint32(1)).view(np.float32)
    return np.where(t.astype(np.float64) < m2, up, t).astype(np.float32)


def test_threshold_is_the_smallest_float_whose_sqrt_rounds_to_s():
    rng = np.random.default_rng(5)
    # squared distances as the kernel sees them: the clamp value, tiny and ordinary magnitudes, the initial 1e10, whole binades
    x = np.concatenate([np.float32([1e-7, 1e10, 1.0, 2.0, 4.0, 3.9999998, 0.25]),
                        rng.uniform(1e-12, 4.0, 200000).astype(np.float32),
                        np.exp(rng.uniform(np.log(1e-30), np.log(1e30), 200000)).astype(np.float32),
                        np.arange(0x3F800000, 0x3F800000 + 70000, dtype=np.uint32).view(np.float32),     # a stretch of [1, 2)
                        np.arange(0x40000000, 0x40000000 + 70000, dtype=np.uint32).view(np.float32)])    # ... and of [2, 4)
    s = np.sqrt(x)
    assert s.dtype == np.float32
    t = preimage_floor(s)
    below = (t.view(np.uint32) - np.uint32(1)).view(np.float32)
    assert np.array_equal(np.sqrt(t), s)            # t itself rounds to s ...
    assert np.all(np.sqrt(below) < s)               # ... the float below it does not ...
    assert np.all(t <= x)                           # ... and the value we started from is one of the floats at or above it
    # at most a handful of floats share a square root: the set [t, x] is small
    assert int(((x.view(np.uint32) - t.view(np.uint32)).astype(np.int64)).max()) <= 3


def test_min_and_sqrt_commute_for_correctly_rounded_sqrt():
    """min(run, sqrt(d)) == sqrt(min(run^2, d)) when run^2 is exactly the square the running value came from -- the identity that lets
    the kernel keep the running distance squared (its initial 1e5 is sqrt(1e10) exactly)."""
    rng = np.random.default_rng(6)
    a = rng.uniform(0, 3, 100000).astype(np.float32)
    b = rng.uniform(0, 3, 100000).astype(np.float32)
    assert np.array_equal(np.minimum(np.sqrt(a), np.sqrt(b)), np.sqrt(np.minimum(a, b)))
    assert np.sqrt(np.float32(1e10)) == np.float32(1e5) and np.float32(1e5) * np.float32(1e5) == np.float32(1e10)
