"""Small host models of bit tricks the kernels rely on (the GPU tests check the kernels; these pin the reasoning):

* v_perm_b32 selectors of the V3 gather (csrc/pointnet.hip::gather_rows16): from two packed gradient words (fp16 hi | fp16 lo << 16)
  0x05040100 assembles the two hi pieces, 0x07060302 the two lo pieces, entry 0 in the low half;
* unsigned order of fp32 bit patterns equals float order for values >= +0, and a NaN of either sign is larger than every finite
  pattern (csrc/sampling.hip::fps_lean keeps running distances as bit patterns and lets NaN distances lose every `min`);
* the two-piece fp16 split (csrc/pointnet.hip::split8v and friends): hi = fp16(a), lo = fp16(2048 (a - hi)) reconstructs a to 2^-22 |a|
  at worst (two roundings of 11-bit significands; 2^-24 typically) inside fp16's range."""
import numpy as np


def v_perm_b32(a, b, sel):
    """D.byte[i] = {a, b}.byte[sel.byte[i]] for selector values 0-7 (b supplies bytes 0-3, a bytes 4-7)."""
    src = [(b >> (8 * i)) & 0xff for i in range(4)] + [(a >> (8 * i)) & 0xff for i in range(4)]
    return sum(src[(sel >> (8 * i)) & 0xff] << (8 * i) for i in range(4))


def test_perm_selectors_of_the_packed_gradient_gather():
    w0, w1 = 0xB0B1A0A1, 0xD0D1C0C1          # (lo << 16 | hi) of entries 0 and 1
    assert v_perm_b32(w1, w0, 0x05040100) == 0xC0C1A0A1   # hi pieces: entry 0 in the low half, entry 1 in the high half
    assert v_perm_b32(w1, w0, 0x07060302) == 0xD0D1B0B1   # lo pieces likewise


def test_unsigned_order_of_float_bits_for_non_negative_values():
    rng = np.random.default_rng(1)
    a = np.abs(rng.standard_normal(100000)).astype(np.float32)
    b = np.abs(rng.standard_normal(100000) * 1e-3).astype(np.float32)
    b[:10] = 0.0
    au, bu = a.view(np.uint32), b.view(np.uint32)
    assert np.array_equal(au < bu, a < b) and np.array_equal(np.minimum(au, bu).view(np.float32), np.minimum(a, b))
    for nan_bits in (0x7FC00000, 0xFFC00000, 0x7F800001):
        assert nan_bits > np.float32(3.0e38).view(np.uint32) and nan_bits > np.float32(np.inf).view(np.uint32) - 1
    assert np.float32(1e10).view(np.uint32) > au.max()  # the initial running distance loses to every real one


def test_two_piece_fp16_split_reconstructs_to_fp32_roundoff():
    rng = np.random.default_rng(2)
    a = (rng.standard_normal(200000) * np.exp(rng.uniform(-6, 9, 200000))).astype(np.float32)
    a = a[np.abs(a) < 65000]
    hi = a.astype(np.float16)
    lo = ((a - hi.astype(np.float32)) * np.float32(2048)).astype(np.float16)
    rec = hi.astype(np.float64) + lo.astype(np.float64) / 2048.0
    big = np.abs(a) > 2.0 ** -3               # below that the lo piece runs into fp16's subnormals (absolute error 2^-36)
    assert np.all(np.abs(rec - a)[big] <= 2.0 ** -22 * np.abs(a)[big])
    assert np.all(np.abs(rec - a)[~big] <= 2.0 ** -25)
