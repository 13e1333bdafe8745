"""The arithmetic of the bit-exact kernels, from the kernels' OWN SOURCE, on the CPU.

hit_adv_amd/csrc/arith.hpp holds the four squared-distance forms, PCT's distance, the exact tie threshold of PCT's sampler and the
three-piece bf16 split as plain C++; common.hpp includes it for gfx950, and tests/native/arith_host.cpp includes the same file for the
host.  Here that host build (g++ -O2 -ffp-contract=off, the library's own contraction setting) is compared BIT FOR BIT with the C oracle
(oracle/pointnet2_oracle.c::pair_value -- independent code, itself pinned against torch's own arithmetic in tests/test_oracle_gram.py and
against the reference's fixtures) on Gaussian, surface-like, near-duplicate, tiny, huge and exactly tied inputs, and with exact integer /
float64 arithmetic where there is no oracle.  The chain `reference = torch = oracle = kernel source` is thereby closed without a GPU;
what the -m gpu tests add is that the hardware's add / mul / fma / sqrt round as IEEE says.  (Unlike docs/models/, nothing here is a
re-description: edit arith.hpp and this test sees it.)"""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from oracle import c_oracle as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GXX = shutil.which("g++")
pytestmark = pytest.mark.skipif(GXX is None, reason="no g++")


@pytest.fixture(scope="module")
def H(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("arith") / "libarith_host.so")
    flags = ["-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "hit_adv_amd", "csrc")]
    if "fma" in open("/proc/cpuinfo").read().split("flags", 1)[-1].split("\n", 1)[0].split():
        flags.append("-mfma")  # fmaf() as one instruction; without it glibc's correctly rounded fmaf: the same bits, slower
    subprocess.check_call([GXX] + flags + [os.path.join(ROOT, "tests", "native", "arith_host.cpp"), "-o", so])
    lib = ctypes.CDLL(so)
    lib.arith_pairwise.argtypes = [ctypes.c_int, ctypes.c_long, ctypes.c_long] + [ctypes.c_void_p] * 3
    return lib


def _clouds(seed):
    g = torch.Generator().manual_seed(seed)
    gauss = torch.randn(300, 3, generator=g)
    gauss = (gauss - gauss.mean(0)) / gauss.norm(dim=1).max()
    sphere = torch.randn(300, 3, generator=g)
    sphere = sphere / sphere.norm(dim=1, keepdim=True) + 0.01 * torch.randn(300, 3, generator=g)
    near = gauss.clone()
    near[::2] = torch.nextafter(near[::2], torch.full_like(near[::2], 2.0))      # neighbours one ulp apart
    lattice = torch.randint(-3, 4, (300, 3), generator=g).float() * 0.25        # exact ties, exact zeros
    tiny = gauss * 1e-19                                                          # squares underflow into the subnormals
    huge = gauss * 3e18                                                           # squares near the top of the range (no overflow)
    mixed = torch.cat([gauss[:100], tiny[:100], lattice[:100]])
    return dict(gauss=gauss, sphere=sphere, near=near, lattice=lattice, tiny=tiny, huge=huge, mixed=mixed)


@pytest.mark.parametrize("form", [0, 1, 2, 3, 4])
def test_kernel_source_distance_forms_equal_the_oracle_bit_for_bit(H, form):
    """include/hitadv.h HITADV_FORM_DIRECT / GRAM / GRAM_KNN / SQUARE_DISTANCE and PCT's get_dists: util/set_distance.py:15-32,
    util/dist_utils.py:148-150, model/pointnet2_utils.py:19-41, util/other_utils.py:237-251 as the oracle restates them."""
    sets = _clouds(5 + form)
    checked = 0
    for xa, x in sets.items():
        for ya, y in sets.items():
            if xa > ya:
                continue
            x, y = x.contiguous(), y.contiguous()
            P = torch.empty(x.shape[0], y.shape[0])
            assert H.arith_pairwise(form, x.shape[0], y.shape[0], x.data_ptr(), y.data_ptr(), P.data_ptr()) == 0
            want = N.pairwise(x[None], y[None], form)[0]
            same = P.view(torch.int32) == want.view(torch.int32)
            both_nan = P.isnan() & want.isnan()
            assert bool((same | both_nan).all()), (form, xa, ya, int((~(same | both_nan)).sum()))
            checked += P.numel()
    assert checked > 2_000_000


def test_direct_form_is_symmetric_and_the_gram_form_is_not_assumed_to_be(H):
    """K2 evaluates (query, reference) pairs in both directions with ONE function: forms 0 and 1 must give d(x, y) == d(y, x) bit for bit
    (csrc/pairwise.hip::nn_min3: "forms 0 and 1 are symmetric in (query, reference)")."""
    s = _clouds(11)
    for form in (0, 1):
        for x in (s['gauss'], s['near'], s['mixed']):
            y = s['sphere']
            A, B = torch.empty(300, 300), torch.empty(300, 300)
            H.arith_pairwise(form, 300, 300, x.data_ptr(), y.data_ptr(), A.data_ptr())
            H.arith_pairwise(form, 300, 300, y.data_ptr(), x.data_ptr(), B.data_ptr())
            assert torch.equal(A.view(torch.int32), B.t().contiguous().view(torch.int32)), form


def test_three_piece_split_is_exact_and_bf16_representable(H):
    """csrc/victim_bf3.hip / knn.hip: a = hi + mid + lo EXACTLY, each piece a bf16 number (its low 16 bits zero), for every finite float
    whose third piece does not underflow -- the premise of "three bf16 pieces = fp32 accuracy" (six exact products per term)."""
    g = torch.Generator().manual_seed(3)
    a = torch.cat([torch.randn(200_000, generator=g), torch.randn(100_000, generator=g) * 1e-12, torch.randn(100_000, generator=g) * 1e12,
                   torch.tensor([0.0, -0.0, 1.0, -1.0, 65504.0, 3.4e38, -3.4e38, 1.17549435e-38])]).contiguous()
    n = a.numel()
    hi, mid, lo = (torch.empty(n, dtype=torch.int32) for _ in range(3))
    pk = torch.empty(n // 2, dtype=torch.int32)
    H.arith_split3(ctypes.c_long(n), ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(hi.data_ptr()), ctypes.c_void_p(mid.data_ptr()),
                   ctypes.c_void_p(lo.data_ptr()), ctypes.c_void_p(pk.data_ptr()))
    for piece in (hi, mid, lo):
        assert int((piece & 0xffff).abs().max()) == 0
    total = hi.view(torch.float32).double() + mid.view(torch.float32).double() + lo.view(torch.float32).double()
    assert torch.equal(total, a.double())
    h = hi.view(torch.float32)
    assert bool((h.abs() <= a.abs()).all()) and bool(((a - h).abs() <= a.abs() * 2.0 ** -7).all())  # truncation, 8 significant bits kept
    want = ((hi[0:n - 1:2].long() & 0xffffffff) >> 16) | (hi[1:n:2].long() & 0xffff0000)
    assert torch.equal(pk.long() & 0xffffffff, want)


def test_pct_tie_threshold_is_the_smallest_float_whose_sqrt_rounds_to_s(H):
    """csrc/arith.hpp::sqrt_preimage_floor (fps_lean's PCT sampler keeps the running distance SQUARED and takes one sqrt per wave and step;
    util/other_utils.py:254-272 compares the ROUNDED square roots): for s = sqrt_rn(x) > 0 the threshold t satisfies sqrt_rn(t) == s and
    sqrt_rn(pred(t)) < s.  numpy's float32 sqrt is correctly rounded."""
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.random(400_000, dtype=np.float32) * 4, (rng.random(100_000, dtype=np.float32) * 1e-6).astype(np.float32),
                        np.float32([1e-7, 1.0, 4.0, 2.0, 1e10, 3.0])])
    s = np.sqrt(x).astype(np.float32)
    s = np.ascontiguousarray(s[s > 0])
    out = np.empty_like(s)
    H.arith_sqrt_preimage_floor(ctypes.c_long(s.size), ctypes.c_void_p(s.ctypes.data), ctypes.c_void_p(out.ctypes.data))
    assert (np.sqrt(out).astype(np.float32) == s).all()
    below = np.nextafter(out, np.float32(0), dtype=np.float32)
    assert (np.sqrt(below).astype(np.float32) < s).all()



def test_kernel_source_adam_follows_torch_optim_adam(H):
    """csrc/arith.hpp::adam_coef / adam_update -- the one Adam step behind adam_k, adam2_k and the deformation's Adam tail -- stepped
    beside torch.optim.Adam (the optimiser the reference builds at HiT_ADV.py:139-145 with lr = 5 x and 3 x attack_lr; CW/*.py: lr =
    attack_lr) for 600 steps (> the headline's num_iter) on the same gradients.  NOT bit for bit, and it cannot be: torch's CPU kernels
    fuse the multiply-adds of lerp and addcmul (round 6 matched torch 2.10's moments bit for bit with m = fma(0.1, g - m, m) and v =
    fma(0.001 g, g, 0.999 v); unfused, 2 of 4,096 parameters are one ulp apart after the FIRST step and a quarter of the first moments
    after the second), the kernels round each operation as written -- that one-ulp freedom per step is where the
    measured "Adam drift" of tests/test_gpu_headline_parity.py starts.  What is held: for ordinary and for huge gradients the second
    moment stays within 4 ulps of torch's and the first within 1e-6 of the gradient scale at every checked step and the parameters within 2e-6 after 600 steps (a wrong bias correction, eps
    inside the square root or a missing 1 / bias_correction1 is off by orders of magnitude more); for exactly zero gradients everything
    is exactly torch's (nothing moves)."""
    H.arith_adam_step.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_double] + [ctypes.c_void_p] * 4
    g = torch.Generator().manual_seed(17)
    n = 3072
    ulps = lambda a, b: (a.view(torch.int32).long() - b.view(torch.int32).long()).abs()  # noqa: E731  (same-sign neighbours)
    for lr in (0.05, 0.03, 0.01):
        p_t = torch.randn(n, generator=g).requires_grad_()
        opt = torch.optim.Adam([p_t], lr=lr, weight_decay=0.)
        p = p_t.detach().clone().contiguous()
        m, v = torch.zeros(n), torch.zeros(n)
        scale = torch.cat([torch.ones(n // 3), torch.zeros(n // 3), torch.full((n // 3,), 1e6)])
        live = scale != 0
        for t in range(1, 601):
            grad = (torch.randn(n, generator=g) * scale).contiguous()
            p_t.grad = grad.clone()
            opt.step()
            H.arith_adam_step(n, t, lr, p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr())
            st = opt.state[p_t]
            if t <= 20 or t % 50 == 0 or t == 600:
                assert int(ulps(v, st['exp_avg_sq']).max()) <= 4, (lr, t)
                dm = (m - st['exp_avg']).abs()  # one rounding of (g - m) * 0.1 more or less: an error at the GRADIENT's scale, not at |m|'s
                assert bool((dm <= 1e-6 * scale).all()), (lr, t, float((dm / scale.clamp_min(1e-30)).max()))
                assert torch.equal(p[~live], p_t.detach()[~live]) and torch.equal(m[~live], st['exp_avg'][~live])
        assert float((p - p_t.detach())[live].abs().max()) <= 2e-6 * max(1.0, float(p_t.detach().abs().max())), lr


CLANGXX = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CLANGXX), reason="no host clang++ with _Float16")
def test_two_piece_fp16_split_is_what_the_kernels_headers_say(tmp_path):
    """csrc/arith.hpp::split_pair -- the split behind every fp16x2 product of the PointNet engine (csrc/victim_bf3.hip's header):
    a = hi + 2^-11 lo + r with hi = fp16(a), lo = fp16(2^11 (a - hi)).  From the real source, on the host: (1) the fused form is the
    spelled-out form's bits (the claim next to the function; victim_bf3.hip's staging and split_weights_k spell it out), for every input
    incl. fp16's subnormal range and beyond its top; (2) |r| <= 2^-22 |a| wherever |a| is inside fp16's NORMAL range (2^-14 <= |a| < 65520), about 2^-24.4 |a| in the
    root mean square and <= 2^-22.5 |a| at the 99.9th percentile; (3) hi is an infinity exactly from |a| >= 65520 on -- the premise of the range
    flag; (4) a product of two split values, a1 b1 + 2^-11 (a1 b2 + a2 b1), is within 3 x 2^-22 of a b."""
    so = str(tmp_path / "libarith_f16.so")
    subprocess.check_call([CLANGXX, "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC",
                           "-I" + os.path.join(ROOT, "hit_adv_amd", "csrc"), os.path.join(ROOT, "tests", "native", "arith_f16_host.cpp"), "-o", so])
    lib = ctypes.CDLL(so)
    rng = np.random.default_rng(1)
    a = np.concatenate([rng.standard_normal(1_000_000).astype(np.float32),
                        (rng.standard_normal(300_000) * 100).astype(np.float32),
                        (rng.standard_normal(300_000) * 1e-3).astype(np.float32),
                        (rng.standard_normal(200_000) * 1e-6).astype(np.float32),      # hi in fp16's subnormal range
                        (rng.random(200_000) * 7e4).astype(np.float32),                # up to and beyond fp16's top
                        np.float32([0.0, -0.0, 65504.0, 65519.99, 65520.0, -65520.0, 1e30, 6.1035156e-05, 5.96e-08, 2.9e-08])])
    a = np.ascontiguousarray(a)
    n = a.size
    hi, lo, lo2 = (np.empty(n, np.uint16) for _ in range(3))
    lib.arith_split_pair(ctypes.c_long(n), *(ctypes.c_void_p(t.ctypes.data) for t in (a, hi, lo, lo2)))
    h, l = hi.view(np.float16), lo.view(np.float16)
    finite = np.isfinite(h)
    assert (lo[finite] == lo2[finite]).all()                                   # (1) the same bits, fused or spelled out
    assert (np.isinf(h) == (np.abs(a) >= 65520.0)).all()                       # (3)
    normal = finite & (np.abs(a.astype(np.float64)) >= 2.0 ** -14) & np.isfinite(l)  # |a| inside fp16's NORMAL range: below it lo's own subnormal spacing bounds r
    with np.errstate(invalid='ignore'):  # (inf - inf where hi overflowed: outside `normal`)
        r = np.abs(a.astype(np.float64) - (h.astype(np.float64) + l.astype(np.float64) / 2048.0))
    rel = r[normal] / np.abs(a[normal].astype(np.float64))
    assert rel.max() <= 2.0 ** -22 and np.quantile(rel, 0.999) <= 2.0 ** -22.5   # (2)
    assert 2.0 ** -25.5 <= np.sqrt((rel ** 2).mean()) <= 2.0 ** -23.5
    k = 500_000                                                                 # (4) products of the first million (all normal here)
    x, y = slice(0, k), slice(k, 2 * k)
    ok = normal[x] & normal[y]
    ab = a[x].astype(np.float64) * a[y].astype(np.float64)
    h64, l64 = h.astype(np.float64), l.astype(np.float64)
    approx = h64[x] * h64[y] + (h64[x] * l64[y] + l64[x] * h64[y]) / 2048.0
    assert (np.abs(approx - ab)[ok] <= 3 * 2.0 ** -22 * np.abs(ab)[ok]).all()
