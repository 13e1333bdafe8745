"""A static guard for round 5's finding (docs/kernels/round5.md section 8): the shipped fps_lean must not contain packed f32 arithmetic
(v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) -- a build that had it lost single running-distance updates beside other streams' kernels.
The kernel source uses plain instructions on purpose; this test compiles csrc/sampling.hip to gfx950 assembly with the Makefile's flags
(hipcc cross-compiles without a GPU) and reads the ISA, so that neither an edit nor a change of flags (the compiler SLP-packs adjacent
scalar f32 operations when -fno-slp-vectorize is dropped) brings the packed instructions back unnoticed."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_shipped_fps_lean_has_no_packed_f32_arithmetic(tmp_path):
    mk = open(os.path.join(ROOT, "hit_adv_amd", "csrc", "Makefile")).read()
    flags = re.search(r"^CXXFLAGS \?= (.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").split()
    assert "-fno-slp-vectorize" in flags and "-ffp-contract=off" in flags
    flags = [f.replace("../../include", os.path.join(ROOT, "include")) for f in flags if f != "-fPIC"]
    out = str(tmp_path / "sampling.s")
    subprocess.check_call([HIPCC] + flags + ["-S", "--cuda-device-only", os.path.join(ROOT, "hit_adv_amd", "csrc", "sampling.hip"), "-o", out],
                          stderr=subprocess.DEVNULL)
    cur, packed, seen = None, {}, set()
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
        if cur and "fps_lean" in cur:
            seen.add(cur)
            if re.search(r"\bv_pk_(add|mul|fma)_f32\b", line):
                packed[cur] = packed.get(cur, 0) + 1
    assert len(seen) >= 14  # both samplers, 4 and 8 waves, the point counts per lane of the launcher
    assert packed == {}, packed
