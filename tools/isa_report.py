#!/usr/bin/env python3
"""STATIC figures of the hot kernels, read from gfx950 assembly compiled with the Makefile's flags -- what a round without a GPU can say
about kernel quality (round 6: profiles/r06_isa_report.txt).  Not a measurement: registers, LDS, scratch and the instruction mix of the
loop regions that hold one tile's matrix instructions, plus, for V1, the cycle count of the additive model round 5 MEASURED on gfx950
(tools/tune/mfma16_valu_overlap.hip: a 16x16x32 fp16 MFMA issues every 16 cycles, an independent vector instruction between them costs
its 4 cycles on top -- they do not overlap).   python tools/isa_report.py [dir with the .s files, default: compile into a temp dir]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_scan  # noqa: E402

HOT = [  # (file, name fragment, MFMAs per tile region, what it is)
    ("victim_bf3", "linear_max_fwd_bf3_kILi128ELi2ELb1ELb0E", 96, "V1 flat fp16x2, shipped (52 % of cfg2's kernel time in round 5)"),
    ("victim_bf3", "linear_max_fwd_bf3_kILi128ELi2ELb1ELb1E", 96, "V1 flat fp16x2, arg-max search deferred (HITADV_V1_DEFER=1; never run)"),
    ("pointnet", "rowmlp_stream_kILi1E", None, "V2 streaming, stage 1"),
    ("pointnet", "rowmlp_stream_kILi2E", None, "V2 streaming, stage 2"),
    ("pointnet", "rowmlp_fwd16_kILi0E", None, "V2 stage 0 (deformation inside)"),
    ("pointnet", "rowmlp_bwd16_kILi1ELi2ELb0E", None, "V3 stage 1"),
    ("pointnet", "rowmlp_bwd16_kILi0ELi2ELb0E", None, "V3 stage 0"),
    ("pointnet", "rowmlp_bwd16_kILi2ELi2ELb0E", None, "V3 stage 2"),
    ("deform", "deform_bwd", None, "deformation backward"),
    ("pairwise", "pairwise3_vec4ILi1E", None, "K1 (the roofline kernel)"),
    ("pairwise", "nn_min3ILi0E", None, "K2 direct form"),
]


def compile_all(out):
    mk = open(os.path.join(ROOT, "hit_adv_amd", "csrc", "Makefile")).read()
    flags = re.search(r"^CXXFLAGS \?= (.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").split()
    flags = [f.replace("../../include", os.path.join(ROOT, "include")) for f in flags if f != "-fPIC"]
    for stem in sorted(set(h[0] for h in HOT)):
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-S", "--cuda-device-only", os.path.join(ROOT, "hit_adv_amd", "csrc", stem + ".hip"),
                               "-o", os.path.join(out, stem + ".s")], stderr=subprocess.DEVNULL)


def resources(path):
    text = open(path).read()
    out = {}
    for item in re.split(r"\n  - (?=\.)", text[text.index("amdhsa.kernels:"):])[1:]:
        g = lambda key: re.search(r"\.%s:\s+(\S+)" % key, item)  # noqa: E731
        if g("name") and g("vgpr_count"):
            out[g("name").group(1)] = dict(vgpr=int(g("vgpr_count").group(1)), sgpr=int(g("sgpr_count").group(1)),
                                           lds=int(g("group_segment_fixed_size").group(1)), scratch=int(g("private_segment_fixed_size").group(1)),
                                           spills=int(g("vgpr_spill_count").group(1)))
    return out


def main():
    d = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp()
    if len(sys.argv) <= 1:
        compile_all(d)
    print("STATIC analysis of HEAD's kernels (hipcc -S with the Makefile's flags) -- NOT a measurement; round 6 had no GPU.\n")
    for stem, frag, mfmas, what in HOT:
        path = os.path.join(d, stem + ".s")
        res = {k: v for k, v in resources(path).items() if frag in k}
        for name, r in sorted(res.items()):
            ins = isa_scan.kernels(path)[name]
            mix = {}
            for _, t in ins:
                if not t.endswith(":"):
                    mix[isa_scan._category(t)] = mix.get(isa_scan._category(t), 0) + 1
            print("%s\n    %s" % (what, name[:118]))
            print("    registers %d vector / %d scalar, static LDS %d B, scratch %d B, spills %d; whole kernel: %s" % (
                r["vgpr"], r["sgpr"], r["lds"], r["scratch"], r["spills"], ", ".join("%d %s" % (v, k) for k, v in sorted(mix.items()))))
            if mfmas:
                for m in isa_scan.tile_regions(path, frag, mfmas):
                    cyc = 16 * m["mfma"] + 4 * m["valu"]
                    print("    one tile (loop region with %d MFMAs): %d vector, %d scalar, %d LDS, %d global  ->  16 x %d + 4 x %d = %d cycles per tile and wave (additive model)" % (
                        mfmas, m["valu"], m.get("salu", 0), m.get("lds", 0), m.get("vmem", 0), m["mfma"], m["valu"], cyc))
            print()


if __name__ == "__main__":
    main()
