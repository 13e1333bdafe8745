#!/usr/bin/env python3
"""Which kernels of two builds differ in their INSTRUCTIONS: `python tools/isa_diff.py dirA dirB` compares, file by file and kernel by
kernel, the gfx950 assembly of two trees of `hipcc -S --cuda-device-only` outputs (labels normalised, comments and directives dropped).
Round 6 used it to state what HEAD's library changes against the last library that ran on a GPU (docs/kernels/round6.md section 5)."""
import glob
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_scan  # noqa: E402


def bodies(path):
    out = {k: [re.sub(r"\.LBB\d+_", ".LBB_", t) for _, t in v] for k, v in isa_scan.kernels(path).items()}
    return {k: v for k, v in out.items() if any(t.startswith("s_endpgm") for t in v)}  # functions only, no data symbols


def main(a, b):
    names = sorted(set(os.path.basename(p) for p in glob.glob(os.path.join(a, "*.s"))) | set(os.path.basename(p) for p in glob.glob(os.path.join(b, "*.s"))))
    total = same = 0
    for n in names:
        pa, pb = os.path.join(a, n), os.path.join(b, n)
        if not (os.path.exists(pa) and os.path.exists(pb)):
            print("%-20s only in %s" % (n, a if os.path.exists(pa) else b))
            continue
        ka, kb = bodies(pa), bodies(pb)
        gone, new = sorted(set(ka) - set(kb)), sorted(set(kb) - set(ka))
        diff = sorted(k for k in set(ka) & set(kb) if ka[k] != kb[k])
        total += len(set(ka) | set(kb))
        same += len(set(ka) & set(kb)) - len(diff)
        if gone or new or diff:
            print("%-20s %3d kernels: %d differ, %d only in A, %d only in B" % (n, len(set(ka) | set(kb)), len(diff), len(gone), len(new)))
            for k in diff:
                print("      differs  %s  (%d -> %d instructions)" % (k[:110], len(ka[k]), len(kb[k])))
            for k in gone[:4]:
                print("      only A   %s" % k[:110])
            for k in new[:4]:
                print("      only B   %s" % k[:110])
        else:
            print("%-20s %3d kernels: identical instructions" % (n, len(ka)))
    print("identical: %d of %d kernels" % (same, total))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
