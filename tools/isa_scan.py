#!/usr/bin/env python3
"""Read gfx950 assembly (hipcc -S --cuda-device-only) kernel by kernel and find the instruction pair that round 5's failing fps_lean
build had and the shipped kernels must not have (docs/kernels/round5.md section 8, docs/kernels/round6.md section 1):

    a 32-bit vector write (v_mov_b32 and the like -- anything that is not itself a packed or 64-bit instruction) into ONE HALF of an
    aligned register pair, and DIRECTLY behind it (no instruction, no s_nop between) a packed f32 arithmetic instruction
    (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) that reads that pair as a source operand.

In the failing build that pair was  `v_mov_b32 v0, v20` / `v_pk_add_f32 v[20:21], v[14:15], v[0:1] op_sel_hi:[1,0]`: the centre's z copied
into the low half of a pair whose high half is another live value, read back by the very next instruction.  Whether the pair is the
cause is NOT established (tools/tune/pk_f32_probe.hip has not had a GPU to run on); until it is, no product kernel may contain it.
`scan(path)` returns {kernel: [(line number, writer, reader), ...]}; `packed_counts(path)` the packed-f32 instructions per kernel.
Used by tests/test_isa_guards.py; `python tools/isa_scan.py file.s ...` prints a summary."""
import re
import sys

PACKED = re.compile(r"^v_pk_(add|mul|fma)_f32\b")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
LABEL = re.compile(r"^(_Z\w+):")


def _regs(text):
    """VGPR numbers named by an operand string, as (first, last) ranges."""
    out = []
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.append((int(m.group(1)), int(m.group(1))))
        else:
            out.append((int(m.group(2)), int(m.group(3))))
    return out


def _split(ins):
    """(mnemonic, destination operand text, source operand text) of one instruction line; modifiers stay with the sources."""
    parts = ins.split(None, 1)
    if len(parts) == 1:
        return parts[0], "", ""
    ops = parts[1].split(",", 1)
    return parts[0], ops[0], ops[1] if len(ops) > 1 else ""


def kernels(path):
    """{mangled kernel name: [(line number, instruction text)]}: instructions only, labels kept as ('label') breaks."""
    out, cur = {}, None
    for no, line in enumerate(open(path), 1):
        m = LABEL.match(line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is None:
            continue
        text = line.split(";")[0].strip()
        if text.startswith(".Lfunc_end") or text.startswith(".section"):
            cur = None
            continue
        if not text or text.startswith(".") and not text.startswith(".LBB"):
            continue
        out[cur].append((no, text))
    return {k: v for k, v in out.items() if v}


def is_vector_write32(mn):
    """A vector instruction that writes one 32-bit VGPR: not packed, not 64-bit, not a memory / LDS / MFMA / compare-to-SGPR one."""
    if not mn.startswith("v_") or mn.startswith("v_pk_") or mn.startswith("v_mfma") or mn.startswith("v_cmp") or mn.startswith("v_smfmac"):
        return False
    if re.search(r"_(b64|f64|u64|i64)(_e32|_e64|_dpp|_sdwa)?$", mn) or mn.startswith("v_readlane") or mn.startswith("v_readfirstlane"):
        return False
    return True


def scan(path):
    hits = {}
    for name, ins in kernels(path).items():
        for (n0, a), (n1, b) in zip(ins, ins[1:]):
            mb = b.split(None, 1)[0]
            if not PACKED.match(mb) or a.endswith(":"):
                continue
            ma, dst_a, _ = _split(a)
            if not is_vector_write32(ma):
                continue
            wrote = _regs(dst_a)
            if len(wrote) != 1 or wrote[0][0] != wrote[0][1]:
                continue
            w = wrote[0][0]
            _, _, src_b = _split(b)
            for lo, hi in _regs(src_b.split(" op_sel")[0].split(" neg_")[0]):
                if lo <= w <= hi:
                    hits.setdefault(name, []).append((n1, a, b))
                    break
    return hits


MEMORY = ("global_", "buffer_", "flat_", "ds_", "scratch_")


def scan_hi_half_forwarding(path):
    """The hazard the compiler cannot pad inside inline asm (LLVM's 'dst_sel forwarding' rule for gfx940+: a VALU write of one HALF of
    a register -- here v_fma_mixhi_f16, the second instruction of the fp16x2 split -- needs one wait state before the next VALU reads
    that register).  {kernel: [(line, writer, reader)]} for a VECTOR-ALU reader directly behind the write; memory and LDS readers are
    interlocked by the hardware and do not count."""
    hits = {}
    for name, ins in kernels(path).items():
        for (n0, a), (n1, b) in zip(ins, ins[1:]):
            if not a.startswith("v_fma_mixhi_f16") or b.endswith(":"):
                continue
            mb, db, sb = _split(b)
            if not mb.startswith("v_") or mb.startswith(MEMORY):
                continue
            w = _regs(_split(a)[1])[0][0]
            reads = _regs(sb) + (_regs(db) if mb.startswith("v_fma_mixhi_f16") or mb.startswith("v_fma_mixlo_f16") else [])
            if any(lo <= w <= hi for lo, hi in reads):
                hits.setdefault(name, []).append((n1, a, b))
    return hits


def _category(t):
    if t.startswith("v_mfma") or t.startswith("v_smfmac"):
        return "mfma"
    if t.startswith("v_"):
        return "valu"
    if t.startswith("s_"):
        return "salu"
    if t.startswith("ds_"):
        return "lds"
    if t.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def loop_regions(ins):
    """[(first, last)] instruction index ranges of the natural loops of one kernel: a backward branch and its target label."""
    labels = {t[:-1]: i for i, (_, t) in enumerate(ins) if t.endswith(":")}
    out = []
    for i, (_, t) in enumerate(ins):
        m = re.match(r"s_c?branch\w*\s+(\.LBB\w+)", t)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            out.append((labels[m.group(1)], i))
    return out


def tile_regions(path, kernel_fragment, mfmas):
    """Instruction mix of the smallest loop regions of a kernel that hold exactly `mfmas` matrix instructions -- one tile's worth:
    [{category: count}], one per region (the kernels keep two copies of a tile, one per wave half's schedule)."""
    out = []
    for name, ins in kernels(path).items():
        if kernel_fragment not in name:
            continue
        for a, b in loop_regions(ins):
            mix = {}
            for _, t in ins[a:b + 1]:
                if not t.endswith(":"):
                    mix[_category(t)] = mix.get(_category(t), 0) + 1
            if mix.get("mfma") == mfmas:
                out.append(mix)
    return out


def packed_counts(path):
    return {name: sum(1 for _, t in ins if PACKED.match(t)) for name, ins in kernels(path).items()}


if __name__ == "__main__":
    for p in sys.argv[1:]:
        h, c = scan(p), packed_counts(p)
        print("%s: %d kernels, %d with packed f32 arithmetic (%d instructions), %d kernels with the suspect pair (%d sites)" % (
            p, len(c), sum(1 for v in c.values() if v), sum(c.values()), len(h), sum(len(v) for v in h.values())))
        for k, v in sorted(h.items()):
            print("   ", k, len(v))
            for n, a, b in v[:3]:
                print("        line %d: %s  ->  %s" % (n, a, b))
