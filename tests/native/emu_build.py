"""Build a kernel file of hit_adv_amd/csrc/ for the CPU wave64 emulator (tests/native/emu/simt_emu.hpp) -- test infrastructure.

The file's text is the product's; three constructs plain C++ cannot parse are rewritten, mechanically:
  1. `kernel<<<grid, block, lds, stream>>>(args)`  ->  `emu::launch(grid, block, lds, [=]() { kernel(args); })`
  2. `extern __shared__ T name[];`                 ->  `T *name = reinterpret_cast<T *>(emu::dyn_lds);`
  2b. operand-less `asm volatile("s_waitcnt ..." ::: "memory")` (the hand-off protocols' waits; also in the staged copies of the headers)  ->  nothing
  3. fps_lean's one LDS atomic written as inline asm (`ds_max_rtn_u64` + its wait)  ->  the same operation in C++
  3b. the fp16x2 split's three instructions written as asm (v_cvt_pk_f16_f32, v_fma_mixlo_f16, v_fma_mixhi_f16)  ->  the same operations in C++
  4. victim_bf3.hip's two scan helpers (compares into lane masks / selects on them, asm for their schedule)  ->  __ballot and a select
and the result is compiled by ROCm's clang++ FOR x86-64 (-ffp-contract=off, like the library) against the emulator header in place of
<hip/hip_runtime.h>.  The extern "C" entry points keep their names and signatures; "device" pointers are host pointers."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "hit_adv_amd", "csrc")
EMU = os.path.join(ROOT, "tests", "native", "emu")
CLANGXX = "/opt/rocm/lib/llvm/bin/clang++"


def _match(text, i, open_ch, close_ch):
    """index just past the bracket that closes the one at text[i]"""
    depth = 0
    for j in range(i, len(text)):
        if text[j] == open_ch:
            depth += 1
        elif text[j] == close_ch:
            depth -= 1
            if depth == 0:
                return j + 1
    raise ValueError("unbalanced %s at %d" % (open_ch, i))


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":  # (not < >: launch configurations hold comparisons, not template arguments)
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    out.append(cur.strip())
    return out


def rewrite_launches(text):
    out, pos, n = "", 0, 0
    while True:
        i = text.find("<<<", pos)
        if i < 0:
            return out + text[pos:], n
        # the kernel expression: an identifier (with namespaces), optionally template arguments, directly in front of <<<
        j = i
        while j > pos and text[j - 1].isspace():
            j -= 1
        if text[j - 1] == ">":  # template arguments: walk back to the matching <
            depth, k = 0, j - 1
            while True:
                if text[k] == ">":
                    depth += 1
                elif text[k] == "<":
                    depth -= 1
                    if depth == 0:
                        break
                k -= 1
            j = k
        k = j
        while k > pos and (text[k - 1].isalnum() or text[k - 1] in "_:"):
            k -= 1
        kernel = text[k:i].strip()
        e = text.find(">>>", i)
        cfg = _split_top(text[i + 3:e].replace("\\\n", " "))
        a = text.find("(", e)
        b = _match(text, a, "(", ")")
        args = text[a + 1:b - 1]
        grid, block = cfg[0], cfg[1]
        lds = cfg[2] if len(cfg) > 2 else "0"
        out += text[pos:k] + "emu::launch(dim3(%s), dim3(%s), (size_t)(%s), [=]() { %s(%s); })" % (grid, block, lds, kernel, args)
        pos = b
        n += 1


def rewrite_extern_shared(text):
    pat = re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?([\w:]+(?:\s+\w+)*?)\s+(\w+)\[\];")
    return pat.subn(lambda m: "%s *%s = reinterpret_cast<%s *>(emu::dyn_lds);" % (m.group(1), m.group(2), m.group(1)), text)


FPS_ASM = re.compile(r'asm volatile\("ds_max_rtn_u64 %0, %1, %2\\n\\ts_waitcnt lgkmcnt\(0\)" : "=v"\(before\) : "v"\(([^)]*\([^)]*\)[^)]*)\), "v"\(key\) : "memory"\);')


def rewrite_fps_asm(text):
    # the LDS address operand is key_at + 8 * j3 with key_at = the LDS address of s_key[0]: the emulator indexes the array itself
    return FPS_ASM.subn("{ before = s_key[j3]; if (key > s_key[j3]) s_key[j3] = key; }", text)


WAITCNT = re.compile(r'asm volatile\("s_waitcnt [^"]*"\s*:::\s*"memory"\);')


PKMAX = re.compile(r'asm\("v_pk_max_u16 %0, %1, %2" : "=v"\((\w+)\) : "v"\((\w+)\), "v"\((\w+)\)\);')


def rewrite_pk_max(text):
    """common.hpp::PieceWatch: a packed maximum of two pairs of unsigned 16-bit halves"""
    return PKMAX.subn(r"\1 = (((\2 & 0xffffu) > (\3 & 0xffffu) ? \2 : \3) & 0xffffu) | (((\2 >> 16) > (\3 >> 16) ? \2 : \3) & 0xffff0000u);", text)


def rewrite_waits(text):
    """operand-less `s_waitcnt` statements (memory-ordering waits of the hand-off protocols): nothing to wait for on one OS thread"""
    return WAITCNT.subn("((void)0);", text)


def stage_headers(out_dir):
    """the kernels' headers, copied beside the generated source with the same rewrites (they hold a wait and dynamic-LDS declarations)"""
    for name in os.listdir(CSRC):
        if name.endswith(".hpp"):
            text = open(os.path.join(CSRC, name)).read()
            text, _ = rewrite_waits(text)
            text, _ = rewrite_extern_shared(text)
            text, _ = rewrite_pk_max(text)
            with open(os.path.join(out_dir, name), "w") as f:
                f.write(text)


V1_CMP = re.compile(r'asm volatile\("v_cmp_eq_f32_e64 %0, %2, %3\\n\\tv_cmp_eq_f32_e64 %1, %4, %5" : "=s"\(ma\), "=s"\(mb\) : "v"\(a\), "v"\(ta\), "v"\(b\), "v"\(tb\)\);')
V1_SEL = re.compile(r'asm volatile\("v_cndmask_b32_e64 %0, %0, %4, %2\\n\\tv_cndmask_b32_e64 %1, %1, %4, %3" : "\+v"\(ca\), "\+v"\(cb\) : "s"\(ma\), "s"\(mb\), "n"\(Q\)\);')


def rewrite_v1_scan_asm(text):
    """csrc/victim_bf3.hip: two compares into SGPR pairs (= lane masks) / two selects on them, written as asm for their schedule"""
    text, a = V1_CMP.subn("ma = __ballot(a == ta); mb = __ballot(b == tb);", text)
    text, b = V1_SEL.subn("{ const int l__ = emu::cur->lin & 63; ca = (ma >> l__) & 1 ? Q : ca; cb = (mb >> l__) & 1 ? Q : cb; }", text)
    return text, a + b


OPND = r"([\w.]+(?:\[[^\]]+\])?)"  # H[p], v[2 * p + 1], h1.x
SPLIT_CVT = re.compile(r'asm\("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"\(' + OPND + r'\) : "v"\(' + OPND + r'\), "v"\(' + OPND + r'\)\);')
SPLIT_LO = re.compile(r'asm\("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:\[0,0,0\] op_sel_hi:\[1,0,0\]" : "=v"\(' + OPND + r'\) : "v"\(' + OPND + r'\), "s"\((\w+)\), "v"\((\w+)\)\);')
SPLIT_HI = re.compile(r'asm\("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:\[1,0,0\] op_sel_hi:\[1,0,0\]" : "\+v"\(' + OPND + r'\) : "v"\(' + OPND + r'\), "s"\((\w+)\), "v"\((\w+)\)\);')


def rewrite_split_asm(text):
    """the fp16x2 split's v_cvt_pk_f16_f32 / v_fma_mixlo_f16 / v_fma_mixhi_f16 (split8v and its copies) -> the same three operations in C++"""
    text, a = SPLIT_CVT.subn(r"\1 = emu_cvt_pk_f16_f32(\2, \3);", text)
    text, b = SPLIT_LO.subn(r"\1 = emu_fma_mix_f16((uint16_t)(\2 & 0xffffu), \3, \4);", text)
    text, c = SPLIT_HI.subn(r"\1 = (\1 & 0xffffu) | ((uint32_t)emu_fma_mix_f16((uint16_t)(\2 >> 16), \3, \4) << 16);", text)
    return text, (a, b, c)


LDS_DMA = re.compile(r'asm volatile\("s_mov_b32 m0, %0\\n\\ts_nop 0\\n\\tglobal_load_lds_dwordx4 %1, %2" ::"s"\(__builtin_amdgcn_readfirstlane\(\(int\)lds_at\)\), "v"\(voff\),\s*"s"\(u\)\s*:\s*"memory"\);', re.S)


def rewrite_lds_dma(text):
    """gemm16.hip's LDS-DMA (global_load_lds_dwordx4 through m0): LDS addresses are not 32-bit numbers here -- the ring kernel is not emulated
    (tests switch it off: hitadv_debug_g16_ring(0)); the statement becomes an abort so that the file still builds"""
    return LDS_DMA.subn('{ (void)u; (void)voff; (void)lds_at; fputs("simt_emu: LDS-DMA (the G16 ring kernel) is not emulated", stderr); abort(); }', text)


def build(stem, out_dir, extra_flags=()):
    """-> path of lib<stem>_emu.so built from hit_adv_amd/csrc/<stem>.hip"""
    stage_headers(out_dir)
    text = open(os.path.join(CSRC, stem + ".hip")).read()
    text, _ = rewrite_waits(text)
    text, n_launch = rewrite_launches(text)
    text, n_ext = rewrite_extern_shared(text)
    n_asm = 0
    if stem == "sampling":
        text, n_asm = rewrite_fps_asm(text)
        assert n_asm == 1, "fps_lean's LDS atomic was not found: the rewrite rule needs updating"
    text, n_split = rewrite_split_asm(text)
    assert 2 * n_split[0] == n_split[1] + n_split[2] and n_split[1] == n_split[2], "one conversion per mixlo / mixhi pair: %r" % (n_split,)
    if stem == "gemm16":
        text, n_dma = rewrite_lds_dma(text)
        assert n_dma == 1, "gemm16.hip's LDS-DMA statement was not found: the rewrite rule needs updating"
    if stem == "victim_bf3":
        text, n_asm = rewrite_v1_scan_asm(text)
        assert n_asm == 2, "V1's compare / select asm helpers were not found: the rewrite rule needs updating"
    assert n_launch > 0 and "<<<" not in text and "extern __shared__" not in text
    src = os.path.join(out_dir, stem + "_emu.cpp")
    with open(src, "w") as f:
        f.write("// GENERATED by tests/native/emu_build.py from hit_adv_amd/csrc/%s.hip (%d launches, %d dynamic LDS arrays, %d asm rewritten)\n" % (
            stem, n_launch, n_ext, n_asm))
        f.write(text)
    so = os.path.join(out_dir, "lib%s_emu.so" % stem)
    cmd = [CLANGXX, "-O1", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-w", "-I" + EMU, "-I" + out_dir, "-I" + CSRC,
           "-I" + os.path.join(ROOT, "include")] + list(extra_flags) + [src, "-o", so]
    subprocess.check_call(cmd)
    return so


if __name__ == "__main__":
    import sys
    import tempfile
    d = tempfile.mkdtemp()
    for s in sys.argv[1:]:
        print(build(s, d))
