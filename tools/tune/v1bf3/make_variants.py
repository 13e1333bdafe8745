#!/usr/bin/env python3
"""Ablation variants of the bf16x3 V1 kernel (csrc/victim_bf3.hip): each removes ONE cost from the tile loop by a textual
edit of the product source, compiles it with the timing harness into tools/tune/v1bf3/<name>; run them on the GPU box
with run.sh.  What a variant loses in time is what that cost contributes (results of the variants are wrong by design)."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(os.path.dirname(here)))
src = open(os.environ.get('V1BF3_SRC', os.path.join(root, 'hit_adv_amd/csrc/victim_bf3.hip'))).read()
OLD = 'V1BF3_SRC' in os.environ  # the 32x32x16 form of the kernel (git history), for comparison
harness = open(os.path.join(here, 'harness.inc')).read()

def sub(s, old, new):
    assert s.count(old) == 1, (old, s.count(old))
    return s.replace(old, new)

NO_SPLIT = ('      for (int i = 0; i < 8; ++i) split3(a[i], p1[i], p2[i], p3[i]);',
            '      for (int i = 0; i < 8; ++i) p1[i] = p2[i] = p3[i] = __float_as_uint(a[i]);')
SCAN_OLD = src[src.index('    const bool ragged = n0 + (tile + 1)'):src.index('    const bool g = tv > bv;  // earlier tiles' if OLD else '  // tile t: stA holds tile t+1')]
NO_SCAN = (SCAN_OLD, '''    for (int ct = 0; ct < 2; ++ct) {
      float tv = (acc[0][ct][0] + acc[1][ct][0]) + (acc[2][ct][0] + acc[3][ct][0]);
      const bool g = tv > bv[ct];
      bv[ct] = g ? tv : bv[ct];
      bi[ct] = g ? tile * 16 : bi[ct];
    }
  };
''')
# only the first two units' fragments are read; later units reuse them (two sets, so that the units of a pair differ and
# the compiler cannot merge their MFMAs)
NO_AREAD = ('      if (u + 1 < 2 * NSL) {', '      if (u + 1 < 2 * NSL && u < 1) {')
FA_FIX = None
FA_FIX2 = None
NO_BARRIER = ('    if (more && !late) stash(have, tile + 1);\n    __syncthreads();', '    if (more && !late) stash(have, tile + 1);')
NO_STAGGER = ('  const bool late = wave >= 4;', '  const bool late = false;')
ALL_LATE = ('  const bool late = wave >= 4;', '  const bool late = true;')
STAMP0 = ('  extern __shared__ __attribute__((aligned(16))) char sB3[];  // 2 buffers x 3 pieces x PIECE',
          '  extern __shared__ __attribute__((aligned(16))) char sB3[];\n  const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();')
STAMP1 = ('  if (S == 1) return;',
          '  if (threadIdx.x == 0) { g_stamp[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st0; g_stamp[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - sr0; }\n  if (S == 1) return;')
STAMPG = ('namespace hitadv {\n', '__device__ unsigned long long g_stamp[4096];\nnamespace hitadv {\n')
CLK = [STAMP0, STAMP1, STAMPG]
HALF = ('        for (int q = 0; q < 6; ++q) fa[(u + 1) & 1][q] = frag(u + 1, q);', '        for (int q = 0; q < 3; ++q) fa[(u + 1) & 1][q] = frag(u + 1, q);\n        for (int q = 3; q < 6; ++q) fa[(u + 1) & 1][q] = fa[u & 1][q];')
BCAST = ('    const char *base = sB3 + (size_t)(tile & 1) * 3 * PIECE + l16 * RS + 16 * g4;', '    const char *base = sB3 + (size_t)(tile & 1) * 3 * PIECE;')
INTER = [('      __builtin_amdgcn_sched_barrier(0);  // the reads stay above this unit\'s MFMAs', ''),
         ('            acc[2 * rp + x][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[x][PA[t]], b[ct][PB[t]], acc[2 * rp + x][ct], 0, 0, 0);\n    }',
          '            acc[2 * rp + x][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[x][PA[t]], b[ct][PB[t]], acc[2 * rp + x][ct], 0, 0, 0);\n'
          '      for (int i = 0; i < 6; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }\n    }')]
LAT = [('#pragma unroll\n    for (int q = 0; q < 6; ++q) fa[0][q] = frag(0, q);\n',
        '    const unsigned long long lt0 = __builtin_amdgcn_s_memtime();\n#pragma unroll\n    for (int q = 0; q < 6; ++q) fa[0][q] = frag(0, q);\n'
        '    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)\n    lat_sum += __builtin_amdgcn_s_memtime() - lt0;\n'),
       ('  float bv[2] = {', '  unsigned long long lat_sum = 0;\n  float bv[2] = {'),
       ('  if (S == 1) return;', '  if (lane == 0) g_stamp[1024 + blockIdx.x * 8 + wave] = lat_sum / ntiles;\n  if (S == 1) return;')]
UNUSED = [('a[x][p] = as_bf16x8(fa[u & 1][3 * x + p]);', 'a[x][p] = as_bf16x8(fa0[3 * x + p]);\n          asm volatile("" ::"v"(fa[u & 1][3 * x + p].x), "v"(fa[u & 1][3 * x + p].y), "v"(fa[u & 1][3 * x + p].z), "v"(fa[u & 1][3 * x + p].w));'),
          ('    for (int q = 0; q < 6; ++q) fa[0][q] = frag(0, q);\n', '    for (int q = 0; q < 6; ++q) fa[0][q] = frag(0, q);\n    uint4 fa0[6];\n#pragma unroll\n    for (int q = 0; q < 6; ++q) fa0[q] = frag(1, q);\n')]
ROW0 = ('    const char *base = sB3 + (size_t)(tile & 1) * 3 * PIECE + l16 * RS + 16 * g4;', '    const char *base = sB3 + (size_t)(tile & 1) * 3 * PIECE + 16 * lane;')
if OLD:
    NO_SCAN = (SCAN_OLD, '    float tv = acc0[0] + acc1[0];\n    int tc = 0;\n')
    NO_AREAD = ('      if (j + 1 < NSL) {', '      if (j + 1 < NSL && j < 0) {')
    FA_FIX = ('fa[j & 1][0]), f1 = as_bf16x8(fa[j & 1][1]), f2 = as_bf16x8(fa[j & 1][2]);', 'fa[0][0]), f1 = as_bf16x8(fa[0][1]), f2 = as_bf16x8(fa[0][2]);')
    FA_FIX_B = ('fa[j & 1][3]), g1 = as_bf16x8(fa[j & 1][4]), g2 = as_bf16x8(fa[j & 1][5]);', 'fa[0][3]), g1 = as_bf16x8(fa[0][4]), g2 = as_bf16x8(fa[0][5]);')
    STAMP0 = (STAMP0[0], STAMP0[1])
    variants = {'old_clk_base': CLK, 'old_clk_noaread': CLK + [NO_AREAD, FA_FIX, FA_FIX_B], 'old_clk_mfmaonly': CLK + [NO_SPLIT, NO_SCAN, NO_AREAD, FA_FIX, FA_FIX_B], 'old_clk_noscan': CLK + [NO_SCAN]}
else:
    variants = {
    'base': [],
    'nosplit': [NO_SPLIT],
    'noscan': [NO_SCAN],
    'noaread': [NO_AREAD],
    'nobarrier': [NO_BARRIER],
    'nostagger': [NO_STAGGER],
    'alllate': [ALL_LATE],
    'mfmaonly': [NO_SPLIT, NO_SCAN, NO_AREAD],
    'clk_base': CLK,
    'clk_half': CLK + [HALF],
    'clk_bcast': CLK + [BCAST],
    'clk_linear': CLK + [ROW0],
    'clk_unused': CLK + UNUSED,
    'clk_lat': [STAMP0, STAMPG, STAMP1] + LAT,
    'clk_inter': CLK + INTER,
    'clk_noaread': CLK + [NO_AREAD],
    'clk_mfmaonly': CLK + [NO_SPLIT, NO_SCAN, NO_AREAD],
    'clk_noscan': CLK + [NO_SCAN],
    }
only = sys.argv[1:]
procs = []
for name, edits in variants.items():
    if only and name not in only:
        continue
    s = src
    for old, new in edits:
        s = sub(s, old, new)
    s = s.replace('#include "common.hpp"', '#include "%s/hit_adv_amd/csrc/common.hpp"' % root)
    s = s.replace('#include "hitadv.h"', '#include "%s/include/hitadv.h"' % root)
    path = '/tmp/v1bf3_%s.hip' % name
    open(path, 'w').write('#define VARIANT "%s"\n' % name + ('#define CLOCKS 1\n' if name.startswith('clk_') else '') + s + '\n' + harness)
    procs.append((name, subprocess.Popen(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fno-slp-vectorize', path, '-o', os.path.join(here, name)])))
    if len(procs) % 4 == 0:
        for n, p in procs[-4:]:
            p.wait()
for n, p in procs:
    print(n, p.wait())
