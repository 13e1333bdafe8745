#!/usr/bin/env python3
"""rocprofv3 --pmc passes of tools/loop_pmc_probe.py -> per-kernel counters of cfg2's stacked iteration.

    python tools/loop_pmc_summary.py OUT.json WRITE.csv FETCH.csv [SQ1.csv SQ2.csv ...]

Dispatches up to the last setup-only kernel (FPS / kNN tables) are dropped; of the rest a kernel belongs to the LOOP if it was
launched a multiple of 39 times (3 stacks x 13 iterations); every kernel's launch count is listed under `all_kernels`.  HBM bytes per MI355X_MICROARCH.md: counters in KiB, WRITE_SIZE exact, FETCH_SIZE doubled (gfx950 reports half
the bytes of a wide coalesced read stream); both raw figures are kept.  Everything is per STACKED iteration (eight attacks of
32 clouds); `per_b32_iteration` divides by eight."""
import collections
import csv
import json
import sys

ITER, STACK = 39, 8  # 3 stacks x 13 iterations; 8 attacks per stack


def load(path):
    per = collections.defaultdict(lambda: collections.defaultdict(float))  # (kernel, dispatch) -> counter -> value (summed over XCDs / SEs)
    rows = list(csv.DictReader(open(path)))
    # the setups (24 of them, all before the first iteration) launch the victim's kernels too (get_gradient), V1 on the very
    # grid the loop uses: everything up to the last dispatch of a setup-only kernel (FPS, the kNN tables) is not the loop
    setup_end = max([int(r['Dispatch_Id']) for r in rows if 'fps<' in r['Kernel_Name'] or 'knn_select' in r['Kernel_Name']] or [0])
    for r in rows:
        if int(r['Dispatch_Id']) <= setup_end:
            continue
        per[(r['Kernel_Name'] + ' grid ' + r.get('Grid_Size', '?'), r['Dispatch_Id'])][r['Counter_Name']] += float(r['Counter_Value'])
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for (k, _), cs in per.items():
        for c, v in cs.items():
            acc[k][c].append(v)
    return acc


def short(k):
    return k.split('(')[0].replace('void ', '') + (' grid ' + k.rsplit(' grid ', 1)[1] if ' grid ' in k else '')


def main():
    out_path, paths = sys.argv[1], sys.argv[2:]
    kernels = collections.defaultdict(dict)
    for path in paths:
        for k, cs in load(path).items():
            for c, v in cs.items():
                kernels[k][c] = dict(total=sum(v), launches=len(v))
    rows, total_w, total_f, everything = {}, 0.0, 0.0, {}
    for k, cs in sorted(kernels.items()):
        n = max(v['launches'] for v in cs.values())
        everything[short(k)[:100]] = n
        if n % ITER:
            continue
        per_iter = n // ITER
        row = dict(launches_per_stacked_iteration=per_iter)
        for c, v in cs.items():
            row[c + '_per_launch'] = v['total'] / v['launches']
        if 'WRITE_SIZE' in cs or 'FETCH_SIZE' in cs:
            w = cs.get('WRITE_SIZE', dict(total=0))['total'] / ITER * 1024
            f = cs.get('FETCH_SIZE', dict(total=0))['total'] / ITER * 1024
            row.update(write_bytes_per_stacked_iteration=int(w), fetch_bytes_raw_per_stacked_iteration=int(f),
                       hbm_bytes_per_stacked_iteration=int(w + 2 * f))
            total_w += w
            total_f += f
        rows[short(k)] = row
    summary = dict(
        note="tools/loop_pmc_probe.py under rocprofv3 --pmc (one pass per counter group), MI355X; a stacked iteration = 8 attacks x 32 "
             "clouds; loop kernels = those launched a multiple of %d times" % ITER,
        write_bytes_per_stacked_iteration=int(total_w), fetch_bytes_raw_per_stacked_iteration=int(total_f),
        hbm_bytes_per_stacked_iteration=int(total_w + 2 * total_f),
        per_b32_iteration=dict(hbm_bytes=int((total_w + 2 * total_f) / STACK), hbm_bytes_fetch_not_doubled=int((total_w + total_f) / STACK)),
        kernels=rows, all_kernels=everything)
    json.dump(summary, open(out_path, 'w'), indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k not in ('kernels', 'all_kernels')}))


if __name__ == '__main__':
    main()
