#!/usr/bin/env python3
"""Summarise a rocprofv3 `--kernel-trace --stats --output-format csv` run: copy the *kernel_stats.csv to
`out` and print per-iteration launch counts / average durations.

    python tools/prof_summary.py <rocprof output dir> <out.csv> [iterations]
"""
import csv
import glob
import os
import shutil
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    iters = float(sys.argv[3]) if len(sys.argv) > 3 else 5003.0
    files = glob.glob(os.path.join(src, '**', '*kernel_stats.csv'), recursive=True)
    if not files:
        sys.exit('no kernel_stats.csv under %s' % src)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    shutil.copy(files[0], out)
    rows = list(csv.DictReader(open(out)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    calls = sum(int(r['Calls']) for r in rows)
    print('total kernel time %.2f s, calls/iter %.1f, sum of durations/iter %.1f us' % (tot / 1e9, calls / iters, tot / iters / 1e3))
    for r in rows[:45]:
        print('%5.2f%% %6.1f/it avg %7.1fus  %s' % (float(r['Percentage']), int(r['Calls']) / iters,
                                                   float(r['AverageNs']) / 1e3, r['Name'][:110]))


if __name__ == '__main__':
    main()
