#!/usr/bin/env python3
"""How much of a rocprofv3 kernel trace overlaps: sum of kernel durations against the time at least one kernel runs, the
largest kernels, and the time nothing runs.   python tools/trace_overlap.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import json
import sys

rows = []
for path in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:70]))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
for s, e, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
total = sum(e - s for s, e, _ in rows)
by = {}
for s, e, n in rows:
    d = by.setdefault(n, [0, 0])
    d[0] += e - s
    d[1] += 1
top = sorted(by.items(), key=lambda kv: -kv[1][0])[:25]
print(json.dumps(dict(kernels=len(rows), span_ms=(t1 - t0) / 1e6, busy_ms=busy / 1e6, idle_ms=(t1 - t0 - busy) / 1e6,
                      sum_of_durations_ms=total / 1e6, mean_concurrency_while_busy=round(total / busy, 3),
                      top=[dict(kernel=k, ms=round(v[0] / 1e6, 2), calls=v[1], us_avg=round(v[0] / v[1] / 1e3, 1)) for k, v in top]), indent=1))
