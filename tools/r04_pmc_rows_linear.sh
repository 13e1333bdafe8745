#!/bin/bash
# gpurun -- bash tools/r04_pmc_rows_linear.sh : HBM traffic of hitadv_rows_linear at cfg4's two levels (separate --pmc passes)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r04 /tmp/prof
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof/rl_$ctr
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof/rl_$ctr -- python3 tools/rows_linear_probe.py > gpurun_out/r04/pmc_rl_$ctr.log 2>&1
  f=$(find /tmp/prof/rl_$ctr -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" /tmp/prof/rl_$ctr.csv
done
python tools/pmc_summary.py /tmp/prof/rl_WRITE_SIZE.csv /tmp/prof/rl_FETCH_SIZE.csv > gpurun_out/r04/rows_linear_traffic.json
cat gpurun_out/r04/rows_linear_traffic.json
