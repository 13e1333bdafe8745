#!/bin/bash
# gpurun -- bash tools/r04_pmc_gal.sh : HBM traffic of cfg4's roofline kernel (rows_linear_gather_k at sa1's shape), merged into kbench_traffic.json
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r04 /tmp/prof
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof/gal_$ctr
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof/gal_$ctr -- python3 tools/gal_probe.py > gpurun_out/r04/pmc_gal_$ctr.log 2>&1
  f=$(find /tmp/prof/gal_$ctr -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" /tmp/prof/gal_$ctr.csv
done
python tools/pmc_summary.py /tmp/prof/gal_WRITE_SIZE.csv /tmp/prof/gal_FETCH_SIZE.csv > gpurun_out/r04/gal_traffic.json
cat gpurun_out/r04/gal_traffic.json; tail -2 gpurun_out/r04/pmc_gal_WRITE_SIZE.log
