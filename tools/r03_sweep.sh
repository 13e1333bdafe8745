#!/bin/bash
# Tuning sweep on the GPU box: attacks in flight x V1 workgroups (bench.py, 8 steps, no side measurements).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03
for combo in "4 128" "4 256" "4 64" "6 128" "8 128" "2 0" "8 64"; do
  set -- $combo
  v=$(HITADV_V1_BLOCKS_IN_FLIGHT=${2/#0/128} timeout 300 python bench.py --steps 8 --warmup 0 --concurrent $1 --no-cpu-baseline --no-single --no-f32 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(round(l['value'],2), l['end_to_end']['us_per_iteration'])")
  echo "in_flight=$1 v1_blocks=$2 -> $v" | tee -a gpurun_out/r03/sweep.txt
done
