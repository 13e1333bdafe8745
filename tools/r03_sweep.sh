#!/bin/bash
# Tuning sweep on the GPU box: hardware queues x attacks in flight x attacks per stack x V1 workgroups (bench.py, no side measurements).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03
for combo in "8 12 4 128" "8 12 4 256" "8 12 6 128" "8 16 4 128" "8 16 8 128" "8 8 4 128" "8 8 8 128" "8 24 8 128" "4 8 4 128"; do
  set -- $combo
  v=$(GPU_MAX_HW_QUEUES=$1 HITADV_STACK=$3 HITADV_V1_BLOCKS_IN_FLIGHT=$4 timeout 400 python bench.py --steps 24 --warmup 0 --concurrent $2 --no-cpu-baseline --no-single --no-f32 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(round(l['value'],2), l['end_to_end']['us_per_iteration'])")
  echo "queues=$1 in_flight=$2 per_stack=$3 v1_blocks=$4 -> $v" | tee -a gpurun_out/r03/sweep6.txt
done
