#!/bin/bash
# gpurun -- bash tools/r04_knob_sweep.sh : cfg2 at 24 in flight / stacks of eight against V1's workgroups, iterations per graph, hardware queues
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
run() { # label, env..., -- args
  label=$1; shift
  env "$@" timeout 300 python bench.py --steps 24 --warmup 0 --no-cpu-baseline --no-single --no-f32 --no-other-configs $EXTRA > gpurun_out/r04/knob_$label.log 2>&1
  python - gpurun_out/r04/knob_$label.log "$label" <<'PY'
import json, sys
try:
    d = json.loads([x for x in open(sys.argv[1]) if x.startswith('{"metric"')][-1]); print(sys.argv[2], '->', round(d['value'], 2), 'clouds/s')
except Exception as e:
    print(sys.argv[2], '-> no line', e)
PY
}
EXTRA=""
run base HITADV_NOOP=1
run v1_96 HITADV_V1_BLOCKS_IN_FLIGHT=96
run v1_192 HITADV_V1_BLOCKS_IN_FLIGHT=192
run v1_256 HITADV_V1_BLOCKS_IN_FLIGHT=256
run q16 GPU_MAX_HW_QUEUES=16
EXTRA="--iters-per-graph 5"; run ipg5 HITADV_NOOP=1
EXTRA="--iters-per-graph 25"; run ipg25 HITADV_NOOP=1
EXTRA="--iters-per-graph 50"; run ipg50 HITADV_NOOP=1
