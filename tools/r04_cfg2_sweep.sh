#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04; mkdir -p $OUT; : > $OUT/cfg2_sweep.txt
run() { # in_flight per_stack
  HITADV_STACK=$2 timeout 400 python bench.py --concurrent $1 --steps $((2*$1)) --warmup 0 --no-cpu-baseline --no-single --no-f32 --no-other-configs > $OUT/sweep_cfg2_$1_$2.log 2>&1
  python - $OUT/sweep_cfg2_$1_$2.log "in_flight=$1 per_stack=$2" >> $OUT/cfg2_sweep.txt <<'PY'
import json, sys
try:
    d = json.loads([x for x in open(sys.argv[1]) if x.startswith('{"metric"')][-1])
    print(sys.argv[2], '->', round(d['value'], 2), 'clouds/s', d['end_to_end']['us_per_iteration'], 'us')
except Exception as e:
    print(sys.argv[2], '-> no line', e)
PY
}
run 12 4; run 16 4; run 24 8; run 24 6; run 18 6; run 12 3; run 24 4
cat $OUT/cfg2_sweep.txt
