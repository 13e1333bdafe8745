#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04; mkdir -p $OUT; : > $OUT/inflight_sweep.txt
run() { # config concurrent steps
  HITADV_UNSTACKED_IN_FLIGHT=$2 timeout 500 python bench.py --config $1 --concurrent $2 --steps $3 --warmup 0 --no-cpu-baseline > $OUT/sweep_$1_$2.log 2>&1
  python - $OUT/sweep_$1_$2.log "$1 in_flight=$2 steps=$3" >> $OUT/inflight_sweep.txt <<'PY'
import json, sys
try:
    d = json.loads([x for x in open(sys.argv[1]) if x.startswith('{"metric"')][-1])
    print(sys.argv[2], '->', round(d['value'], 3), 'clouds/s')
except Exception as e:
    print(sys.argv[2], '-> no line', e)
PY
}
run cfg3 6 6; run cfg3 8 8; run cfg3 12 12; run cfg4 3 3; run cfg4 4 4; run cfg4 6 6
cat $OUT/inflight_sweep.txt
