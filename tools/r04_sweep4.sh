#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
out=gpurun_out/r04/sweep4.txt; : > $out
run() { # label, steps, concurrent, env...
  label=$1; steps=$2; conc=$3; shift 3
  env "$@" timeout 400 python bench.py --steps $steps --warmup 0 --concurrent $conc --no-cpu-baseline --no-single --no-f32 --no-other-configs > gpurun_out/r04/sw4_$label.log 2>&1
  python - gpurun_out/r04/sw4_$label.log "$label" >> $out <<'PY'
import json, sys
try:
    d = json.loads([x for x in open(sys.argv[1]) if x.startswith('{"metric"')][-1]); print(sys.argv[2], '->', round(d['value'], 2), 'clouds/s')
except Exception as e:
    print(sys.argv[2], '-> no line', e)
PY
}
run v128 24 24 HITADV_NOOP=1
run v256 24 24 HITADV_V1_BLOCKS_IN_FLIGHT=256
run v224 24 24 HITADV_V1_BLOCKS_IN_FLIGHT=224
run v128b 24 24 HITADV_NOOP=1
run v256_s12 36 36 HITADV_V1_BLOCKS_IN_FLIGHT=256 HITADV_STACK=12
cat $out
