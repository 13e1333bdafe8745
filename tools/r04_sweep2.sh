#!/bin/bash
# gpurun -- bash tools/r04_sweep2.sh : cfg2 after the V1 changes -- attacks per stack x attacks in flight x V1 workgroups
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
out=gpurun_out/r04/sweep2.txt; : > $out
run() { # label, steps, concurrent, env...
  label=$1; steps=$2; conc=$3; shift 3
  env "$@" timeout 400 python bench.py --steps $steps --warmup 0 --concurrent $conc --no-cpu-baseline --no-single --no-f32 --no-other-configs > gpurun_out/r04/sw2_$label.log 2>&1
  python - gpurun_out/r04/sw2_$label.log "$label" >> $out <<'PY'
import json, sys
try:
    d = json.loads([x for x in open(sys.argv[1]) if x.startswith('{"metric"')][-1]); print(sys.argv[2], '->', round(d['value'], 2), 'clouds/s')
except Exception as e:
    print(sys.argv[2], '-> no line', e)
PY
}
run s8_c24_v128 24 24 HITADV_NOOP=1
run s8_c24_v160 24 24 HITADV_V1_BLOCKS_IN_FLIGHT=160
run s8_c24_v192 24 24 HITADV_V1_BLOCKS_IN_FLIGHT=192
run s8_c24_v96 24 24 HITADV_V1_BLOCKS_IN_FLIGHT=96
run s8_c32_v128 32 32 HITADV_NOOP=1
run s6_c24_v128 24 24 HITADV_STACK=6
run s10_c30_v128 30 30 HITADV_STACK=10
run s12_c36_v128 36 36 HITADV_STACK=12
run s6_c30_v128 30 30 HITADV_STACK=6
cat $out
