#!/bin/bash
# gpurun -- bash tools/r04_bench_all.sh [tag] : the four configurations' bench lines (no CPU baseline legs)
set -u
cd "$GRAFT_REPO_ROOT"
TAG=${1:-x}
OUT=gpurun_out/r04; mkdir -p $OUT
timeout 400 python bench.py --steps 24 --warmup 12 --no-cpu-baseline --no-single --no-f32 --no-other-configs > $OUT/bench_cfg2_$TAG.log 2>&1
for c in cfg3 cfg4 cfg5; do timeout 500 python bench.py --config $c --no-cpu-baseline > $OUT/bench_${c}_$TAG.log 2>&1; done
python - $OUT $TAG <<'PY'
import json, sys
for c in ('cfg2', 'cfg3', 'cfg4', 'cfg5'):
    try:
        l = [x for x in open('%s/bench_%s_%s.log' % (sys.argv[1], c, sys.argv[2])) if x.startswith('{"metric"')][-1]
        d = json.loads(l)
        print(c, round(d['value'], 3), 'clouds/s', round(d['ms_per_step'], 1), 'ms/step', 'in flight', d['config'].get('attacks_in_flight_per_gpu'))
    except Exception as e:
        print(c, 'no line', e)
PY
