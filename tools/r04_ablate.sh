#!/bin/bash
# What each kernel family costs the cfg2 LOOP: the headline job with the family's launches removed (diagnostic library,
# make -C hit_adv_amd/csrc ablate; results of such runs are garbage, only the clock is read).  gpurun -- bash tools/r04_ablate.sh
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04
mkdir -p $OUT
export HITADV_LIBRARY=$PWD/tools/build/libhitadv_hip_ablate.so
: > $OUT/ablate.txt
for what in none fc v2 v3 v1 v2,v3 fc,v2,v3 fc,v1,v2,v3; do
  HITADV_ABLATE=$what timeout 300 python bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-single --no-f32 --no-other-configs \
    > $OUT/ablate_$what.log 2>&1
  python - "$what" $OUT/ablate_$what.log >> $OUT/ablate.txt <<'PY'
import json, sys
lines = [l for l in open(sys.argv[2]) if l.startswith('{"metric"')]
if not lines:
    print("ablate=%s -> no line" % sys.argv[1])
else:
    d = json.loads(lines[-1])
    print("ablate=%-14s -> %.2f clouds/s  %.1f us per attack-iteration" % (sys.argv[1], d['value'], d['end_to_end']['us_per_iteration']))
PY
done
cat $OUT/ablate.txt
