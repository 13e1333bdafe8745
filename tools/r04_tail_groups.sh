set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
run() { HITADV_STACK=$2 timeout 300 python bench.py --concurrent $1 --steps $3 --warmup 0 --no-cpu-baseline --no-single --no-f32 --no-other-configs > gpurun_out/r04/tail_$1_$2_$3.log 2>&1; python - gpurun_out/r04/tail_$1_$2_$3.log "in_flight=$1 per_stack=$2 steps=$3" <<'PY'
import json, sys
d = json.loads([x for x in open(sys.argv[1]) if x.startswith('{"metric"')][-1]); print(sys.argv[2], '->', round(d['value'], 2), 'clouds/s', d['config'].get('attacks_per_stack'))
PY
}
run 21 7 21; run 27 9 27; run 30 10 30; run 36 12 36; run 24 8 24; run 20 8 20
