set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
run() { HITADV_STACK=$2 timeout 300 python bench.py --concurrent $1 --steps $3 --warmup 0 --no-cpu-baseline --no-single --no-f32 --no-other-configs > gpurun_out/r04/tail_$1_$2_$3.log 2>&1; python - gpurun_out/r04/tail_$1_$2_$3.log "in_flight=$1 per_stack=$2 steps=$3" <<'PY'
import json, sys
d = json.loads([x for x in open(sys.argv[1]) if x.startswith('{"metric"')][-1]); print(sys.argv[2], '->', round(d['value'], 2), 'clouds/s')
PY
}
run 8 4 8; run 8 3 8; run 8 2 8; run 12 4 20; run 10 4 20; run 20 4 20
