#!/bin/bash
# gpurun -- bash tools/r04_probe2.sh : cfg5's sweep after the memcpy nodes left the captured loops
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/r04; mkdir -p $OUT /tmp/prof
HITADV_CW_TIMELINE=1 python tools/cfg5_short.py 2>&1 | grep -v amdgpu | tail -8
rm -rf /tmp/prof/c5
GPU_MAX_HW_QUEUES=8 timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof/c5 -- python3 tools/cfg5_short.py > $OUT/cfg5_short2.log 2>&1
tail -1 $OUT/cfg5_short2.log
python tools/trace_overlap.py /tmp/prof/c5 > $OUT/cfg5_overlap2.json 2>&1
head -8 $OUT/cfg5_overlap2.json; grep -c copyBuffer $OUT/cfg5_overlap2.json; grep -A3 copyBuffer $OUT/cfg5_overlap2.json | head -8
for m in "" "--sequential-sweep"; do timeout 400 python bench.py --config cfg5 --no-cpu-baseline $m > $OUT/bench_cfg5c$m.log 2>&1; tail -c 400 $OUT/bench_cfg5c$m.log | cut -c1-200; done
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_attack.py -m gpu -q -k "in_flight_at_once or cw_ or cfg5 or g7 or g9 or g13 or g14 or g15 or g19 or g20 or g24" --timeout=900 > $OUT/pytest_cw2.log 2>&1; tail -8 $OUT/pytest_cw2.log
