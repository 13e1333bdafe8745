#!/bin/bash
# gpurun -- bash tools/r04_probe.sh : V3 phase stamps at B = 32 / 128, and how much of cfg5's kernels overlap
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/r04; mkdir -p $OUT /tmp/prof
python tools/v3_phases.py 32 > $OUT/v3_phases_B32.json 2>&1
python tools/v3_phases.py 128 > $OUT/v3_phases_B128.json 2>&1
for mode in "" "--sequential-sweep"; do
  rm -rf /tmp/prof/c5
  GPU_MAX_HW_QUEUES=8 timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof/c5 -- python3 tools/cfg5_short.py $mode > $OUT/cfg5_short$mode.log 2>&1
  tail -1 $OUT/cfg5_short$mode.log
  python tools/trace_overlap.py /tmp/prof/c5 > $OUT/cfg5_overlap$mode.json 2>&1
  head -8 $OUT/cfg5_overlap$mode.json
done
cat $OUT/v3_phases_B128.json | head -60
