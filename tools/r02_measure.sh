#!/bin/bash
# Round-2 measurement session on the GPU box (gpurun -- bash tools/r02_measure.sh).  Bulky traces go to /tmp on the box;
# only the summaries come back through gpurun_out/ and are copied to profiles/ by hand afterwards.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/r02
mkdir -p $OUT /tmp/prof
# 1. bench lines (default flags = what the driver runs, plus the other configs)
timeout 600 python bench.py > $OUT/bench_cfg2.log 2>&1; tail -1 $OUT/bench_cfg2.log > $OUT/bench_line_cfg2.json
for c in cfg3 cfg4 cfg5; do timeout 900 python bench.py --config $c > $OUT/bench_$c.log 2>&1; tail -1 $OUT/bench_$c.log > $OUT/bench_line_$c.json; done
# 2. kernel timings
timeout 300 python tools/kbench.py > $OUT/kbench.json 2> /dev/null
# 3. rocprofv3 kernel statistics of the bench command itself (one attack, one step), per configuration
for c in cfg2 cfg3 cfg4 cfg5; do
  rm -rf /tmp/prof/$c
  timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/$c -- python3 bench.py --config $c --steps 1 --warmup 0 --concurrent 1 --no-cpu-baseline --no-single > $OUT/prof_$c.log 2>&1
  f=$(find /tmp/prof/$c -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/${c}_kernel_stats.csv
  tail -1 $OUT/prof_$c.log > $OUT/prof_line_$c.json
done
# 4. HBM traffic of K1 / K2: separate --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share one)
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof/pmc_$ctr
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof/pmc_$ctr -- python3 tools/k1_probe.py > $OUT/pmc_$ctr.log 2>&1
  f=$(find /tmp/prof/pmc_$ctr -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/pmc_$ctr.csv
done
python tools/pmc_summary.py $OUT/pmc_WRITE_SIZE.csv $OUT/pmc_FETCH_SIZE.csv > $OUT/kbench_traffic.json 2>> $OUT/pmc_WRITE_SIZE.log
rm -f $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv  # raw per-dispatch rows stay on the box; the summary is what is kept
ls -la $OUT
