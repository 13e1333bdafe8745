#!/bin/bash
# Round-3 measurement session on the GPU box (gpurun -- bash tools/r03_measure.sh [what ...]).  Bulky traces go to /tmp on the
# box; only the summaries come back through gpurun_out/ and are copied to profiles/ by hand afterwards.
# The profiled runs export GPU_MAX_HW_QUEUES=8 in the SHELL: rocprofv3's preloaded library starts the HIP runtime before
# bench.py runs, so the os.environ.setdefault at the top of bench.py would come too late (the runtime reads it once).
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/r03
mkdir -p $OUT /tmp/prof
WHAT=${*:-bench_cfg2 prof_headline prof_c1}
for w in $WHAT; do
  case $w in
    bench_cfg2)
      timeout 900 python bench.py > $OUT/bench_cfg2.log 2>&1; tail -1 $OUT/bench_cfg2.log > $OUT/bench_line_cfg2.json ;;
    bench_cfg3|bench_cfg4|bench_cfg5)
      c=${w#bench_}
      timeout 1200 python bench.py --config $c > $OUT/bench_$c.log 2>&1; tail -1 $OUT/bench_$c.log > $OUT/bench_line_$c.json ;;
    prof_headline)  # the configuration the headline runs: twelve attacks in flight as three stacks of four, eight hardware queues
      rm -rf /tmp/prof/hl
      GPU_MAX_HW_QUEUES=8 timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/hl -- \
        python3 bench.py --steps 12 --warmup 0 --no-cpu-baseline --no-single --no-f32 > $OUT/prof_headline.log 2>&1
      f=$(find /tmp/prof/hl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/cfg2_headline_kernel_stats.csv
      tail -1 $OUT/prof_headline.log > $OUT/prof_line_headline.json ;;
    prof_c4)  # four attacks in flight, one stream each (round 2's headline configuration; HITADV_STACK=1 switches the stacking off)
      rm -rf /tmp/prof/c4
      GPU_MAX_HW_QUEUES=8 HITADV_STACK=1 timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/c4 -- \
        python3 bench.py --steps 4 --warmup 0 --concurrent 4 --no-cpu-baseline --no-single --no-f32 > $OUT/prof_c4.log 2>&1
      f=$(find /tmp/prof/c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/cfg2_c4_kernel_stats.csv
      tail -1 $OUT/prof_c4.log > $OUT/prof_line_c4.json ;;
    prof_c1)
      rm -rf /tmp/prof/c1
      GPU_MAX_HW_QUEUES=8 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/c1 -- \
        python3 bench.py --steps 1 --warmup 0 --concurrent 1 --no-cpu-baseline --no-single --no-f32 > $OUT/prof_c1.log 2>&1
      f=$(find /tmp/prof/c1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/cfg2_c1_kernel_stats.csv
      tail -1 $OUT/prof_c1.log > $OUT/prof_line_c1.json ;;
    prof_cfg3|prof_cfg4|prof_cfg5)
      c=${w#prof_}
      rm -rf /tmp/prof/$c
      GPU_MAX_HW_QUEUES=8 timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/$c -- \
        python3 bench.py --config $c --steps 1 --warmup 0 --concurrent 1 --no-cpu-baseline > $OUT/prof_$c.log 2>&1
      f=$(find /tmp/prof/$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${c}_kernel_stats.csv
      tail -1 $OUT/prof_$c.log > $OUT/prof_line_$c.json ;;
    kbench)
      timeout 300 python tools/kbench.py > $OUT/kbench.json 2> $OUT/kbench.err ;;
    pmc_k1)  # HBM traffic of K1 / K2: separate --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share one)
      for ctr in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/prof/pmc_$ctr
        timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof/pmc_$ctr -- python3 tools/k1_probe.py > $OUT/pmc_$ctr.log 2>&1
        f=$(find /tmp/prof/pmc_$ctr -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/pmc_$ctr.csv
      done
      python tools/pmc_summary.py $OUT/pmc_WRITE_SIZE.csv $OUT/pmc_FETCH_SIZE.csv > $OUT/kbench_traffic.json 2>> $OUT/pmc_WRITE_SIZE.log
      rm -f $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv ;;
    tests)
      timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -15 $OUT/pytest_gpu.log
      cp gpurun_out/parity_report_gpu.json $OUT/parity_report.json; cp gpurun_out/parity_rows_gpu.json $OUT/parity_rows.json ;;
  esac
done
ls -la $OUT
