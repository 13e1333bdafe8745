#!/bin/bash
# gpurun -- bash tools/r05_measure.sh <what> ... : this round's measurements; outputs under gpurun_out/r05/
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; OUT=gpurun_out/r05; mkdir -p $OUT /tmp/prof
export GPU_MAX_HW_QUEUES=8
for what in "$@"; do
  case $what in
    pmc_loop)  # counters of cfg2's loop kernels at the stacked launch size: HBM bytes (separate passes) + SQ groups
      i=0
      for grp in "WRITE_SIZE" "FETCH_SIZE" \
                 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
                 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
                 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
        i=$((i+1)); rm -rf /tmp/prof/lp$i
        timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/prof/lp$i -- python3 tools/loop_pmc_probe.py > $OUT/pmc_loop_$i.log 2>&1
        f=$(find /tmp/prof/lp$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" /tmp/prof/pmc_loop_$i.csv
        tail -2 $OUT/pmc_loop_$i.log
      done
      python tools/loop_pmc_summary.py $OUT/loop_traffic.json /tmp/prof/pmc_loop_*.csv ;;
    tests)
      timeout 1700 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -15 $OUT/pytest_gpu.log
      cp gpurun_out/parity_report_gpu.json $OUT/parity_report.json 2>/dev/null; cp gpurun_out/parity_rows_gpu.json $OUT/parity_rows.json 2>/dev/null ;;
    bench)
      timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench_line.json ;;
    bench2)  # the N-rank path on the one GPU a lease has (diagnostic)
      timeout 600 python3 bench.py --gpus 2 --ranks-share-gpu --steps 4 --warmup 1 --no-single --no-other-configs --no-cpu-baseline > $OUT/bench_two_ranks_one_gpu.json 2> $OUT/bench2.err
      tail -c 1200 $OUT/bench_two_ranks_one_gpu.json; tail -5 $OUT/bench2.err ;;
    prof)  # rocprofv3 kernel stats of the headline command
      rm -rf /tmp/prof/cfg2
      timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/cfg2 -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-single --no-f32 --no-other-configs --no-cpu-baseline > $OUT/prof_cfg2.log 2>&1
      f=$(find /tmp/prof/cfg2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/cfg2_headline_kernel_stats.csv
      grep -m1 '^{"metric"' $OUT/prof_cfg2.log > $OUT/prof_line_headline.json; head -12 $OUT/cfg2_headline_kernel_stats.csv ;;
    probes)  # the FPS step's parts, G16's ablations and counters
      timeout 100 ./tools/tune/fps_step_probe > $OUT/fps_step_probe.txt 2>&1; tail -3 $OUT/fps_step_probe.txt
      timeout 100 ./tools/tune/clock_probe > $OUT/clock_probe.txt 2>&1
      timeout 600 python tools/tune/g16_ablate.py 2>/dev/null | tail -1 > $OUT/g16_ablation.json; cat $OUT/g16_ablation.json
      bash tools/g16_pmc.sh > $OUT/g16_pmc.txt 2>&1; cat $OUT/g16_pmc.txt ;;
    pmc_k1)  # HBM traffic of K1 / K2 (and cfg4's / cfg5's roofline kernel): separate --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share one)
      for ctr in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/prof/pmc_$ctr
        timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof/pmc_$ctr -- python3 tools/k1_probe.py > $OUT/pmc_$ctr.log 2>&1
        f=$(find /tmp/prof/pmc_$ctr -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" /tmp/prof/pmc_$ctr.csv
      done
      python tools/pmc_summary.py /tmp/prof/pmc_WRITE_SIZE.csv /tmp/prof/pmc_FETCH_SIZE.csv > $OUT/kbench_traffic.json 2>> $OUT/pmc_WRITE_SIZE.log
      for c in cfg4 cfg5; do
        for ctr in FETCH_SIZE WRITE_SIZE; do
          rm -rf /tmp/prof/gar_$ctr
          timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof/gar_$ctr -- python3 tools/gar_probe.py $c > $OUT/pmc_gar_${c}_$ctr.log 2>&1
          f=$(find /tmp/prof/gar_$ctr -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" /tmp/prof/gar_${c}_$ctr.csv
        done
        python tools/pmc_summary.py /tmp/prof/gar_${c}_WRITE_SIZE.csv /tmp/prof/gar_${c}_FETCH_SIZE.csv > $OUT/gar_traffic_$c.json
      done
      python - $OUT <<'PY'
import json, sys
out = sys.argv[1]
merged = json.load(open(out + '/kbench_traffic.json'))
for c in ('cfg4', 'cfg5'):
    d = json.load(open('%s/gar_traffic_%s.json' % (out, c)))
    for k, v in d.items():
        if 'group_add_relu_fwd_k' in k or 'rows_linear_gather' in k:
            merged[k + '@' + c] = v
json.dump(merged, open(out + '/kbench_traffic.json', 'w'), indent=1)
print(json.dumps({k: v['hbm_bytes_per_launch'] for k, v in merged.items()}))
PY
      ;;
    kbench)
      timeout 300 python tools/kbench.py > $OUT/kbench.json 2> $OUT/kbench.err; tail -c 600 $OUT/kbench.json ;;
  esac
done
ls -la $OUT
