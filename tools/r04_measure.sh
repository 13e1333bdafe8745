#!/bin/bash
# Round-4 measurement session on the GPU box (gpurun -- bash tools/r04_measure.sh [what ...]).  Bulky traces go to /tmp on the
# box; only the summaries come back through gpurun_out/ and are copied to profiles/ by hand afterwards.
# The profiled runs export GPU_MAX_HW_QUEUES=8 in the SHELL: rocprofv3's preloaded library starts the HIP runtime before
# bench.py runs, so the os.environ.setdefault at the top of bench.py would come too late (the runtime reads it once).
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/r04
mkdir -p $OUT /tmp/prof
WHAT=${*:-bench_cfg2 prof_headline prof_c1}
for w in $WHAT; do
  case $w in
    bench_cfg2)
      timeout 900 python bench.py > $OUT/bench_cfg2.log 2>&1; tail -1 $OUT/bench_cfg2.log > $OUT/bench_line_cfg2.json ;;
    bench_cfg3|bench_cfg4|bench_cfg5)
      c=${w#bench_}
      timeout 1200 python bench.py --config $c > $OUT/bench_$c.log 2>&1; tail -1 $OUT/bench_$c.log > $OUT/bench_line_$c.json ;;
    prof_headline)  # the configuration the headline runs: twenty-four attacks in flight as three stacks of eight, eight hardware queues
      rm -rf /tmp/prof/hl
      GPU_MAX_HW_QUEUES=8 timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/hl -- \
        python3 bench.py --steps 24 --warmup 0 --no-cpu-baseline --no-single --no-f32 --no-other-configs > $OUT/prof_headline.log 2>&1
      f=$(find /tmp/prof/hl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/cfg2_headline_kernel_stats.csv
      tail -1 $OUT/prof_headline.log > $OUT/prof_line_headline.json ;;
    prof_c4)  # four attacks in flight, one stream each (round 2's headline configuration; HITADV_STACK=1 switches the stacking off)
      rm -rf /tmp/prof/c4
      GPU_MAX_HW_QUEUES=8 HITADV_STACK=1 timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/c4 -- \
        python3 bench.py --steps 4 --warmup 0 --concurrent 4 --no-cpu-baseline --no-single --no-f32 --no-other-configs > $OUT/prof_c4.log 2>&1
      f=$(find /tmp/prof/c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/cfg2_c4_kernel_stats.csv
      tail -1 $OUT/prof_c4.log > $OUT/prof_line_c4.json ;;
    prof_c1)
      rm -rf /tmp/prof/c1
      GPU_MAX_HW_QUEUES=8 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/c1 -- \
        python3 bench.py --steps 1 --warmup 0 --concurrent 1 --no-cpu-baseline --no-single --no-f32 --no-other-configs > $OUT/prof_c1.log 2>&1
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
    pmc_gar)  # HBM traffic of cfg4's / cfg5's roofline kernel (group_add_relu_fwd_k at the two shapes), merged into kbench_traffic.json
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
try:
    merged = json.load(open(out + '/kbench_traffic.json'))
except Exception:
    merged = {}
for c in ('cfg4', 'cfg5'):
    d = json.load(open('%s/gar_traffic_%s.json' % (out, c)))
    for k, v in d.items():
        if 'group_add_relu_fwd_k' in k:
            merged[k + '@' + c] = v
json.dump(merged, open(out + '/kbench_traffic.json', 'w'), indent=1)
print(json.dumps({k: v['hbm_bytes_per_launch'] for k, v in merged.items()}))
PY
      ;;
    pmc_mfma)  # SQ counters of V1 and G16, one pass per counter group (a pass fails as a whole on an unknown counter name)
      i=0
      for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
                 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
                 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16"; do
        i=$((i+1)); rm -rf /tmp/prof/pm$i
        timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/prof/pm$i -- python3 tools/mfma_pmc_probe.py > $OUT/pmc_mfma_$i.log 2>&1
        f=$(find /tmp/prof/pm$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" /tmp/prof/pmc_mfma_$i.csv
      done
      python - > $OUT/mfma_pmc.txt <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('/tmp/prof/pmc_mfma_*.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        per[(r['Kernel_Name'][:70], r['Dispatch_Id'])][r['Counter_Name']] += float(r['Counter_Value'])
    for (k, d), cs in per.items():
        for c, v in cs.items():
            acc[(k, cs.get('Grid_Size', 0))][c].append(v)
    for (k, _), cs in acc.items():
        if 'linear_max_fwd_bf3' in k or 'gemm_f16x2' in k:
            print(f.split('_')[-1][:-4], k, {c: round(sum(v) / len(v)) for c, v in cs.items()}, 'launches', len(next(iter(cs.values()))))
PY
      ;;
    tests)
      timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -15 $OUT/pytest_gpu.log
      cp gpurun_out/parity_report_gpu.json $OUT/parity_report.json; cp gpurun_out/parity_rows_gpu.json $OUT/parity_rows.json ;;
  esac
done
ls -la $OUT
