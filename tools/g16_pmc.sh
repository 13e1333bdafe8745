cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p /tmp/prof
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY"; do
  i=$((i+1)); rm -rf /tmp/prof/g$i
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/prof/g$i -- python3 tools/g16_pmc_probe.py > /tmp/prof/g$i.log 2>&1
  f=$(find /tmp/prof/g$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r['Kernel_Name']
    if 'gemm_f16x2' not in k: continue
    name='ring' if 'ring' in k else 'staged'
    agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name,d in agg.items():
    print(name, {c: round(sum(v[1:])/max(1,len(v[1:]))) for c,v in d.items()})
PY
done
