TAG=r04s; OUT=gpurun_out/$TAG; mkdir -p $OUT; R=$PWD; export TMPDIR=/tmp
i=0
for SET in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  ( cd /tmp && RW_LAYERS=layer16 timeout 120 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$R/$OUT/p$i" -o pmc -- \
      python "$R/scripts/dconv_bench.py" > "$R/$OUT/p$i.log" 2>&1 ); echo "pass $i exit $?"
done
python - <<PY
import csv, glob, json, re
acc = {}
for path in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = re.sub(r'\(.*$', '', re.sub(r'^void ', '', row['Kernel_Name']))
        if ('dconv' not in k and 'wino36h' not in k) or 'pack' in k: continue
        k = k + ' grid=' + row.get('Grid_Size', '?')
        e = acc.setdefault(k, {}).setdefault(row['Counter_Name'], [0.0, set()])
        e[0] += float(row['Counter_Value']); e[1].add((path, row['Dispatch_Id']))
out = {k: {n: v[0] / max(len(v[1]), 1) for n, v in c.items()} for k, c in acc.items()}
json.dump(out, open("$OUT/pmc_lds_summary.json", 'w'), indent=1, sort_keys=True)
for k, v in sorted(out.items()): print(k, json.dumps(v, sort_keys=True))
PY
rm -rf $OUT/p[0-9]
