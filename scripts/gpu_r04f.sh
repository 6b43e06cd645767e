#!/bin/bash
TAG=${1:-r04f}; OUT=gpurun_out/$TAG; mkdir -p $OUT; R=$PWD; export TMPDIR=/tmp
export RW_BATCH=64 RW_UP_ALGO=wino RW_LAYERS=layer13 RW_UPW_MM=split
echo "== up product split"; python scripts/conv_bench.py 2>&1 | grep layer | cut -c1-22,100-190
for a in 6 16 32 64 48 112 118; do
  echo "== up UW_ABL=$a split"; RW_HIP_LIB=$R/scripts/probe/abl/lib_uwabl_$a.so python scripts/conv_bench.py 2>&1 | grep layer | cut -c1-22,100-190
done
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  for mm in f32 split; do
    ( cd /tmp && RW_UPW_MM=$mm RW_OUT=$TAG/pmc_run.json timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$R/$OUT/p${i}_$mm" -o pmc -- \
        python "$R/scripts/conv_bench.py" > "$R/$OUT/p${i}_$mm.log" 2>&1 ); echo "pass $i $mm exit $?"
  done
done
python - <<PY
import csv, glob, json, re
acc = {}
for path in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = re.sub(r'\(.*$', '', re.sub(r'^void ', '', row['Kernel_Name']))
        if 'conv_up_wino' not in k: continue
        e = acc.setdefault(k, {}).setdefault(row['Counter_Name'], [0.0, set()])
        e[0] += float(row['Counter_Value']); e[1].add((path, row['Dispatch_Id']))
out = {k: {n: v[0] / max(len(v[1]), 1) for n, v in c.items()} for k, c in acc.items()}
json.dump(out, open("$OUT/pmc_upw_summary.json", 'w'), indent=1, sort_keys=True)
for k, v in sorted(out.items()): print(k, json.dumps(v, sort_keys=True))
PY
rm -rf $OUT/p[0-9]_*
