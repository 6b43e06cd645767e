#!/bin/bash
# SQ counters of the fused upsampling kernels (own --pmc passes, kernel-trace only): layer 17 in the persistent form,
# layer 13 in the one-workgroup-per-CU form.  usage: bash scripts/gpu_tconv_pmc.sh <tag>
TAG=${1:-r05pmc}; OUT=gpurun_out/$TAG; mkdir -p $OUT; R=$PWD; export TMPDIR=/tmp
export RW_BATCH=64 RW_TCONV_ONLY=1
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  for V in "0 layer17" "16 layer13"; do
    set -- $V
    ( cd /tmp && RW_TCONV_TY=$1 RW_LAYERS=$2 timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$R/$OUT/p${i}_$1" -o pmc -- \
        python "$R/scripts/tconv_bench.py" > "$R/$OUT/p${i}_$1.log" 2>&1 ); echo "pass $i form $1 exit $?"
  done
done
python - <<PY
import csv, glob, json, re
acc = {}
for path in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = re.sub(r'\(.*$', '', re.sub(r'^void ', '', row['Kernel_Name']))
        if 'tconv_blur' not in k: continue
        e = acc.setdefault(k, {}).setdefault(row['Counter_Name'], [0.0, set()])
        e[0] += float(row['Counter_Value']); e[1].add((path, row['Dispatch_Id']))
out = {k: {n: v[0] / max(len(v[1]), 1) for n, v in c.items()} for k, c in acc.items()}
json.dump(out, open("$OUT/pmc_tconv_summary.json", 'w'), indent=1, sort_keys=True)
for k, v in sorted(out.items()): print(k, json.dumps(v, sort_keys=True))
PY
rm -rf $OUT/p[1-3]_*[0-9]
