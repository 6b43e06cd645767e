#!/bin/bash
# Round 4: where does the time of the split-operand F(4x4,3x3) kernels go?  (1) timing ablations (libraries built with
# -DW4_ABL=<bits> under scripts/probe/abl/lib_w4habl_<bits>.so, results WRONG), (2) SQ counters of the product.
# bash scripts/gpu_w4h_abl.sh <tag>
TAG=${1:-r04c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; R=$PWD; export TMPDIR=/tmp
export RW_BATCH=64 RW_ALGO=winograd4 RW_LAYERS=${RW_LAYERS:-layer10,layer14,layer18} RW_W4_MM=split
for ps in 0 1; do
  export RW_W4H_PS=$ps
  echo "== product ps=$ps"; RW_OUT=$TAG/abl_product_ps$ps.json python scripts/conv_bench.py 2>&1 | grep layer | cut -c1-22,120-175
  for a in $ABLS; do
    echo "== W4_ABL=$a ps=$ps"; RW_HIP_LIB=$R/scripts/probe/abl/lib_w4habl_$a.so RW_OUT=$TAG/abl_${a}_ps$ps.json python scripts/conv_bench.py 2>&1 | grep layer | cut -c1-22,120-175
  done
done
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  for ps in 0 1; do
    ( cd /tmp && RW_W4H_PS=$ps RW_LAYERS=layer10,layer18 RW_OUT=$TAG/pmc_run.json timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$R/$OUT/p${i}_$ps" -o pmc -- \
        python "$R/scripts/conv_bench.py" > "$R/$OUT/p${i}_$ps.log" 2>&1 ); echo "pass $i ps $ps exit $?"
  done
done
python - <<PY
import csv, glob, json, re
acc = {}
for path in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = re.sub(r'\(.*$', '', re.sub(r'^void ', '', row['Kernel_Name']))
        if 'wino36' not in k: continue
        k = k + ' grid=' + row.get('Grid_Size', '?')
        e = acc.setdefault(k, {}).setdefault(row['Counter_Name'], [0.0, set()])
        e[0] += float(row['Counter_Value']); e[1].add((path, row['Dispatch_Id']))
out = {k: {n: v[0] / max(len(v[1]), 1) for n, v in c.items()} for k, c in acc.items()}
json.dump(out, open("$OUT/pmc_w4h_summary.json", 'w'), indent=1, sort_keys=True)
for k, v in sorted(out.items()): print(k, json.dumps(v, sort_keys=True))
PY
rm -rf $OUT/p[0-9]_[01]
