#!/bin/bash
# direct-16 kernels: where does the time go?  (1) batch-64 check against the fp32 direct kernel, (2) timing ablations
# (scripts/probe/abl/lib_dcabl_<bits>.so: 1 = no staging loads, 2 = no MFMAs, 4 = no weight loads, 8 = no epilogue),
# (3) SQ / TCC counters of the product.   bash scripts/gpu_dconv_abl.sh <tag>
TAG=${1:-r04p}; OUT=gpurun_out/$TAG; mkdir -p $OUT; R=$PWD; export TMPDIR=/tmp
export RW_LAYERS=${RW_LAYERS:-layer14,layer16,layer17,layer18}
echo "== product (with checks on layer16)"; RW_CHECK=1 python scripts/dconv_bench.py 2>&1 | grep layer | tee $OUT/product.jsonl
for a in $ABLS; do
  echo "== DC_ABL=$a"; RW_HIP_LIB=$R/scripts/probe/abl/lib_dcabl_$a.so python scripts/dconv_bench.py 2>&1 | grep layer | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['layer'], {k: v for k, v in d.items() if k.endswith('_ms')})" | tee $OUT/abl_$a.txt
done
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  [ -n "$NOPMC" ] && break
  ( cd /tmp && RW_LAYERS=layer16,layer17 timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$R/$OUT/p$i" -o pmc -- \
      python "$R/scripts/dconv_bench.py" > "$R/$OUT/p$i.log" 2>&1 ); echo "pass $i exit $?"
done
python - <<PY
import csv, glob, json, re
acc = {}
for path in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = re.sub(r'\(.*$', '', re.sub(r'^void ', '', row['Kernel_Name']))
        if 'dconv' not in k or 'pack' in k: continue
        k = k + ' grid=' + row.get('Grid_Size', '?')
        e = acc.setdefault(k, {}).setdefault(row['Counter_Name'], [0.0, set()])
        e[0] += float(row['Counter_Value']); e[1].add((path, row['Dispatch_Id']))
out = {k: {n: v[0] / max(len(v[1]), 1) for n, v in c.items()} for k, c in acc.items()}
json.dump(out, open("$OUT/pmc_dconv_summary.json", 'w'), indent=1, sort_keys=True)
for k, v in sorted(out.items()): print(k, json.dumps(v, sort_keys=True))
PY
rm -rf $OUT/p[0-9]
