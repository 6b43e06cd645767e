#!/bin/bash
# LDS counters of the forward's kernels (own --pmc passes, kernel-trace only): is the F(4x4,3x3) loop held by the LDS?
TAG="${1:-r03_lds}"; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
i=0
for SET in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_CMD_FIFO_FULL" \
           "SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$R/$OUT/p$i" -o pmc -- \
      python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extra > "$R/$OUT/p$i.log" 2>&1 ); echo "pass $i exit $?"
done
python - <<PY
import csv, glob, json, re
acc = {}
for path in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = re.sub(r'\(.*$', '', re.sub(r'^void ', '', row['Kernel_Name']))
        e = acc.setdefault(k, {}).setdefault(row['Counter_Name'], [0.0, set()])
        e[0] += float(row['Counter_Value']); e[1].add(row['Dispatch_Id'])
out = {}
for k, c in acc.items():
    out[k] = {n: v[0] / max(len(v[1]), 1) for n, v in c.items()}
json.dump(out, open("$OUT/pmc_lds_summary.json", 'w'), indent=1, sort_keys=True)
for k in ('conv_wino36b_ns_kernel', 'conv_up_wino_kernel', 'conv_up_wino36_kernel', 'conv_wino36_rgb_ns_kernel', 'conv_wino16_kernel<2, 2, 8, false>'):
    if k in out: print(k, json.dumps(out[k]))
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
