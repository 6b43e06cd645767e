#!/bin/bash
# Winograd kernel: timing ablations (tuning builds under scripts/probe/) + one PMC pass + the kernel tests
OUT=gpurun_out/r02e; mkdir -p $OUT
export RW_BATCH=64 RW_LAYERS=layer10,layer14,layer18 RW_ALGO=winograd
echo "== product"; RW_OUT=r02e/cb.json python scripts/conv_bench.py 2>&1 | grep layer
for a in 1 3 8 16 31; do echo "== abl $a"; RW_HIP_LIB=$PWD/scripts/probe/lib_wn_abl$a.so RW_OUT=r02e/cb_abl$a.json python scripts/conv_bench.py 2>&1 | grep layer; done
export TMPDIR=/tmp
R=$PWD
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU \
    --kernel-trace --output-format csv -d "$R/$OUT/raw" -o pmc -- python "$R/scripts/conv_bench.py" > "$R/$OUT/pmc_run.log" 2>&1 ); echo "pmc exit $?"
python - <<PY
import csv, glob, json, re
acc = {}
for path in glob.glob("$OUT/raw/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = re.sub(r'\(.*$', '', re.sub(r'^void ', '', row['Kernel_Name']))
        if 'wino' not in k: continue
        k += ' grid=%s' % row.get('Grid_Size', '')
        e = acc.setdefault(k, {}).setdefault(row['Counter_Name'], [0.0, set()])
        e[0] += float(row['Counter_Value']); e[1].add(row['Dispatch_Id'])
out = {}
for k, c in acc.items():
    out[k] = {n: v[0] / max(len(v[1]), 1) for n, v in c.items()}
json.dump(out, open("$OUT/pmc_wino_summary.json", 'w'), indent=1, sort_keys=True)
for k, v in out.items(): print(k, {a: round(b) for a, b in v.items()})
PY
rm -rf "$OUT/raw"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout 600 -p no:cacheprovider --tb=short -k "winograd" > $OUT/pytest_wino.log 2>&1; echo "pytest exit $?"; tail -15 $OUT/pytest_wino.log
