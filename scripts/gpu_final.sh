#!/bin/bash
# End-of-round measurement: full GPU suite, the bench line, rocprofv3 kernel-trace stats, PMC (HBM traffic, MFMA busy),
# per-layer conv bench, watermark job.   bash scripts/gpu_final.sh <tag>
TAG="${1:-r02n}"
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
bash scripts/gpu_round.sh $TAG "smoke tests bench prof" > $OUT/round.log 2>&1
grep -n "passed\|failed\|smoke ok\|bench exit\|rocprof exit" $OUT/summary.txt
# HBM traffic: one pass per counter
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$R/$OUT/$C" -o pmc -- \
      python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extra > "$R/$OUT/$C.log" 2>&1 ); echo "$C exit $?"
done
python scripts/pmc_summary.py "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE" > "$OUT/pmc_summary.json"
for C in FETCH_SIZE WRITE_SIZE; do rm -rf "$OUT/$C"; done
# matrix-pipe counters
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU \
    --kernel-trace --output-format csv -d "$R/$OUT/mfma" -o pmc -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extra > "$R/$OUT/mfma.log" 2>&1 ); echo "mfma pmc exit $?"
python - <<PY
import csv, glob, json, re
acc = {}
for path in glob.glob("$OUT/mfma/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = re.sub(r'\(.*$', '', re.sub(r'^void ', '', row['Kernel_Name']))
        e = acc.setdefault(k, {}).setdefault(row['Counter_Name'], [0.0, set()])
        e[0] += float(row['Counter_Value']); e[1].add(row['Dispatch_Id'])
out = {}
for k, c in acc.items():
    out[k] = {n: v[0] / max(len(v[1]), 1) for n, v in c.items()}
    out[k]['launches'] = max(len(v[1]) for v in c.values())
json.dump(out, open("$OUT/pmc_mfma_summary.json", 'w'), indent=1, sort_keys=True)
PY
rm -rf "$OUT/mfma"
RW_BATCH=64 RW_OUT=$TAG/cb_all.json python scripts/conv_bench.py 2>&1 | grep layer > $OUT/cb_all.log
timeout 900 python bench.py --workload watermark --steps 1 --warmup 1 > $OUT/watermark.json 2> $OUT/watermark.err; echo "watermark exit $?"
timeout 300 python bench.py --workload edit --steps 3 --warmup 1 > $OUT/edit.json 2> $OUT/edit.err; echo "edit exit $?"
echo done
python scripts/solve_probe.py > "$OUT/solve_probe.log" 2>&1; grep out_ch "$OUT/solve_probe.log"; cp gpurun_out/solve_probe.json "$OUT/" 2>/dev/null
RW_OUT=$TAG/micro_probe.json RW_SPECS=0,8:256,4:512,2:1024 python scripts/micro_probe.py > "$OUT/micro.log" 2>&1; grep spec "$OUT/micro.log"
bash scripts/gpu_sweep_prof.sh > "$OUT/sweep_prof.log" 2>&1; mkdir -p "$OUT/sweep_prof"; cp gpurun_out/sweep_prof/prof/sweep_kernel_stats.csv gpurun_out/sweep_prof/prof_edit/edit_kernel_stats.csv gpurun_out/sweep_prof/sweep.json "$OUT/sweep_prof/" 2>/dev/null
echo final done
