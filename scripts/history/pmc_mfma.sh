#!/bin/bash
# Matrix-pipe utilisation and effective clock of the bench kernels: ONE rocprofv3 --pmc pass (SQ + GRBM
# counters only, no other trace domain).  Usage: bash scripts/pmc_mfma.sh [tag]
TAG="${1:-pmc_mfma}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
R=$PWD
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    --kernel-trace --output-format csv -d "$R/$OUT/raw" -o pmc -- \
    python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$R/$OUT/run.log" 2>&1 ); echo "exit $?"
python - <<PY
import csv, glob, json, re
acc = {}
for path in glob.glob("$OUT/raw/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = re.sub(r'\(.*$', '', re.sub(r'^void ', '', row['Kernel_Name']))
        e = acc.setdefault(k, {}).setdefault(row['Counter_Name'], [0.0, set()])
        e[0] += float(row['Counter_Value']); e[1].add(row['Dispatch_Id'])
out = {}
for k, c in acc.items():
    out[k] = {n: v[0] / max(len(v[1]), 1) for n, v in c.items()}
    out[k]['launches'] = max(len(v[1]) for v in c.values())
json.dump(out, open("$OUT/pmc_mfma_summary.json", 'w'), indent=1, sort_keys=True)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0))[:8]:
    print(k[:58], {a: round(b) for a, b in v.items()})
PY
F=$(find "$OUT/raw" -name "*counter_collection.csv" | head -1); [ -n "$F" ] && cp "$F" "$OUT/counter_collection.csv"
T=$(find "$OUT/raw" -name "*kernel_trace.csv" | head -1); [ -n "$T" ] && python scripts/trace_stats.py "$T" 0.3 > "$OUT/kernel_stats_steady.csv"
rm -rf "$OUT/raw"
