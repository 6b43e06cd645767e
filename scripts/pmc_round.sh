#!/bin/bash
# HBM traffic of the bench kernels: one rocprofv3 --pmc pass per counter (never combined with
# other trace domains), summarised by scripts/pmc_summary.py.  Usage: bash scripts/pmc_round.sh [tag]
TAG="${1:-pmc}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
R=$PWD
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$R/$OUT/$C" -o pmc -- \
      python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$R/$OUT/$C.log" 2>&1 ); echo "$C exit $?"
done
python scripts/pmc_summary.py "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE" > "$OUT/pmc_summary.json"
python - <<PY
import json
d = json.load(open("$OUT/pmc_summary.json"))
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get('hbm_bytes_per_launch_raw', 0))[:8]:
    print(k[:60], {a: round(b) for a, b in v.items() if 'per_launch' in a})
PY
# the per-dispatch CSVs are large; keep the summary and one small CSV per counter
for C in FETCH_SIZE WRITE_SIZE; do
  F=$(find "$OUT/$C" -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && cp "$F" "$OUT/${C}_counter_collection.csv"
  rm -rf "$OUT/$C"
done
