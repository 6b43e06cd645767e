#!/bin/bash
# Round 5 measurement call: A/B of the forward's kernel choices on ONE box (same process conditions), rocprofv3
# kernel-trace stats of the default and of the two-pass layer 17, HBM counter passes (one --pmc pass per counter).
# Usage (repo root on the GPU box): bash scripts/gpu_r05_measure.sh <tag> ["ab prof pmc"]
TAG="${1:-r05}"
WHAT="${2:-ab prof pmc}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
R=$PWD
B="--steps 10 --warmup 2 --no-extra --no-cpu-baseline"
if [[ "$WHAT" == *ab* ]]; then
  : > "$OUT/ab.jsonl"
  for V in "" "RW_UP_FUSED=0" "RW_MM_DIRECT16=auto" "RW_MM_DIRECT16=auto RW_UP_FUSED=0" "RW_MM_DIRECT16=conv RW_UP_FUSED=0" "RW_MM=f32" ${AB_EXTRA}; do
    echo "== $V"
    L=$(env $V python bench.py $B 2>"$OUT/ab.err" | tail -1)
    python - "$V" "$L" >> "$OUT/ab.jsonl" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(json.dumps(dict(env=sys.argv[1], images_per_s=d['value'], ms_per_step=d['ms_per_step'], parity=d['parity']['linf'],
                      per_kernel={k: (v.get('launches'), v.get('ms')) for k, v in d['roofline'].get('per_kernel', {}).items()})))
PY
    tail -1 "$OUT/ab.jsonl" | cut -c1-400
  done
fi
prof() {   # name, env...
  local name=$1; shift
  ( cd /tmp && env "$@" timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/prof_$name" -o bench -- \
      python "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-extra > "$R/$OUT/prof_$name.log" 2>&1 ); echo "rocprof $name exit $?"
  local T=$(find "$OUT/prof_$name" -name "*kernel_trace.csv" | head -1)
  [ -n "$T" ] && python scripts/trace_stats.py "$T" 0.3 > "$OUT/${name}_kernel_stats_steady.csv" && head -16 "$OUT/${name}_kernel_stats_steady.csv"
  local F=$(find "$OUT/prof_$name" -name "*kernel_stats*.csv" | head -1)
  [ -n "$F" ] && cp "$F" "$OUT/${name}_kernel_stats.csv"
  rm -rf "$OUT/prof_$name"
}
if [[ "$WHAT" == *prof* ]]; then
  prof default RW_NOP=1
  prof twopass17 RW_UP_FUSED=0
fi
if [[ "$WHAT" == *pmc* ]]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && env ${PMC_ENV} timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$R/$OUT/$C" -o pmc -- \
        python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extra > "$R/$OUT/$C.log" 2>&1 ); echo "$C exit $?"
  done
  python scripts/pmc_summary.py "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE" > "$OUT/pmc_summary.json"
  python - <<PY
import json
d = json.load(open("$OUT/pmc_summary.json"))
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get('hbm_bytes_per_launch_raw', 0))[:14]:
    print(k[:60], {a: round(b) for a, b in v.items() if 'per_launch' in a or 'step' in a})
PY
  rm -rf "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE"
fi
echo done
