#!/bin/bash
# One gpurun call: smoke, GPU parity tests, the bench line (with its `extra` block), and a rocprofv3 kernel-trace
# summary of the headline workload.  Usage (repo root on the GPU box): bash scripts/gpu_round.sh [tag] [what]
#   what: any of "smoke tests bench prof" (default all)
TAG="${1:-r02}"
WHAT="${2:-smoke tests bench prof}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
: > "$OUT/summary.txt"
if [[ "$WHAT" == *smoke* ]]; then
  echo "== smoke" | tee -a "$OUT/summary.txt"
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke exit $?" | tee -a "$OUT/summary.txt"
  tail -3 "$OUT/smoke.log" | tee -a "$OUT/summary.txt"
fi
if [[ "$WHAT" == *tests* ]]; then
  echo "== pytest -m gpu" | tee -a "$OUT/summary.txt"
  timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --tb=short --durations=15 ${PYTEST_ARGS} > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" | tee -a "$OUT/summary.txt"
  tail -60 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
fi
if [[ "$WHAT" == *bench* ]]; then
  echo "== bench" | tee -a "$OUT/summary.txt"
  SECONDS=0; timeout 1200 python bench.py --steps 10 --warmup 2 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench exit $?" | tee -a "$OUT/summary.txt"
  cat "$OUT/bench.json" | tee -a "$OUT/summary.txt"; grep -v "^\s" "$OUT/bench.err" | tail -5 | tee -a "$OUT/summary.txt"
  echo "bench wall seconds: $SECONDS" | tee -a "$OUT/summary.txt"
fi
if [[ "$WHAT" == *prof* ]]; then
  echo "== rocprofv3 kernel-trace stats" | tee -a "$OUT/summary.txt"
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-extra ${BENCH_ARGS} > "$OLDPWD/$OUT/prof_bench.log" 2>&1 ); echo "rocprof exit $?" | tee -a "$OUT/summary.txt"
  tail -2 "$OUT/prof_bench.log" | tee -a "$OUT/summary.txt"
  F=$(find "$OUT/prof" -name "*kernel_stats*.csv" | head -1)
  [ -n "$F" ] && head -25 "$F" | tee -a "$OUT/summary.txt"
  T=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
  [ -n "$T" ] && python scripts/trace_stats.py "$T" 0.3 > "$OUT/kernel_stats_steady.csv" && head -25 "$OUT/kernel_stats_steady.csv" | tee -a "$OUT/summary.txt"
  find "$OUT/prof" -name "*kernel_trace*" -size +20M -delete 2>/dev/null
fi
echo done
