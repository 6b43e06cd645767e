#!/bin/bash
# One gpurun call: smoke, GPU parity tests, the bench line, and a rocprofv3 kernel-trace summary.
# Usage (from the repo root on the GPU box): bash scripts/gpu_round.sh [tag]
TAG="${1:-r01}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== smoke" | tee "$OUT/summary.txt"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke exit $?" | tee -a "$OUT/summary.txt"
tail -3 "$OUT/smoke.log" | tee -a "$OUT/summary.txt"
echo "== pytest -m gpu" | tee -a "$OUT/summary.txt"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --tb=short > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" | tee -a "$OUT/summary.txt"
tail -40 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
echo "== bench" | tee -a "$OUT/summary.txt"
timeout 900 python bench.py --steps 10 --warmup 2 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench exit $?" | tee -a "$OUT/summary.txt"
cat "$OUT/bench.json" | tee -a "$OUT/summary.txt"; tail -5 "$OUT/bench.err" | tee -a "$OUT/summary.txt"
for wl in ffhq256 edit; do
  timeout 900 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_$wl.json" 2> "$OUT/bench_$wl.err"
  echo "bench $wl exit $?" | tee -a "$OUT/summary.txt"; cat "$OUT/bench_$wl.json" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/bench_$wl.err" | tee -a "$OUT/summary.txt"
done
echo "== rocprofv3 kernel-trace stats" | tee -a "$OUT/summary.txt"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$OLDPWD/$OUT/prof_bench.log" 2>&1 ); echo "rocprof exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof" -name "*.csv" | head -8 | tee -a "$OUT/summary.txt"
F=$(find "$OUT/prof" -name "*kernel_stats*.csv" | head -1)
[ -n "$F" ] && head -25 "$F" | tee -a "$OUT/summary.txt"
T=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python scripts/trace_stats.py "$T" 0.3 > "$OUT/kernel_stats_steady.csv"   # warm-up launches left out
# keep only the small summaries (traces can be large)
find "$OUT/prof" -name "*kernel_trace*" -size +20M -delete 2>/dev/null
echo done
