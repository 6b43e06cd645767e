#!/bin/bash
# direct-16 kernels: parity tests, then timings.  Usage: bash scripts/gpu_dconv.sh [tag] [what: "tests bench"]
TAG="${1:-dconv}"; WHAT="${2:-tests bench}"
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
if [[ "$WHAT" == *tests* ]]; then
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "direct16 or abi" -p no:cacheprovider --tb=short ${PYTEST_ARGS} > "$OUT/pytest.log" 2>&1
  echo "pytest exit $?"; tail -25 "$OUT/pytest.log"
fi
if [[ "$WHAT" == *bench* ]]; then
  timeout 600 python scripts/dconv_bench.py > "$OUT/bench.jsonl" 2> "$OUT/bench.err"; echo "bench exit $?"
  cat "$OUT/bench.jsonl"; tail -5 "$OUT/bench.err"
fi
