#!/bin/bash
# micro-batch probe + the GPU tests touched this session
OUT=gpurun_out/${1:-micro}
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --timeout 600 -p no:cacheprovider --tb=short -k "key_methods or fast_mconv or micro" > "$OUT/pytest.log" 2>&1; echo "pytest exit $?"
tail -5 "$OUT/pytest.log"
RW_OUT=${1:-micro}/micro_probe.json timeout 900 python scripts/micro_probe.py > "$OUT/micro.log" 2>&1; echo "probe exit $?"
grep "spec" "$OUT/micro.log"
tail -3 "$OUT/micro.log"
