#!/bin/bash
OUT=gpurun_out/r02l; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout 600 -p no:cacheprovider --tb=short -k "winograd" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest.log
RW_BATCH=64 RW_LAYERS=layer8,layer10,layer12,layer14,layer16,layer18 RW_ALGO=winograd RW_OUT=r02l/cb.json python scripts/conv_bench.py 2>&1 | grep layer
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d['value'], d['ms_per_step'])
for k,v in list(d['roofline']['per_kernel'].items())[:6]: print(k, v)
PY
