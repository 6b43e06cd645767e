#!/bin/bash
OUT=gpurun_out/r02m; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout 600 -p no:cacheprovider --tb=short -k "transposed" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest.log
RW_BATCH=64 RW_LAYERS=layer9,layer11,layer13,layer15,layer17 RW_OUT=r02m/cb_up.json python scripts/conv_bench.py 2>&1 | grep layer
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d['value'], d['ms_per_step'])
for k,v in list(d['roofline']['per_kernel'].items())[:6]: print(k, v)
PY
