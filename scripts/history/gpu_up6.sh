#!/bin/bash
OUT=gpurun_out/up6; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "winograd_f4 or one_pass_upsampling" 2>&1 | tail -3 | tee $OUT/test.log
RW_BATCH=64 RW_ALGO=winograd4 RW_LAYERS=layer10,layer14,layer16,layer18 RW_OUT=up6/cb.json timeout 300 python scripts/conv_bench.py 2>&1 | grep layer | tee $OUT/cb.log
for i in 1 2; do
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench$i.json 2> $OUT/bench.err; echo "bench exit $?"
python - <<PY
import json
d=json.load(open('$OUT/bench$i.json')); print(d['value'], d['ms_per_step']); print(json.dumps({k:v for k,v in d['roofline']['per_kernel'].items() if 'wino36' in k}))
PY
done
