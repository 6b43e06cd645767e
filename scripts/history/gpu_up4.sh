#!/bin/bash
# One-pass upsampling StyledConv (conv_up_wino36_kernel): parity test, per-layer time against the two-pass route, bench.
OUT=gpurun_out/up4; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "one_pass_upsampling" 2>&1 | tail -5 | tee $OUT/test.log
for M in fused wino+blur; do
  RW_BATCH=64 RW_UP_ALGO=$M RW_LAYERS=layer13,layer15,layer17 RW_OUT=up4/cb_$M.json timeout 300 python scripts/conv_bench.py 2>&1 | grep layer | tee $OUT/cb_$M.log
done
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
python - <<PY
import json
d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step']); print(json.dumps(d['roofline']['per_kernel']))
PY
RW_UP_FUSED=0 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench_off.json 2> $OUT/bench_off.err
python -c "
import json
d=json.load(open('$OUT/bench_off.json')); print('off', d['value'], d['ms_per_step'])"
