#!/bin/bash
OUT=gpurun_out/up5; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "winograd_f4 or one_pass_upsampling" 2>&1 | tail -5 | tee $OUT/test.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
python - <<PY
import json
d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step']); print(json.dumps(d['roofline']['per_kernel']))
PY
RW_RGB_F4=0 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench_off.json 2> $OUT/bench_off.err
python -c "
import json
d=json.load(open('$OUT/bench_off.json')); print('off', d['value'], d['ms_per_step'])"
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_model.py -x -q 2>&1 | tail -5 | tee $OUT/test2.log
