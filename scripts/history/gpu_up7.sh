#!/bin/bash
OUT=gpurun_out/up7; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -k "winograd_f4 or one_pass_upsampling or premultiplied or golden or blur" 2>&1 | tail -3 | tee $OUT/test.log
for i in 1 2; do
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench$i.json 2> $OUT/bench.err; echo "bench exit $?"
python - <<PY
import json
d=json.load(open('$OUT/bench$i.json')); print(d['value'], d['ms_per_step']); print(json.dumps({k:v for k,v in d['roofline']['per_kernel'].items() if 'wino36' in k}))
PY
done
RW_PRESCALE=0 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench_off.json 2> $OUT/bench_off.err
python -c "
import json
d=json.load(open('$OUT/bench_off.json')); print('off', d['value'], d['ms_per_step'])"
