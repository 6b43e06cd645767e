#!/bin/bash
OUT=gpurun_out/r02i; mkdir -p $OUT
for b in 1 2 4 8 16; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --batch $b > $OUT/bench_b$b.json 2> $OUT/bench_b$b.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_b$b.json"))
pk=d['roofline']['per_kernel']
print("batch $b: %.1f img/s  %.2f ms/step  " % (d['value'], d['ms_per_step']), {k.replace('conv_','').replace('_kernel',''): v['tflops'] for k,v in list(pk.items())[:5]})
PY
done
