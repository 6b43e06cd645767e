#!/bin/bash
# interleaved A/B of the default forward (batch 64) under environment settings; usage: gpu_ab_env.sh <tag> "ENV=a ENV2=b" "ENV=c" ...
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out; : > $out/ab.txt
for rep in 1 2 3; do
  for setting in "$@"; do
    v=$(env $setting timeout 300 python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity']['linf'])")
    echo "$setting : $v" | tee -a $out/ab.txt
  done
done
