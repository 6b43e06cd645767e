#!/bin/bash
# interleaved A/B of the default forward (batch 64) under alternative libraries; usage: gpu_ab_libs2.sh <tag> <lib.so> ...  (first = the product)
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out; : > $out/ab_libs.txt
for rep in 1 2 3; do
  for lib in "$@"; do
    v=$(RW_HIP_LIB=$PWD/rewriting_amd/$lib timeout 300 python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity']['linf'])")
    echo "$lib : $v" | tee -a $out/ab_libs.txt
  done
done
