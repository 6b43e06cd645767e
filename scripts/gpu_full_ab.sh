#!/bin/bash
# the full default bench line (with extras) under alternative libraries, interleaved; usage: gpu_full_ab.sh <tag> <lib.so> ...
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out; : > $out/full_ab.jsonl
for rep in 1 2; do
  for lib in "$@"; do
    RW_HIP_LIB=$PWD/rewriting_amd/$lib timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(dict(lib='$lib', value=d['value'], parity=d['parity']['linf'], **d['extra'])))" | tee -a $out/full_ab.jsonl
  done
done
