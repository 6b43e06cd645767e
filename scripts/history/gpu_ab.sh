#!/bin/bash
# same-box A/B of two libraries on the headline forward, interleaved (3 rounds), and the F(2,2) run length
OUT=gpurun_out/ab; mkdir -p $OUT
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['per_kernel']
print('$1', d['value'], d['ms_per_step'], {n.split('_kernel')[0]: round(v['ms']/v['launches'],3) for n,v in k.items() if 'wino' in n})"; }
for i in 1 2 3; do
  run new
  RW_HIP_LIB=$PWD/scripts/probe/abl/lib_oldw4.so run old
done
for g in 2 8 16; do RW_UPWINO_GPW=$g run upwino_gpw$g; done
RW_WINO4_GPW=8 run wino4_gpw8
RW_WINO4_GPW=4 run wino4_gpw4
