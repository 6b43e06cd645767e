#!/bin/bash
# same-box interleaved A/B of several builds of the library on the headline forward:  bash scripts/gpu_ab_libs.sh <rounds> <lib.so> ...
R="$1"; shift
for i in $(seq 1 $R); do for L in "$@"; do
  RW_HIP_LIB="$PWD/$L" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['per_kernel']
print('$L', d['value'], d['ms_per_step'], d['parity']['linf'], {n.split('_kernel')[0]: round(v['ms']/v['launches'],3) for n,v in k.items() if v['ms'] > 10})"
done; done
