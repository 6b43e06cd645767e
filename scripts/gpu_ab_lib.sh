#!/bin/bash
# same-box interleaved A/B of two builds of the library on the headline forward:  bash scripts/gpu_ab_lib.sh <libA.so> <libB.so>
A="$1"; B="$2"
run() { RW_HIP_LIB="$2" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['per_kernel']
print('$1', d['value'], d['ms_per_step'], d['parity']['linf'], {n.split('_kernel')[0]: round(v['ms']/v['launches'],3) for n,v in k.items() if 'wino36' in n or 'up_wino' in n or 'dconv' in n})"; }
for i in 1 2 3; do run A "$A"; run B "$B"; done
