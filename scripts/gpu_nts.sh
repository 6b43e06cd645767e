#!/bin/bash
# non-temporal stores of the conv output maps: A/B libraries under scripts/probe/abl/
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['per_kernel']
print('$1', d['value'], d['ms_per_step'], d['parity']['linf'], {n.split('_kernel')[0]: round(v['ms']/v['launches'],3) for n,v in k.items() if 'wino' in n})"; }
for i in 1 2 3; do
  run base
  for a in w4 uw both; do RW_HIP_LIB=$PWD/scripts/probe/abl/lib_nts_$a.so run nts_$a; done
done
