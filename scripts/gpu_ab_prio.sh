#!/bin/bash
# same-box interleaved A/B: s_setprio 1 / 3 around the 36 MFMAs of a k-quad in the F(4x4,3x3) kernels (-DW4_SETPRIO)
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['per_kernel']
print('$1', d['value'], d['ms_per_step'], {n.split('_kernel')[0]: round(v['ms']/v['launches'],3) for n,v in k.items() if 'wino36' in n})"; }
for i in 1 2; do
  run base
  RW_HIP_LIB=$PWD/scripts/probe/abl/lib_w4_prio1.so run prio1
  RW_HIP_LIB=$PWD/scripts/probe/abl/lib_w4_prio3.so run prio3
done
