#!/bin/bash
# blur + noise + act with non-temporal loads (1) / stores (2) / both (3): A/B libraries under scripts/probe/abl/
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['parity']['linf'])"; }
for i in 1 2; do
  run base
  for a in 1 2 3; do RW_HIP_LIB=$PWD/scripts/probe/abl/lib_blnt_$a.so run blur_nt$a; done
done
python scripts/blur_probe.py 2>&1 | tail -4
for a in 1 2 3; do echo "nt=$a"; RW_HIP_LIB=$PWD/scripts/probe/abl/lib_blnt_$a.so python scripts/blur_probe.py 2>&1 | tail -4; done
