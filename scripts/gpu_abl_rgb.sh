#!/bin/bash
# what the RGB branch costs the step: bench with the skip upsampling / the whole ToRGB ablated (results wrong)
for m in base noup norgb base; do
  echo "== $m"
  RW_ABL_RGB=$m timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('parity'))"
done
