#!/bin/bash
# same-box A/B of the border-strip kernels: the GEMM strip kernel against the batched im2col launch
for rep in 1 2; do
for m in gemm im2col; do
  echo "== $m"
  RW_UP_STRIPS=$m timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['parity']['linf'])"
done
done
for m in gemm im2col; do
  RW_UP_STRIPS=$m timeout 300 python bench.py --workload sweep --steps 3 --warmup 1 2>/dev/null | tail -1 | cut -c1-140
done
