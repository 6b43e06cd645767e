#!/bin/bash
# same-box A/B of the point split of the F(4x4,3x3) kernels (RW_W4_PSPLIT=1; build the library with -DW4_PSPLIT=1 first:
# RW_EXTRA_FLAGS=-DW4_PSPLIT=1 bash rewriting_amd/csrc/build.sh): kernel tests under both, then the forward
for m in 0 1; do
  echo "== tests RW_W4_PSPLIT=$m"
  RW_W4_PSPLIT=$m timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "winograd_f4 or one_pass or to_rgb" 2>&1 | tail -3
done
for rep in 1 2; do
for m in 0 1; do
  echo "== bench RW_W4_PSPLIT=$m"
  RW_W4_PSPLIT=$m timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['parity']['linf'], {k: v['ms'] for k, v in d['roofline']['per_kernel'].items() if 'wino36' in k})"
done
done
