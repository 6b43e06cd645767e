#!/bin/bash
# the edit workload (key collect + 2001-step solve) with two builds of rw_tconv.hip, interleaved
OUT=gpurun_out/$1; mkdir -p $OUT; : > $OUT/edit_ab.txt
for rep in 1 2 3; do for L in librewriting_hip lib_tc_old; do
  v=$(RW_HIP_LIB=$PWD/rewriting_amd/$L.so timeout 300 python bench.py --workload edit --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d.get('value'), {k: v for k, v in d.get('extra', {}).items() if 'edit' in k})")
  echo "$L : $v" | tee -a $OUT/edit_ab.txt
done; done
