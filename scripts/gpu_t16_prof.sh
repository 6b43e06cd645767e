#!/bin/bash
OUT=gpurun_out/${1:-r06q}; mkdir -p $OUT
for n in prof prof_fa; do
  [ -f rewriting_amd/lib_tc_$n.so ] || continue
  RW_HIP_LIB=$PWD/rewriting_amd/lib_tc_$n.so timeout 300 python scripts/t16_prof.py 2>/dev/null | grep "^{" | tee $OUT/t16_$n.jsonl
done
