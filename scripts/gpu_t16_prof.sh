#!/bin/bash
# cycle counters of tconv_body's forms (RW_TCONV_TY = 16 / 32) from a -DTC_PROF=1 build: bash scripts/gpu_t16_prof.sh <tag> [lib names]
OUT=gpurun_out/${1:-r06q}; shift; mkdir -p $OUT
for n in ${@:-prof}; do
  [ -f rewriting_amd/lib_tc_$n.so ] || continue
  for ty in 16 32; do
    RW_TCONV_TY=$ty RW_HIP_LIB=$PWD/rewriting_amd/lib_tc_$n.so timeout 300 python scripts/t16_prof.py 2>/dev/null | grep "^{" | sed "s/^{/{\"ty\": $ty, /" | tee -a $OUT/t16_$n.jsonl
  done
done
