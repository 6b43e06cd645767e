#!/bin/bash
TAG=${1:-r04e}; OUT=gpurun_out/$TAG; mkdir -p $OUT; R=$PWD
export RW_BATCH=64 RW_UP_ALGO=wino RW_LAYERS=layer9,layer13,layer15
for mm in f32 split; do
  echo "== up product $mm"; RW_UPW_MM=$mm python scripts/conv_bench.py 2>&1 | grep layer | cut -c1-22,100-190
  for a in 2 4 6 8; do
    echo "== up UW_ABL=$a $mm"; RW_UPW_MM=$mm RW_HIP_LIB=$R/scripts/probe/abl/lib_uwabl_$a.so python scripts/conv_bench.py 2>&1 | grep layer | cut -c1-22,100-190
  done
done
unset RW_UP_ALGO
export RW_ALGO=winograd4 RW_LAYERS=layer10,layer12,layer14,layer16,layer18 RW_W4_MM=split RW_W4H_PS=0
echo "== w4h <2,2>"; python scripts/conv_bench.py 2>&1 | grep layer | cut -c1-22,120-175
echo "== w4h <4,3> wg8"; RW_W4H_WG8=1 python scripts/conv_bench.py 2>&1 | grep layer | cut -c1-22,120-175
