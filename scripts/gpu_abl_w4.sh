#!/bin/bash
# Timing ablations of the F(4x4,3x3) kernels (libraries built with -DW4_ABL=<bits> under scripts/probe/abl/; results of
# an ablated build are WRONG): 1 = no input-transform arithmetic, 2 = no patch fetch, 4 = no weight copies, 8 = no output
# transform / stores, 16 = no barriers.  bash scripts/gpu_abl_w4.sh [layers]
OUT=gpurun_out/abl_w4; mkdir -p $OUT
export RW_BATCH=64 RW_LAYERS=${1:-layer10,layer14,layer16,layer17,layer18} RW_ALGO=winograd4 RW_UP_ALGO=fused
echo "== product"; RW_OUT=abl_w4/product.json python scripts/conv_bench.py 2>&1 | grep layer | cut -c1-40,95-150
for a in 1 2 4 8 16 9 3 11; do
  echo "== W4_ABL=$a"; RW_HIP_LIB=$PWD/scripts/probe/abl/lib_w4abl_$a.so RW_OUT=abl_w4/abl_$a.json python scripts/conv_bench.py 2>&1 | grep layer | cut -c1-40,95-150
done
