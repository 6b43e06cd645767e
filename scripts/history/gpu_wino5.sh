#!/bin/bash
OUT=gpurun_out/r02g; mkdir -p $OUT
export RW_BATCH=64 RW_LAYERS=layer10,layer14,layer16,layer18 RW_ALGO=winograd
echo "== gen2 product"; RW_OUT=r02g/cb.json python scripts/conv_bench.py 2>&1 | grep layer
for a in 4 16 20; do echo "== gen2 abl $a"; RW_HIP_LIB=$PWD/scripts/probe/lib_wn_abl$a.so RW_OUT=r02g/cb_abl$a.json python scripts/conv_bench.py 2>&1 | grep layer; done
