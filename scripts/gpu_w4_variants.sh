#!/bin/bash
# F(4x4,3x3) stride-1 kernel: workgroup shapes / run lengths per layer (conv_bench at batch 64)
OUT=gpurun_out/w4var; mkdir -p $OUT
export RW_BATCH=64 RW_LAYERS=${1:-layer10,layer12,layer14,layer16,layer18} RW_ALGO=winograd4
echo "== default <2,2> gpw16"; RW_OUT=w4var/v3.json python scripts/conv_bench.py 2>&1 | grep layer | cut -c1-40,95-150
echo "== <4,3> gpw4";  RW_WINO4_V=2 RW_OUT=w4var/v2.json python scripts/conv_bench.py 2>&1 | grep layer | cut -c1-40,95-150
echo "== <4,3> gpw8";  RW_WINO4_V=2 RW_WINO4_GPW=8 RW_OUT=w4var/v2g8.json python scripts/conv_bench.py 2>&1 | grep layer | cut -c1-40,95-150
echo "== <4,3> gpw16"; RW_WINO4_V=2 RW_WINO4_GPW=16 RW_OUT=w4var/v2g16.json python scripts/conv_bench.py 2>&1 | grep layer | cut -c1-40,95-150
echo "== <2,2> gpw4";  RW_WINO4_GPW=4 RW_OUT=w4var/v3g4.json python scripts/conv_bench.py 2>&1 | grep layer | cut -c1-40,95-150
