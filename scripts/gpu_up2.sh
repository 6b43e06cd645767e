#!/bin/bash
# ablations of conv_up_halo_kernel (library built with -DRW_ABLATION): 1 = no B reads, 2 = no weight loads, 4 = no patch fetch / staging,
# 8 = no barriers, 16 = no stores
export RW_BATCH=64 RW_LAYERS=${RW_LAYERS:-layer9,layer11,layer15,layer17} RW_IMPL=7
echo "== product (tiles only, impl 7)"; python scripts/conv_bench.py 2>&1 | grep layer | cut -c1-60,95-140
for a in 2 4 6 8 16 22 30 31; do echo "== abl $a"; RW_HIP_LIB=$PWD/scripts/probe/lib_conv_abl.so RW_CONV_ABL=$a python scripts/conv_bench.py 2>&1 | grep layer | cut -c1-60,95-140; done
