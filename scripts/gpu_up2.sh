#!/bin/bash
export RW_BATCH=64 RW_LAYERS=layer11,layer15,layer17 RW_IMPL=7
echo "== product (tiles only, impl 7)"; python scripts/conv_bench.py 2>&1 | grep layer
for a in 1 2 4 16 6 22 23; do echo "== abl $a"; RW_HIP_LIB=$PWD/scripts/probe/lib_conv_abl.so RW_CONV_ABL=$a python scripts/conv_bench.py 2>&1 | grep layer; done
