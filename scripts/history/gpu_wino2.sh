#!/bin/bash
OUT=gpurun_out/r02d; mkdir -p $OUT
export RW_BATCH=64 RW_LAYERS=layer8,layer10,layer12,layer14,layer16,layer18
echo "== direct"; RW_OUT=r02d/cb_direct.json python scripts/conv_bench.py 2>&1 | grep layer
echo "== wino default (AD2, block auto)"; RW_ALGO=winograd RW_OUT=r02d/cb_wino.json python scripts/conv_bench.py 2>&1 | grep layer
echo "== wino AD1"; RW_WINO_AD=1 RW_ALGO=winograd RW_OUT=r02d/cb_wino_ad1.json python scripts/conv_bench.py 2>&1 | grep layer
echo "== wino block 0"; RW_WINO_BLOCK=0 RW_ALGO=winograd RW_OUT=r02d/cb_wino_b0.json python scripts/conv_bench.py 2>&1 | grep layer
echo "== wino block 128"; RW_WINO_BLOCK=128 RW_ALGO=winograd RW_OUT=r02d/cb_wino_b128.json python scripts/conv_bench.py 2>&1 | grep layer
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout 600 -p no:cacheprovider --tb=short -k "winograd" > $OUT/pytest_wino.log 2>&1; echo "pytest exit $?"; tail -15 $OUT/pytest_wino.log
