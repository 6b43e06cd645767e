#!/bin/bash
OUT=gpurun_out/r02k; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout 600 -p no:cacheprovider --tb=short -k "winograd" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest.log
export RW_BATCH=64 RW_LAYERS=layer8,layer10,layer12,layer14 RW_ALGO=winograd
echo "== tile auto (<4,1> for 128+)"; RW_OUT=r02k/cb_auto.json python scripts/conv_bench.py 2>&1 | grep layer
echo "== tile 64 (<2,2>)"; RW_WINO_TILE=64 RW_OUT=r02k/cb_t64.json python scripts/conv_bench.py 2>&1 | grep layer
