#!/bin/bash
OUT=gpurun_out/r02f; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout 600 -p no:cacheprovider --tb=short -k "winograd" > $OUT/pytest_wino.log 2>&1; echo "pytest exit $?"; tail -15 $OUT/pytest_wino.log
export RW_BATCH=64 RW_LAYERS=layer8,layer10,layer12,layer14,layer16,layer18 RW_ALGO=winograd
echo "== gen2 (16x16x4)"; RW_OUT=r02f/cb_g2.json python scripts/conv_bench.py 2>&1 | grep layer
echo "== gen2 gpw=1"; RW_WINO_GPW=1 RW_OUT=r02f/cb_g2_gpw1.json python scripts/conv_bench.py 2>&1 | grep layer
echo "== gen2 gpw=4"; RW_WINO_GPW=4 RW_OUT=r02f/cb_g2_gpw4.json python scripts/conv_bench.py 2>&1 | grep layer
echo "== gen1 (32x32x2 pairs)"; RW_WINO_V=1 RW_OUT=r02f/cb_g1.json python scripts/conv_bench.py 2>&1 | grep layer
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --conv-algo winograd > $OUT/bench_wino.json 2> $OUT/bench_wino.err; echo "bench wino exit $?"; cat $OUT/bench_wino.json; tail -3 $OUT/bench_wino.err
