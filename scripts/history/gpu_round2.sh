#!/bin/bash
# r02j: winograd tests after the buffer-load change, per-layer conv bench, sweep layers, watermark workload
OUT=gpurun_out/r02j; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_workloads.py -m gpu -q -x --timeout 600 -p no:cacheprovider --tb=short -k "winograd or workloads or frechet or sample_set" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest.log
RW_BATCH=64 RW_LAYERS=layer8,layer10,layer12,layer14,layer16,layer18 RW_ALGO=winograd RW_OUT=r02j/cb_wino.json python scripts/conv_bench.py 2>&1 | grep layer
for L in 8 10 14; do
  timeout 600 python bench.py --workload sweep --size 1024 --layer $L --seeds 10000 --steps 2 --warmup 1 > $OUT/sweep_l$L.json 2> $OUT/sweep_l$L.err; echo "sweep $L exit $?"; cat $OUT/sweep_l$L.json; tail -2 $OUT/sweep_l$L.err
done
timeout 900 python bench.py --workload watermark --steps 1 --warmup 1 > $OUT/watermark.json 2> $OUT/watermark.err; echo "watermark exit $?"; cat $OUT/watermark.json; tail -3 $OUT/watermark.err
