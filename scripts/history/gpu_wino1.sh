#!/bin/bash
OUT=gpurun_out/r02c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout 600 -p no:cacheprovider --tb=short -k "winograd" > $OUT/pytest_wino.log 2>&1; echo "pytest exit $?"; tail -40 $OUT/pytest_wino.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --conv-algo winograd > $OUT/bench_wino.json 2> $OUT/bench_wino.err; echo "bench wino exit $?"; cat $OUT/bench_wino.json; tail -5 $OUT/bench_wino.err
timeout 900 python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cat $OUT/bench.json; tail -5 $OUT/bench.err
