#!/bin/bash
# Round 4: the F(2,2) transposed convolution on the 16-bit matrix pipe against the fp32 kernel: parity, then per-layer
# times at batch 64.  gpurun -- 'bash scripts/gpu_upwh.sh <tag>'
TAG=${1:-r04d}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -m pytest tests/test_gpu_kernels.py -x -q -k "transposed_conv_f22" 2>&1 | tail -8 | tee $OUT/pytest_upwh.log
export RW_BATCH=64 RW_UP_ALGO=wino RW_LAYERS=layer9,layer11,layer13,layer15,layer17
for mm in f32 split; do
  RW_UPW_MM=$mm RW_OUT=$TAG/conv_bench_upw_${mm}.json python scripts/conv_bench.py 2>&1 | grep layer | cut -c1-22,100-190
done
