#!/bin/bash
# Round 4: the F(4x4,3x3) kernels on the 16-bit matrix pipe (f16 operand pairs) against the fp32 ones -- parity tests,
# then per-layer times at batch 64.  Usage: gpurun -- 'bash scripts/gpu_w4h.sh <tag>'
TAG=${1:-r04b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -m pytest tests/test_gpu_kernels.py -x -q -k "winograd_f4 or one_pass_upsampling" 2>&1 | tail -15 > $OUT/pytest_w4h.log
cat $OUT/pytest_w4h.log
export RW_BATCH=64 RW_ALGO=winograd4 RW_LAYERS=layer10,layer12,layer14,layer16,layer18
for mm in f32 split; do
  for ps in 0 1; do
    [ $mm = f32 ] && [ $ps = 1 ] && continue
    RW_W4_MM=$mm RW_W4H_PS=$ps RW_OUT=$TAG/conv_bench_${mm}_ps${ps}.json python scripts/conv_bench.py 2>&1 | grep layer
  done
done
export RW_LAYERS=layer15,layer17 RW_UP_ALGO=fused
for mm in f32 split; do
  for ps in 0 1; do
    [ $mm = f32 ] && [ $ps = 1 ] && continue
    RW_W4_MM=$mm RW_W4H_PS=$ps RW_OUT=$TAG/conv_bench_up_${mm}_ps${ps}.json python scripts/conv_bench.py 2>&1 | grep layer
  done
done
