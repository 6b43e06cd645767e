#!/bin/bash
# round 6: ToRGB sums left by the producing convolution (rw_dconv3x3_rgb_partial_f32) -- kernel parity, model parity, A/B in the forward
out=gpurun_out/${1:-r06j}; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "leaves_the_to_rgb_sums" > $out/pytest_rgbp.log 2>&1; tail -3 $out/pytest_rgbp.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "generator" > $out/pytest_fullsize_gen.log 2>&1; tail -3 $out/pytest_fullsize_gen.log
for v in 1 0 1 0; do
  echo "RW_RGB_PARTIAL=$v"; RW_RGB_PARTIAL=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity'])" | tee -a $out/ab_rgb_partial.txt
done
