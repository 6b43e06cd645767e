#!/bin/bash
# The specialised persistent form of rw_tconv.hip: parity (every form), then per-layer times of the forms side by side.
# usage: bash scripts/gpu_tconv_ws.sh <tag>
T=${1:-r05o}; O=gpurun_out/$T; mkdir -p $O
# one small case first, under its own short limit: a barrier mismatch hangs the kernel
timeout 120 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "fused_transposed and specialised and case2" > $O/first.log 2>&1
echo "first: $?" >> $O/first.log; tail -3 $O/first.log
if ! grep -q "passed" $O/first.log; then exit 1; fi
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "fused_transposed" > $O/pytest_tconv.log 2>&1
tail -15 $O/pytest_tconv.log
for TY in 0 16 8; do
  RW_TCONV_TY=$TY RW_TCONV_ONLY=1 timeout 300 python scripts/tconv_bench.py 2>/dev/null | grep "^{" | sed "s/^{/{\"form\": \"$TY\", /" >> $O/tconv_forms.jsonl
done
cat $O/tconv_forms.jsonl
