#!/bin/bash
# The specialised persistent form of rw_tconv.hip: parity (every form), per-layer times of the forms side by side, cycle
# counters from inside the kernel (lib_tc_prof.so = the library built with -DTC_PROF=1, if present).
# usage: bash scripts/gpu_tconv_ws.sh <tag> [forms, default "0 16"]
T=${1:-r05o}; O=gpurun_out/$T; mkdir -p $O; FORMS=${2:-0 16}
# one small case first, under its own short limit: a barrier mismatch hangs the kernel
timeout 120 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "fused_transposed and specialised and case2" > $O/first.log 2>&1
echo "first: $?" >> $O/first.log; tail -3 $O/first.log
if ! grep -q "passed" $O/first.log; then exit 1; fi
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "fused_transposed" > $O/pytest_tconv.log 2>&1
tail -8 $O/pytest_tconv.log
: > $O/tconv_forms.jsonl
for TY in $FORMS; do
  RW_TCONV_TY=$TY RW_TCONV_ONLY=1 timeout 300 python scripts/tconv_bench.py 2>/dev/null | grep "^{" | sed "s/^{/{\"form\": \"$TY\", /" >> $O/tconv_forms.jsonl
done
cat $O/tconv_forms.jsonl
if [ -f rewriting_amd/lib_tc_prof.so ]; then
  RW_HIP_LIB=$PWD/rewriting_amd/lib_tc_prof.so RW_LAYERS=layer17,layer15,layer13,layer9 timeout 300 python scripts/tconv_prof.py > $O/prof.jsonl 2>$O/prof.err
  cat $O/prof.jsonl
fi
