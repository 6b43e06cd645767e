#!/bin/bash
# round 6: the pipelined persistent form of rw_tconv.hip (RW_TCONV_TY=2) against the specialised one (0) -- parity, then
# stand-alone times per layer; libraries rewriting_amd/lib_pp_mq<k>.so (built with -DTC_PP_MQ=k) are timed too, and
# lib_tc_prof.so (-DTC_PROF=1) leaves cycle counters
out=gpurun_out/${1:-r06e}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "fused_transposed_conv_and_blur and pipelined" > $out/pytest_pp.log 2>&1; tail -3 $out/pytest_pp.log
for ty in 0 2; do
  RW_TCONV_TY=$ty RW_TCONV_ONLY=1 RW_LAYERS=layer13,layer15,layer17 timeout 300 python scripts/tconv_bench.py > $out/tconv_bench_ty$ty.jsonl 2>&1; echo "form $ty"; grep fused_ms $out/tconv_bench_ty$ty.jsonl
done
for lib in rewriting_amd/lib_pp_mq*.so; do
  [ -f $lib ] || continue
  n=$(basename $lib .so)
  RW_HIP_LIB=$PWD/$lib RW_TCONV_TY=2 timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "fused_transposed_conv_and_blur and pipelined" 2>&1 | tail -1
  RW_HIP_LIB=$PWD/$lib RW_TCONV_TY=2 RW_TCONV_ONLY=1 RW_LAYERS=layer13,layer15,layer17 timeout 300 python scripts/tconv_bench.py > $out/tconv_bench_$n.jsonl 2>&1; echo "form 2, $n"; grep fused_ms $out/tconv_bench_$n.jsonl
done
if [ -f rewriting_amd/lib_tc_prof.so ]; then
  RW_TCONV_TY=2 RW_HIP_LIB=$PWD/rewriting_amd/lib_tc_prof.so RW_LAYERS=layer17,layer15 timeout 200 python scripts/tconv_prof.py > $out/prof_ty2.jsonl 2>$out/prof_ty2.err; cat $out/prof_ty2.jsonl
fi
