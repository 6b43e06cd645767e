#!/bin/bash
# round 6: stand-alone times of the fused upsampling kernel per layer for alternative builds of rw_tconv.hip (rewriting_amd/lib_tc_<name>.so),
# after the kernel's parity cases with each; usage: gpu_tconv_lib_ab.sh <tag> <TY or ""> <layers> <name> ...
OUT=gpurun_out/$1; TY=$2; LAYERS=$3; shift 3; mkdir -p $OUT; : > $OUT/tconv_lib_ab.jsonl
for n in librewriting_hip "$@" librewriting_hip "$@"; do
  L=rewriting_amd/lib_tc_$n.so; [ $n = librewriting_hip ] && L=rewriting_amd/librewriting_hip.so
  [ -f $L ] || continue
  if [ ! -f $OUT/parity_$n.log ]; then
    RW_HIP_LIB=$PWD/$L timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "fused_transposed_conv_and_blur" > $OUT/parity_$n.log 2>&1; echo "$n parity: $(tail -1 $OUT/parity_$n.log)"
  fi
  RW_TCONV_TY=$TY RW_TCONV_ONLY=1 RW_LAYERS=$LAYERS RW_HIP_LIB=$PWD/$L timeout 300 python scripts/tconv_bench.py 2>/dev/null | grep "^{" | \
    python -c "
import json,sys
r={json.loads(l)['layer']: json.loads(l)['fused_ms'] for l in sys.stdin}
print(json.dumps(dict(ty='$TY', lib='$n', **r)))" | tee -a $OUT/tconv_lib_ab.jsonl
done
