#!/bin/bash
# stand-alone times per layer and form (RW_TCONV_TY) for the default library and alternative builds lib_tc_<name>.so, after the parity cases
# usage: gpu_tconv_forms_lib.sh <tag> "<forms>" <name> ...
OUT=gpurun_out/$1; FORMS=$2; shift 2; mkdir -p $OUT; : > $OUT/tconv_forms_lib.jsonl
for n in "$@"; do
  RW_HIP_LIB=$PWD/rewriting_amd/lib_tc_$n.so timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "fused_transposed_conv_and_blur" > $OUT/parity_$n.log 2>&1; echo "$n parity: $(tail -1 $OUT/parity_$n.log)"
done
for rep in 1 2; do for n in librewriting_hip "$@"; do
  L=rewriting_amd/lib_tc_$n.so; [ $n = librewriting_hip ] && L=rewriting_amd/librewriting_hip.so
  for ty in $FORMS; do
  RW_HIP_LIB=$PWD/$L RW_TCONV_TY=$ty RW_TCONV_ONLY=1 timeout 300 python scripts/tconv_bench.py 2>/dev/null | grep "^{" | \
    python -c "
import json,sys
r={json.loads(l)['layer']: json.loads(l)['fused_ms'] for l in sys.stdin}
print(json.dumps(dict(lib='$n', ty='$ty', **r)))" | tee -a $OUT/tconv_forms_lib.jsonl
  done
done; done
