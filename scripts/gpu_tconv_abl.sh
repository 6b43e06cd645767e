#!/bin/bash
# timing ablations of rw_tconv.hip (libraries built with -DTC_ABL=n: results wrong by construction)
OUT=gpurun_out/${1:-r05g}; mkdir -p $OUT; : > $OUT/tconv_abl.jsonl
for TY in 16 8; do
  for L in librewriting_hip lib_tc_abl1 lib_tc_abl2 lib_tc_abl3 lib_tc_abl8 lib_tc_abl16 lib_tc_abl32 lib_tc_abl64; do
    [ -f rewriting_amd/$L.so ] || continue
    RW_TCONV_TY=$TY RW_TCONV_ONLY=1 RW_LAYERS=layer13,layer17 RW_HIP_LIB=$PWD/rewriting_amd/$L.so python scripts/tconv_bench.py 2>/dev/null | grep "^{" | \
      python -c "
import json,sys
r={json.loads(l)['layer']: json.loads(l)['fused_ms'] for l in sys.stdin}
print(json.dumps(dict(ty=$TY, lib='$L', **r)))" | tee -a $OUT/tconv_abl.jsonl
  done
done
