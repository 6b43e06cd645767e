#!/bin/bash
# round 6: timing ablations (lib_tc_abl<bits>.so, scripts/build_tconv_abl.sh) of the one-workgroup-per-CU form on the layers it serves
OUT=gpurun_out/${1:-r06o}; mkdir -p $OUT; : > $OUT/t16_abl.jsonl
for L in librewriting_hip lib_tc_abl1 lib_tc_abl2 lib_tc_abl3 lib_tc_abl8 lib_tc_abl10; do
  [ -f rewriting_amd/$L.so ] || continue
  RW_TCONV_TY=16 RW_TCONV_ONLY=1 RW_LAYERS=layer9,layer11,layer13 RW_HIP_LIB=$PWD/rewriting_amd/$L.so timeout 300 python scripts/tconv_bench.py 2>/dev/null | grep "^{" | \
    python -c "
import json,sys
r={json.loads(l)['layer']: json.loads(l)['fused_ms'] for l in sys.stdin}
print(json.dumps(dict(ty=16, lib='$L', **r)))" | tee -a $OUT/t16_abl.jsonl
done
