#!/bin/bash
# round 6: timing ablations (lib_tc_abl<bits>.so, scripts/build_tconv_abl.sh) of tconv_body's forms on the layers they serve
OUT=gpurun_out/${1:-r06o}; mkdir -p $OUT; : > $OUT/t16_abl.jsonl
for L in librewriting_hip $(ls rewriting_amd/lib_tc_abl*.so | xargs -n1 basename | sed 's/.so$//' | sort -t l -k3 -n); do
  [ -f rewriting_amd/$L.so ] || continue
  for ty in 16 32; do
  RW_TCONV_TY=$ty RW_TCONV_ONLY=1 RW_LAYERS=${RW_LAYERS:-layer11,layer13} RW_HIP_LIB=$PWD/rewriting_amd/$L.so timeout 300 python scripts/tconv_bench.py 2>/dev/null | grep "^{" | \
    python -c "
import json,sys
r={json.loads(l)['layer']: json.loads(l)['fused_ms'] for l in sys.stdin}
print(json.dumps(dict(ty=$ty, lib='$L', **r)))" | tee -a $OUT/t16_abl.jsonl
  done
done
