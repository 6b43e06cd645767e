#!/bin/bash
# round 6: parity of the fused upsampling kernel's forms, then stand-alone times per layer and form (RW_TCONV_TY)
OUT=gpurun_out/$1; shift; mkdir -p $OUT; : > $OUT/tconv_forms.jsonl
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "fused_transposed_conv_and_blur" > $OUT/parity.log 2>&1; echo "parity: $(tail -1 $OUT/parity.log)"
for rep in 1 2; do for ty in "$@"; do
  RW_TCONV_TY=$ty RW_TCONV_ONLY=1 timeout 300 python scripts/tconv_bench.py 2>/dev/null | grep "^{" | \
    python -c "
import json,sys
r={json.loads(l)['layer']: json.loads(l)['fused_ms'] for l in sys.stdin}
print(json.dumps(dict(ty='$ty', **r)))" | tee -a $OUT/tconv_forms.jsonl
done; done
