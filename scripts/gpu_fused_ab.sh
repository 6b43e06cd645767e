#!/bin/bash
# In-forward A/B on one box, interleaved twice: the default forward against the fused upsampling kernel (rw_tconv.hip) on
# every upsampling layer / on the layers of at most 128 input channels, in its automatic form selection and forced forms.
# usage: bash scripts/gpu_fused_ab.sh <tag>
T=${1:-r05r}; O=gpurun_out/$T; mkdir -p $O; : > $O/ab.jsonl
B="--steps 10 --warmup 2 --no-extra --no-cpu-baseline"
for REP in 1 2; do
  for V in "RW_NOP=1" "RW_UP_FUSED2=1 RW_UP_FUSED2_MAX_IN=512" "RW_UP_FUSED2=1 RW_UP_FUSED2_MAX_IN=128" \
           "RW_UP_FUSED2=1 RW_UP_FUSED2_MAX_IN=512 RW_TCONV_TY=16" "RW_UP_FUSED2=1 RW_UP_FUSED2_MAX_IN=512 RW_TCONV_TY=0"; do
    L=$(env $V timeout 300 python bench.py $B 2>$O/ab.err | tail -1)
    python - "$V" "$L" >> $O/ab.jsonl <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(json.dumps(dict(env=sys.argv[1], images_per_s=d['value'], ms_per_step=d['ms_per_step'], parity=d['parity']['linf'],
                      per_kernel={k: (v.get('launches'), v.get('ms')) for k, v in d['roofline'].get('per_kernel', {}).items()})))
PY
    tail -1 $O/ab.jsonl | cut -c1-600
  done
done
