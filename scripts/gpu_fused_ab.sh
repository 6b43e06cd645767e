#!/bin/bash
# In-forward A/B on one box, interleaved twice.  usage: bash scripts/gpu_fused_ab.sh <tag> [variant ...]
# (default variants: the forward without / with the fused upsampling kernel, the RGB branch joined before it, side streams
# at a lower priority)
T=${1:-r05r}; O=gpurun_out/$T; mkdir -p $O; : > $O/ab.jsonl; shift
B="--steps 10 --warmup 2 --no-extra --no-cpu-baseline"
if [ $# -eq 0 ]; then set -- "RW_UP_FUSED2=0" "RW_NOP=1" "RW_UP_FUSED2_JOIN=1" "RW_SIDE_PRIORITY=1" "RW_SIDE_PRIORITY=1 RW_UP_FUSED2_JOIN=1"; fi
for REP in 1 2; do
  for V in "$@"; do
    L=$(env $V timeout 300 python bench.py $B 2>$O/ab.err | tail -1)
    python - "$V" "$L" >> $O/ab.jsonl <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(json.dumps(dict(env=sys.argv[1], images_per_s=d['value'], ms_per_step=d['ms_per_step'], parity=d['parity']['linf'],
                      per_kernel={k: (v.get('launches'), v.get('ms')) for k, v in d['roofline'].get('per_kernel', {}).items()})))
PY
    tail -1 $O/ab.jsonl | cut -c1-420
  done
done
