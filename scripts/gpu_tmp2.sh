export RW_LAYERS=layer16,layer17
show() { python -c "
import sys, json
for l in sys.stdin:
    if 'layer' not in l: continue
    d = json.loads(l); print(d['layer'], {k: v for k, v in d.items() if k in ('direct16_ms', 'f4_split_ms')})"; }
for a in 1 4 5 16 17; do echo "== ABL $a"; RW_HIP_LIB=$PWD/scripts/probe/abl/lib_dcabl_$a.so python scripts/dconv_bench.py | show; done
