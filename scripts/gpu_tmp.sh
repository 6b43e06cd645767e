export RW_LAYERS=layer16,layer17
show() { python -c "
import sys, json
for l in sys.stdin:
    if 'layer' not in l: continue
    d = json.loads(l); print(d['layer'], {k: v for k, v in d.items() if k.endswith('_ms') or k.startswith('rel')})"; }
echo "== product"; python scripts/dconv_bench.py | show
RW_HIP_LIB=$PWD/scripts/probe/abl/lib_dcprof.so python scripts/dconv_prof.py
