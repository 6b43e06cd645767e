#!/bin/bash
# registers / spills / LDS of every kernel of one source file: bash scripts/kernel_resources.sh rewriting_amd/csrc/rw_tconv.hip [extra flags]
src=$1; shift
extra="$(sed -n 's|^// hipcc-flags: ||p' "$src" | head -1)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Xclang -target-feature -Xclang -packed-fp32-ops $extra "$@" -Rpass-analysis=kernel-resource-usage -c "$src" -o /tmp/kr_$$.o 2>&1 | \
  python3 -c "
import re,sys
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r'remark:\s+(.*?) \[-Rpass', l)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith('Function Name:'): cur=t.split(':',1)[1].strip(); rows[cur]={}
    elif cur and ':' in t:
        k,v=t.split(':',1); rows[cur][k.strip()]=v.strip()
for k,r in rows.items():
    print('%-60s VGPR %s AGPR %s spillV %s spillS %s scratch %s LDS %s occ %s' % (k[:60], r.get('VGPRs'), r.get('AGPRs'), r.get('VGPRs Spill'), r.get('SGPRs Spill'), r.get('ScratchSize [bytes/lane]'), r.get('LDS Size [bytes/block]'), r.get('Occupancy [waves/SIMD]')))
"
rm -f /tmp/kr_$$.o
