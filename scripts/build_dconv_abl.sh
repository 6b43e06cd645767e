#!/bin/bash
# Timing-ablation builds of rw_dconv.hip (results WRONG): scripts/probe/abl/lib_dcabl_<bits>.so for every <bits> given.
# Usage (repo root): bash scripts/build_dconv_abl.sh 1 2 4 8 ["-DDC_X=1" as DC_FLAGS env]
set -e
R=$PWD; C=$R/rewriting_amd/csrc; mkdir -p scripts/probe/abl /tmp/dcabl
(cd $C && bash build.sh > /dev/null)
for a in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Xclang -target-feature -Xclang -packed-fp32-ops -DDC_ABL=$a $DC_FLAGS -c $C/rw_dconv.hip -o /tmp/dcabl/rw_dconv_$a.o &
done
wait
for a in "$@"; do
  OBJS=$(ls $C/build/*.o | grep -v rw_dconv.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/dcabl/rw_dconv_$a.o -o scripts/probe/abl/lib_dcabl_$a.so
  echo "built scripts/probe/abl/lib_dcabl_$a.so"
done
