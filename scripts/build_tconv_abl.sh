#!/bin/bash
# Timing-ablation builds of rw_tconv.hip (results WRONG by construction): rewriting_amd/lib_tc_abl<bits>.so for every <bits> given.
# Usage (repo root): bash scripts/build_tconv_abl.sh 1 2 8 ...   [TC_FLAGS="-DTC_X=1" adds flags; NAME=<suffix> names the library lib_tc_<suffix>.so when one <bits> is given]
R=$PWD; C=$R/rewriting_amd/csrc; mkdir -p /tmp/tcabl
OBJS=$(ls $C/build/*.o | grep -v rw_tconv.o)
EXTRA="$(sed -n 's|^// hipcc-flags: ||p' $C/rw_tconv.hip | head -1)"
for a in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Xclang -target-feature -Xclang -packed-fp32-ops $EXTRA -DTC_ABL=$a $TC_FLAGS -c $C/rw_tconv.hip -o /tmp/tcabl/rw_tconv_$a.o &
done
wait
for a in "$@"; do
  n=${NAME:-abl$a}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/tcabl/rw_tconv_$a.o -o rewriting_amd/lib_tc_$n.so
  echo "built rewriting_amd/lib_tc_$n.so"
done
