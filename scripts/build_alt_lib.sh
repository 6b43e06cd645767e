#!/bin/bash
# One source of the library rebuilt with extra flags, linked with the product's other objects:
#   bash scripts/build_alt_lib.sh <name> <source.hip> <flags...>   ->  rewriting_amd/lib_alt_<name>.so   (run csrc/build.sh first)
name=$1; src=$2; shift 2
R=$PWD; C=$R/rewriting_amd/csrc; mkdir -p /tmp/altlib
base=$(basename ${src%.hip})
EXTRA="$(sed -n 's|^// hipcc-flags: ||p' $C/$base.hip | head -1)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Xclang -target-feature -Xclang -packed-fp32-ops $EXTRA "$@" -c $C/$base.hip -o /tmp/altlib/${base}_$name.o 2>&1 | grep -v "is not a recognized feature for this target"
OBJS=$(ls $C/build/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/altlib/${base}_$name.o -o rewriting_amd/lib_alt_$name.so && echo "built rewriting_amd/lib_alt_$name.so"
