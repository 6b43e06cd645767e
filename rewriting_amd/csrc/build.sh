#!/bin/bash
# Builds librewriting_hip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../librewriting_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
  -Wno-unused-result \
  "$HERE/rw_ops.hip" "$HERE/rw_conv.hip" "$HERE/rw_stats.hip" "$HERE/rw_solve.hip" \
  -o "$OUT" "$@"
echo "built $OUT"
