#!/bin/bash
# Builds librewriting_hip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.  Each source is compiled
# to its own object (in parallel, only when it is newer than its object) and the objects are linked.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${RW_LIB_OUT:-$HERE/../librewriting_hip.so}"
OBJ="${RW_OBJ_DIR:-$HERE/build}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $RW_EXTRA_FLAGS"
mkdir -p "$OBJ"
if [ "$(cat "$OBJ/.flags" 2>/dev/null)" != "$FLAGS $*" ]; then rm -f "$OBJ"/*.o; echo "$FLAGS $*" > "$OBJ/.flags"; fi
pids=()
for src in "$HERE"/*.hip; do
  obj="$OBJ/$(basename "${src%.hip}").o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$HERE/rw_common.h" -nt "$obj" ] || [ "$HERE/../../include/rewriting_hip.h" -nt "$obj" ]; then
    # per-file flags: a leading comment line "// hipcc-flags: ..." in the source
    extra="$(sed -n 's|^// hipcc-flags: ||p' "$src" | head -1)"
    "$HIPCC" $FLAGS $extra "$@" -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "$OBJ"/*.o -o "$OUT"
echo "built $OUT"
