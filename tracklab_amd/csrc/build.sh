#!/usr/bin/env bash
# Build libtlk.so (gfx950 only) in-tree: tracklab_amd/lib/libtlk.so
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT" "$HERE/.obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I"$HERE/../../include" -I"$HERE" -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result)
objs=()
pids=()
for src in "$HERE"/*.hip; do
  obj="$HERE/.obj/$(basename "${src%.hip}").o"
  objs+=("$obj")
  if [[ ! -f "$obj" || "$src" -nt "$obj" || "$HERE/tlk_common.hpp" -nt "$obj" || "$HERE/tlk_strongsort_common.hpp" -nt "$obj" || "$HERE/tlk_cosine.hpp" -nt "$obj" || "$HERE/tlk_bytetrack_common.hpp" -nt "$obj" || "$HERE/tlk_ocsort_common.hpp" -nt "$obj" || "$HERE/../../include/tlk.h" -nt "$obj" ]]; then
    "$HIPCC" "${FLAGS[@]}" -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/libtlk.so" "${objs[@]}" -ldl
echo "built $OUT/libtlk.so"
