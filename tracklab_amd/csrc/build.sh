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
  stale=0
  [[ ! -f "$obj" || "$src" -nt "$obj" ]] && stale=1
  for h in "$HERE"/*.hpp "$HERE/../../include/tlk.h"; do [[ "$h" -nt "$obj" ]] && stale=1; done
  if [[ $stale == 1 ]]; then
    "$HIPCC" "${FLAGS[@]}" -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/libtlk.so" "${objs[@]}" -ldl
echo "built $OUT/libtlk.so"
