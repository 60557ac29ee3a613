#!/usr/bin/env bash
# Debug build of libtlk with 7 KB of unused LDS AHEAD of the lists (the layout move of r01's fault) and LDS guard words in the OC-SORT / Deep-OC-SORT work-area carve (tlk_ocsort_common.hpp, -DTLK_LDS_CANARY -DTLK_LDS_PREPAD=7168):
# tracklab_amd/lib/libtlk_canary.so = the two tracker objects rebuilt with the guards + every other object of the normal build.
# Used by tests/test_gpu_canary.py (TLK_LIB_PATH selects the library).  Run tracklab_amd/csrc/build.sh first.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/../tracklab_amd/csrc" && pwd)"
OUT="$HERE/../lib"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I"$HERE/../../include" -I"$HERE" -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result
       -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops -DTLK_LDS_CANARY -DTLK_LDS_PREPAD=7168)
objs=()
pids=()
for src in "$HERE"/*.hip; do
  base="$(basename "${src%.hip}")"
  if [[ "$base" == "tlk_ocsort" || "$base" == "tlk_deepocsort" ]]; then
    obj="$HERE/.obj/${base}_canary.o"
    if [[ ! -f "$obj" || "$src" -nt "$obj" || "$HERE/tlk_ocsort_common.hpp" -nt "$obj" || "$HERE/tlk_common.hpp" -nt "$obj" ]]; then
      "$HIPCC" "${FLAGS[@]}" -c "$src" -o "$obj" 2> >(grep -v "is not a recognized feature for this target" >&2) &
      pids+=($!)
    fi
  else
    obj="$HERE/.obj/${base}.o"
    [[ -f "$obj" ]] || { echo "missing $obj: run tracklab_amd/csrc/build.sh first" >&2; exit 1; }
  fi
  objs+=("$obj")
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/libtlk_canary.so" "${objs[@]}" -ldl
echo "built $OUT/libtlk_canary.so"
