#!/usr/bin/env bash
# Build libtlk.so (gfx950 only) in-tree: tracklab_amd/lib/libtlk.so
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT" "$HERE/.obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I"$HERE/../../include" -I"$HERE" -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result)
# No packed-FP32 VALU code (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) in kernels that may share compute units with the networks'
# MFMA kernels, i.e. everything launched on a side stream (trackers, camera-motion estimators, evaluation): measured on MI355X, the
# LK kernel of tlk_cmc.hip, whose dx/dy update the compiler had packed, gave lane-dependent results in ~0.1 % of its iterations while a
# ResNet-50 forward ran on another stream and exact ones alone (profiles/r02_pk_f32_overlap.md); without packed code it is exact in both.
# The pre/post-processing kernels below run in stream order with the networks and keep the default code generation they were measured with.
PACKED_OK="tlk_image tlk_pil tlk_nms tlk_epilogue tlk_pose tlk_gemm"
# tlk_dwconv spells its packed FMAs in assembly, in the plain form only (no op_sel: the form measured safe, profiles/r03_pk_f32_root_cause.md);
# the assembler needs the feature for them, the compiler's own vectoriser stays off, and tools/audit_pk_f32.py checks the result
PACKED_ASM="tlk_dwconv"
NOPK=(-fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops)
objs=()
pids=()
for src in "$HERE"/*.hip; do
  obj="$HERE/.obj/$(basename "${src%.hip}").o"
  objs+=("$obj")
  stale=0
  [[ ! -f "$obj" || "$src" -nt "$obj" ]] && stale=1
  for h in "$HERE"/*.hpp "$HERE/../../include/tlk.h" "$HERE/build.sh"; do [[ "$h" -nt "$obj" ]] && stale=1; done
  if [[ $stale == 1 ]]; then
    extra=("${NOPK[@]}")
    for ok in $PACKED_OK; do [[ "$(basename "${src%.hip}")" == "$ok" ]] && extra=(); done
    for ok in $PACKED_ASM; do [[ "$(basename "${src%.hip}")" == "$ok" ]] && extra=(-fno-slp-vectorize); done
    # (the host pass of the same command does not know the AMDGPU feature and says so: filtered, everything else is shown)
    "$HIPCC" "${FLAGS[@]}" "${extra[@]}" -c "$src" -o "$obj" 2> >(grep -v "is not a recognized feature for this target" >&2) &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/libtlk.so" "${objs[@]}" -ldl
echo "built $OUT/libtlk.so"
