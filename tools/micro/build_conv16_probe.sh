#!/usr/bin/env bash
# Build the standalone convolution probes (gfx950) against the in-tree libtlk.so.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$HERE/../.."
for name in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -I"$ROOT/include" "$HERE/$name.hip" -o "$HERE/$name" \
      -L"$ROOT/tracklab_amd/lib" -ltlk -Wl,-rpath,'$ORIGIN/../../tracklab_amd/lib'
  echo "built $HERE/$name"
done
