#!/usr/bin/env bash
# GPU box (r05): end-to-end A/B of the routes -- config 3, resident frames, ids checked against the oracle chain in every run
#   fp32: dense ReID batch on / off, direct stem kernel on / off
#   f16 : library route (MIOpen / CK / hipBLASLt + tlk_bias_act) vs libtlk's 16-bit kernels everywhere, at 24 and 1 frames per step
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
one() {  # label, env..., -- bench args
  local label="$1"; shift
  local envs=()
  while [[ "$1" != "--" ]]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --workload config3 --warmup 3 --no-cpu-baseline --no-latency-leg --no-f32-leg --no-live-traffic --no-h2d-leg "$@" 2>/dev/null | python -c "
import json, sys
lines = [l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')]
d = json.loads(lines[-1])
r = d.get('roofline') or {}
print('$label: value', round(d['value'], 1), 'frames/s,', round(d['ms_per_step'], 2), 'ms/step, ids == oracle:', d['parity']['track_ids_equal_oracle'] if d.get('parity') else None,
      '| roofline', r.get('kernel', '')[:24], round(r.get('frac') or 0, 3), r.get('units_per_launch', '')[:80])"
}
if [[ "${1:-all}" == "all" || "$1" == "f32" ]]; then
one "fp32 dense ReID + stem kernel (default)" -- --dtype f32 --steps 8 --check-frames 24
one "fp32 slot layout (TLK_DENSE_REID=0)    " TLK_DENSE_REID=0 -- --dtype f32 --steps 8 --check-frames 24
one "fp32 padded stem (TLK_STEM=0)          " TLK_STEM=0 -- --dtype f32 --steps 8 --check-frames 24
fi
if [[ "${1:-all}" == "all" || "$1" == "f16" ]]; then
for fps in 24 1; do
  one "f16 F=$fps library route   (TLK_CONV_F16=0)" TLK_CONV_F16=0 -- --dtype f16 --frames-per-step $fps --steps $((fps == 1 ? 150 : 10)) --check-frames 24
  one "f16 F=$fps libtlk kernels  (TLK_CONV_F16=1)" TLK_CONV_F16=1 -- --dtype f16 --frames-per-step $fps --steps $((fps == 1 ? 150 : 10)) --check-frames 24
done
fi
