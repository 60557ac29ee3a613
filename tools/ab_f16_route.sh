#!/usr/bin/env bash
# GPU box: f16 backbones on libtlk's 16-bit convolution kernel (TLK_CONV_F16=1) vs the library route, at 1 and 24 frames per step
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
for fps in 1 24; do for r in 0 1; do
  TLK_CONV_F16=$r python bench.py --workload config3 --dtype f16 --frames-per-step $fps --steps $((fps == 1 ? 150 : 10)) --warmup 5 --no-cpu-baseline --no-latency-leg --no-f32-leg \
    --no-live-traffic --check-frames 24 --no-h2d-leg 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('frames_per_step $fps TLK_CONV_F16=$r: value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), 'parity', d['parity']['track_ids_equal_oracle'])"
done; done
