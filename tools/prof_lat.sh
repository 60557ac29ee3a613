#!/usr/bin/env bash
# GPU box: what the one-frame-per-step f16 chain spends its time on (steady state: tail of the kernel trace)
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-lat}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_lat1 -- python "$R/bench.py" --workload config3 --dtype f16 --frames-per-step 1 --steps 150 --warmup 10 \
  --no-cpu-baseline --no-latency-leg --no-f32-leg --no-live-traffic --check-frames 0 --no-h2d-leg > "$OUT/lat1.json" 2> "$OUT/lat1.err"
tail -c 200 "$OUT/lat1.err"
python - "$OUT/lat1.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"])
PY
python "$R/tools/trace_tail_summary.py" /tmp/prof_lat1 400 40 | tee "$OUT/lat1_tail.txt"
