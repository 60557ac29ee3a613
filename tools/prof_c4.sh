#!/usr/bin/env bash
# GPU box: kernel mix of the RTMPose forward (config 4) and of the one-frame f16 step (online latency).  Output: gpurun_out/<dir>/
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-c4}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_pose" -- python "$R/tools/probe_rtmpose.py" 2400 f16 3 > "$OUT/pose.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_lat1" -- python "$R/bench.py" --workload config3 --dtype f16 --frames-per-step 1 --steps 100 --warmup 10 \
  --no-cpu-baseline --no-latency-leg --no-f32-leg --no-live-traffic --check-frames 0 > "$OUT/lat1.json" 2> "$OUT/lat1.err"
find "$OUT" -name '*kernel_trace.csv' -delete
tail -2 "$OUT/pose.txt"; tail -c 300 "$OUT/lat1.err"
for d in prof_pose prof_lat1; do
  f=$(find "$OUT/$d" -name '*kernel_stats.csv' | head -1)
  echo "== $d"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms", tot / 1e6)
for r in rows[:28]:
    print(f'{r["Name"][:120]:120s} {int(r["Calls"]):6d} {float(r["TotalDurationNs"])/1e6:9.3f} ms  avg {float(r["AverageNs"])/1e3:9.1f} us  {float(r["Percentage"]):5.1f}%')
PY
done
