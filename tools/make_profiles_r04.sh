#!/usr/bin/env bash
# Runs ON THE GPU BOX (via gpurun): the r04 bench lines of every workload + the fp32 rocprofv3 kernel trace of the default command.
# Afterwards HERE: python tools/collect_profiles_r04.py gpurun_out/round4
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/round4"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
for wl in config2 config4 config5 config3h config1; do
  python bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-latency-leg --no-f32-leg --no-live-traffic --check-frames 96 > "$OUT/bench_$wl.json" 2> "$OUT/bench_$wl.err"
done
python bench.py --workload config2 --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline --no-latency-leg --no-f32-leg --no-live-traffic --check-frames 96 > "$OUT/bench_config2_f16.json" 2> "$OUT/bench_config2_f16.err"
python bench.py --workload config4 --dtype f16 --steps 8 --warmup 3 --no-cpu-baseline --no-latency-leg --no-f32-leg --no-live-traffic --check-frames 96 > "$OUT/bench_config4_f16.json" 2> "$OUT/bench_config4_f16.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_f32" -- \
  python "$R/bench.py" --workload config3 --steps 10 --warmup 3 --no-cpu-baseline --no-latency-leg --no-f32-leg --no-live-traffic --check-frames 0 > "$OUT/prof_f32.json" 2> "$OUT/prof_f32.err"
find "$OUT" -name '*kernel_trace.csv' -delete
cd "$R"; python tests/perf/bench_trackers.py 120 64 > "$OUT/trackers.log" 2>&1
python tools/probe_decode_nms.py > "$OUT/decode_nms.txt" 2>&1
python tools/probe_rtmpose.py 2400 f16 3 2>&1 | grep -v amdgpu.ids > "$OUT/pose_f16.txt"
(timeout 400 python tools/fuzz_gpu.py 300; FUZZ_MAX_OBJECTS=320 FUZZ_BIG_CAPACITY=1 timeout 300 python tools/fuzz_gpu.py 20 ocsort bytetrack botsort deepocsort) 2>&1 | grep "trials\|DIVERGENCE" > "$OUT/fuzz_gpu.txt"
cat "$OUT/fuzz_gpu.txt"
for f in "$OUT"/bench_*.json; do python - "$f" <<'PY'
import json, sys, os
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(os.path.basename(sys.argv[1]), d.get("dtype"), round(d["value"], 1), "frames/s", round(d["ms_per_step"], 1), "ms", "ids==oracle:", (d.get("parity") or {}).get("track_ids_equal_oracle"))
except Exception as e:
    print(os.path.basename(sys.argv[1]), "FAILED", e)
PY
done
