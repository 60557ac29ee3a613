#!/usr/bin/env bash
# GPU box: depthwise / SPP kernel tests, the pose forward breakdown, config 4 at f16 (and fp32 with a 2nd argument)
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-dw}"
mkdir -p "$OUT"
cd "$R"
timeout 300 python -m pytest tests/test_gpu_dwconv.py tests/test_gpu_spp.py -x -q 2>&1 | tail -15
timeout 200 python tools/probe_rtmpose.py 2400 f16 3 2>&1 | grep -v amdgpu.ids | tee "$OUT/pose_f16.txt"
TLK_CONV_F16=1 timeout 200 python tools/probe_rtmpose.py 2400 f16 3 2>&1 | grep -v amdgpu.ids | tee "$OUT/pose_f16_conv16.txt"
timeout 300 python bench.py --workload config4 --dtype f16 --steps 8 --warmup 3 --no-cpu-baseline --no-latency-leg --no-f32-leg --no-live-traffic --check-frames 96 > "$OUT/bench_config4_f16.json" 2> "$OUT/bench_config4_f16.err"
L="config4_f16"
if [ "${2:-}" = "f32" ]; then
  timeout 300 python bench.py --workload config4 --steps 5 --warmup 2 --no-cpu-baseline --no-latency-leg --no-f32-leg --no-live-traffic --check-frames 48 > "$OUT/bench_config4.json" 2> "$OUT/bench_config4.err"
  L="config4_f16 config4"
fi
for f in $L; do tail -c 300 "$OUT/bench_$f.err"; python - "$OUT/bench_$f.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], "value", round(d["value"], 1), "ms", round(d["ms_per_step"], 1), d["dtype"], "parity", d["parity"]["track_ids_equal_oracle"])
PY
done
