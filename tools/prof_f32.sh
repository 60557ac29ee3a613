#!/usr/bin/env bash
# Runs ON THE GPU BOX: the fp32 (reference-precision) config-3 bench line, then rocprofv3 --kernel-trace --stats of the same command.
# Afterwards HERE: python tools/rocprof_csv_md.py gpurun_out/<dir>/prof_f32 profiles/rNN_config3_f32_rocprof
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-r04f}"
mkdir -p "$OUT"
cd "$R"
python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_f32" -- \
  python "$R/bench.py" --workload config3 --steps 10 --warmup 3 --no-cpu-baseline --no-latency-leg --no-f32-leg --no-live-traffic --check-frames 0 > "$OUT/prof_f32.json" 2> "$OUT/prof_f32.err"
find "$OUT" -name '*kernel_trace.csv' -delete
tail -c 400 "$OUT/bench_default.err"
python - "$OUT/bench_default.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], d["dtype"], "f16", d.get("value_f16"), "roofline", d["roofline"]["achieved"], d["roofline"]["frac"],
      "conv ms", d["roofline"].get("conv_ms_per_step"), "parity", d["parity"]["track_ids_equal_oracle"], "lat", [(l["frames_per_step"], l["n_streams"], round(l["fps"], 1)) for l in (d.get("latency") or [])])
PY
