#!/usr/bin/env bash
# Runs ON THE GPU BOX (via gpurun): everything profiles/r06_* is built from, under gpurun_out/round6/.
#   part "bench"  : the driver-form default line + the other workloads' lines
#   part "trace"  : rocprofv3 --kernel-trace --stats of the fp32 and f16 commands (+ steady-state tails of the f16 ones)
#   part "pmc"    : FETCH_SIZE / WRITE_SIZE of every convolution launch of the fp32 command (separate passes, kernel-trace only)
#   part "probes" : the standalone convolution probes, the split leg under rocprofv3, the tracker bench, the precision envelope
# Afterwards HERE: python tools/collect_profiles_r06.py gpurun_out/round6
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/round6"
mkdir -p "$OUT"
PARTS="${1:-bench trace pmc probes}"
SHORT="--no-cpu-baseline --no-latency-leg --no-f32-leg --no-live-traffic"
cd "$R"
if [[ "$PARTS" == *bench* ]]; then
  python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
  for wl in config2 config4 config5 config3h config1; do
    timeout 600 python bench.py --workload $wl --steps 6 --warmup 2 $SHORT --check-frames 96 > "$OUT/bench_$wl.json" 2> "$OUT/bench_$wl.err"
  done
  for wl in config2 config4 config3h; do
    timeout 600 python bench.py --workload $wl --dtype f16 --steps 10 --warmup 3 $SHORT --check-frames 96 > "$OUT/bench_${wl}_f16.json" 2> "$OUT/bench_${wl}_f16.err"
  done
fi
if [[ "$PARTS" == *trace* ]]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_f32" -- \
    python "$R/bench.py" --workload config3 --steps 10 --warmup 3 $SHORT --no-h2d-leg --check-frames 0 > "$OUT/prof_f32.json" 2> "$OUT/prof_f32.err"
  python "$R/tools/trace_tail_summary.py" "$OUT/prof_f32" 1200 40 > "$OUT/tail_f32.txt"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_f16" -- \
    python "$R/bench.py" --workload config3 --dtype f16 --steps 12 --warmup 3 $SHORT --no-h2d-leg --check-frames 0 > "$OUT/prof_f16.json" 2> "$OUT/prof_f16.err"
  python "$R/tools/trace_tail_summary.py" "$OUT/prof_f16" 400 45 > "$OUT/tail_f16.txt"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_f16_lat" -- \
    python "$R/bench.py" --workload config3 --dtype f16 --frames-per-step 1 --steps 150 --warmup 10 $SHORT --no-h2d-leg --check-frames 0 > "$OUT/prof_f16_lat.json" 2> "$OUT/prof_f16_lat.err"
  python "$R/tools/trace_tail_summary.py" "$OUT/prof_f16_lat" 400 45 > "$OUT/tail_f16_lat.txt"
  find "$OUT" -name '*kernel_trace.csv' -delete
  cd "$R"
fi
if [[ "$PARTS" == *pmc* ]]; then
  cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "/tmp/pmc_$c" -- \
      python "$R/bench.py" --workload config3 --steps 2 --warmup 1 $SHORT --no-h2d-leg --check-frames 0 > "$OUT/pmc_$c.json" 2> "$OUT/pmc_$c.err"
  done
  python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for path in glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            n = r["Kernel_Name"]
            if r["Counter_Name"] != c or not any(k in n for k in ("conv_f32_mfma_kernel", "conv16x_kernel", "conv_stem3_kernel")):
                continue
            agg[n[:100]][c].append(float(r["Counter_Value"]))
rows, tot = [], collections.Counter()
for n, d in agg.items():
    f, w = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
    rows.append({"kernel": n, "launches_fetch_pass": len(f), "launches_write_pass": len(w),
                 "fetch_bytes_per_launch": 2 * 1024 * sum(f) / max(len(f), 1),         # KB, x 2: the guide's gfx950 correction for wide coalesced reads
                 "write_bytes_per_launch": 1024 * sum(w) / max(len(w), 1)})
    tot["fetch"] += 2 * 1024 * sum(f); tot["nf"] += len(f); tot["write"] += 1024 * sum(w); tot["nw"] += len(w)
res = {"per_instantiation": sorted(rows, key=lambda r: -(r["fetch_bytes_per_launch"] + r["write_bytes_per_launch"]) * r["launches_fetch_pass"]),
       "mean_traffic_bytes_per_conv_launch": tot["fetch"] / max(tot["nf"], 1) + tot["write"] / max(tot["nw"], 1),
       "conv_launches_seen": [tot["nf"], tot["nw"]]}
json.dump(res, open(out + "/pmc_conv_f32.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "per_instantiation"}))
PY
  cd "$R"
fi
if [[ "$PARTS" == *probes* ]]; then
  cd "$R"
  timeout 300 tools/micro/conv16_probe 2400 split -1,0 > "$OUT/conv16x_split2400.txt" 2>&1
  timeout 300 tools/micro/conv16_probe 2400 f16 -1,0 > "$OUT/conv16x_reid2400.txt" 2>&1
  timeout 300 tools/micro/conv16_probe 100 f16 -1,0 > "$OUT/conv16x_reid100.txt" 2>&1
  timeout 300 tools/micro/conv16_probe 24 f16 -1,0 "" yolox > "$OUT/conv16x_yolox24.txt" 2>&1
  timeout 300 tools/micro/conv16_probe 1 f16 -1,0 "" yolox > "$OUT/conv16x_yolox1.txt" 2>&1
  timeout 600 tools/micro/conv32_probe 2211 hrnet 5 > "$OUT/conv_f32_hrnet2211.txt" 2>&1
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_split" -- python "$R/tools/probe_split_leg.py" 1 6 > "$OUT/prof_split.txt" 2>&1; find "$OUT/prof_split" -name '*kernel_trace.csv' -delete )
  timeout 600 python tests/perf/bench_trackers.py 120 64 > "$OUT/trackers.log" 2>&1
  timeout 900 python -m pytest tests/test_gpu_precision.py -q -s -m gpu -k "envelope or saturated" > "$OUT/precision_envelope.txt" 2>&1
  timeout 600 python tools/probe_overlap2.py 0 1 2 3 > "$OUT/overlap2.txt" 2>&1
fi
for f in "$OUT"/bench_*.json "$OUT"/prof_*.json; do python - "$f" <<'PY'
import json, sys, os
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(os.path.basename(sys.argv[1]), d.get("dtype"), round(d["value"], 1), "frames/s", round(d["ms_per_step"], 1), "ms", "ids==oracle:", (d.get("parity") or {}).get("track_ids_equal_oracle"),
          "| f16", d.get("value_f16"), "split", d.get("value_f32_split"), "roofline", round((d.get("roofline") or {}).get("frac") or 0, 3))
except Exception as e:
    print(os.path.basename(sys.argv[1]), "FAILED", e)
PY
done
