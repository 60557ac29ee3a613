#!/usr/bin/env bash
# Runs ON THE GPU BOX (via gpurun): produces everything profiles/ is built from, under gpurun_out/round/.
#   bench lines (config3, config2), rocprofv3 kernel-trace stats of the same bench commands, PMC traffic passes of the
#   byte-moving kernels (FETCH_SIZE and WRITE_SIZE in separate passes, never mixed with sys-trace).
# Afterwards, HERE: python tools/collect_profiles.py gpurun_out/round rNN
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/round"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
python bench.py --workload config3 --steps 20 --warmup 5 > "$OUT/bench_config3.json" 2> "$OUT/bench_config3.err"
python bench.py --workload config3 --dtype f32 --steps 8 --warmup 3 --no-cpu-baseline --no-latency-leg --no-f32-leg --no-live-traffic --check-frames 96 > "$OUT/bench_config3_f32.json" 2> "$OUT/bench_config3_f32.err"
python bench.py --workload config2 --steps 20 --warmup 5 --no-f32-leg > "$OUT/bench_config2.json" 2> "$OUT/bench_config2.err"
for wl in config1 config4 config5 config3h config3s config3b config3c config3d config2b; do
  python bench.py --workload $wl --steps 8 --warmup 3 --no-cpu-baseline --no-latency-leg --no-f32-leg --no-live-traffic --check-frames 96 > "$OUT/bench_$wl.json" 2> "$OUT/bench_$wl.err"
done
cd /tmp && export TMPDIR=/tmp
for wl in config3 config2 config3s; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$wl" -- \
    python "$R/bench.py" --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-latency-leg --no-f32-leg --no-live-traffic --check-frames 0 > "$OUT/prof_$wl.json" 2> "$OUT/prof_$wl.err"
  find "$OUT/prof_$wl" -name '*kernel_trace.csv' -delete       # keep the stats, drop the raw trace (size)
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_probe" -- python "$R/tools/probe_kernels.py" > "$OUT/kt_probe.log" 2>&1
find "$OUT/kt_probe" -name '*kernel_trace.csv' -delete
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python "$R/tools/probe_kernels.py" > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python "$R/tools/probe_kernels.py" > "$OUT/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_mfma" -- python "$R/tools/probe_mfma.py" > "$OUT/kt_mfma.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d "$OUT/pmc_mfma" -- python "$R/tools/probe_mfma.py" > "$OUT/pmc_mfma.log" 2>&1
python "$R/tests/perf/bench_trackers.py" 120 64 > "$OUT/trackers.log" 2>&1; cp "$R/gpurun_out/trackers.json" "$OUT/trackers.json"
find "$OUT" -name '*kernel_trace.csv' -delete
echo "profiles raw data in $OUT"; ls "$OUT"
