#!/usr/bin/env bash
# GPU box, second part of the r05 profile set: per-layer tables of the convolution kernels (standalone probes), the f16 route A/B, the fused
# f16 stem against the library route, the association kernels alone.  Output: gpurun_out/round5/.
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/round5"; mkdir -p "$OUT"
cd "$R"
P16=tools/micro/conv16_probe; P32=tools/micro/conv32_probe
timeout 120 $P16 2400 f16 -1,0 > "$OUT/conv16x_reid2400.txt" 2>&1
timeout 120 $P16 100 f16 -1,0 > "$OUT/conv16x_reid100.txt" 2>&1
timeout 120 $P16 24 f16 -1,0 "" yolox > "$OUT/conv16x_yolox24.txt" 2>&1
timeout 120 $P16 1 f16 -1,0 "" yolox > "$OUT/conv16x_yolox1.txt" 2>&1
timeout 120 $P16 2400 split -1,0 > "$OUT/conv16x_split2400.txt" 2>&1
timeout 120 $P32 2400 all > "$OUT/conv_f32_resnet2400.txt" 2>&1
timeout 120 $P32 2211 hrnet > "$OUT/conv_f32_hrnet2211.txt" 2>&1
timeout 120 python tools/probe_stem16.py 2400 2>&1 | grep -v amdgpu.ids > "$OUT/stem16.txt"; timeout 60 python tools/probe_stem16.py 100 2>&1 | grep -v amdgpu.ids >> "$OUT/stem16.txt"
timeout 600 bash tools/r05_ab.sh f16 > "$OUT/ab_f16.txt" 2>&1
timeout 300 python tests/perf/bench_trackers.py 120 64 > "$OUT/trackers.log" 2>&1; cp gpurun_out/trackers.json "$OUT/trackers.json" 2>/dev/null
timeout 300 python -m pytest tests/test_gpu_precision.py -q -s -m gpu -k "envelope or saturated" 2>&1 | grep -v amdgpu.ids > "$OUT/precision_envelope.txt"
tail -4 "$OUT/ab_f16.txt"; tail -3 "$OUT/stem16.txt"
