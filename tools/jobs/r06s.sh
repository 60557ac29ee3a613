set -x
mkdir -p gpurun_out/r06s
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_dwconv.py tests/test_gpu_spp.py tests/test_gpu_pose.py tests/test_gpu_modules.py -m gpu -q -x 2>&1 | tail -5
python tools/probe_dwconv.py > /dev/null 2>&1
PROBE_DW_SHAPES=1 PROBE_DW_VARIANTS=1 timeout 600 python tools/probe_dwconv.py > gpurun_out/r06s/dwconv_probe.txt 2>&1
head -12 gpurun_out/r06s/dwconv_probe.txt; grep -A8 "^SPP" gpurun_out/r06s/dwconv_probe.txt
SHORT="--no-cpu-baseline --no-latency-leg --no-f32-leg --no-live-traffic"
timeout 600 python bench.py --workload config4 --dtype f16 --steps 10 --warmup 3 $SHORT --check-frames 96 > gpurun_out/r06s/bench_config4_f16.json 2> gpurun_out/r06s/bench_config4_f16.err
python - <<'PY'
import json
for f in ('bench_config4_f16',):
    j=json.loads([l for l in open(f'gpurun_out/r06s/{f}.json') if l.startswith('{')][-1])
    print(f, round(j['value'],1), j.get('value_resident'), j.get('ms_per_step'))
PY
