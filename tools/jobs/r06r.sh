set -x
mkdir -p gpurun_out/r06r
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dwconv.py -m gpu -q -x 2>&1 | tail -5
PROBE_DW_SHAPES=1 PROBE_DW_VARIANTS=1 timeout 600 python tools/probe_dwconv.py > gpurun_out/r06r/dwconv_probe.txt 2>&1
head -30 gpurun_out/r06r/dwconv_probe.txt
SHORT="--no-cpu-baseline --no-latency-leg --no-f32-leg --no-live-traffic"
timeout 600 python bench.py --workload config4 --steps 6 --warmup 2 $SHORT --check-frames 96 > gpurun_out/r06r/bench_config4.json 2> gpurun_out/r06r/bench_config4.err
timeout 600 python bench.py --workload config4 --dtype f16 --steps 10 --warmup 3 $SHORT --check-frames 96 > gpurun_out/r06r/bench_config4_f16.json 2> gpurun_out/r06r/bench_config4_f16.err
python - <<'PY'
import json
for f in ('bench_config4','bench_config4_f16'):
    try:
        j=json.loads([l for l in open(f'gpurun_out/r06r/{f}.json') if l.startswith('{')][-1])
        print(f, round(j['value'],1), j.get('value_resident'), j.get('ms_per_step'))
    except Exception as e:
        print(f, 'failed', e)
PY
tail -3 gpurun_out/r06r/bench_config4.err
