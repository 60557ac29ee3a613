set -x
mkdir -p gpurun_out/r06i
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_zz_stage_overlap.py tests/test_gpu_conv.py tests/test_gpu_heads.py -m gpu -q -s 2>&1 | grep -E "passed|failed|FAILED|^   \(|one frame|Error" | tail -12
timeout 600 python bench.py --workload config3h --steps 6 --warmup 2 --no-cpu-baseline --no-latency-leg --no-f32-leg --no-live-traffic --check-frames 96 > gpurun_out/r06i/bench_config3h.json 2> gpurun_out/r06i/bench_config3h.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r06i/bench_config3h.json') if l.startswith('{')][-1])
print('config3h', j['value'], j['ms_per_step'], j['roofline']['frac'], j['parity']['track_ids_equal_oracle'])
for r in j['roofline']['per_instantiation'][:6]: print(r)
PY
( time timeout 900 python bench.py > gpurun_out/r06i/bench_default.json 2> gpurun_out/r06i/bench_default.err ) 2>> gpurun_out/r06i/bench_time.txt
tail -3 gpurun_out/r06i/bench_default.err; cat gpurun_out/r06i/bench_time.txt
