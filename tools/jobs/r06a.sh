set -x
mkdir -p gpurun_out/r06a
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06a/gpu_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06a/gpu_tests.txt
tail -5 gpurun_out/r06a/gpu_tests.txt
timeout 300 tools/micro/conv16_probe 2400 split -1,1,2,3,4,5,6,7 "+res" > gpurun_out/r06a/split_res_probe.txt 2>&1
tail -40 gpurun_out/r06a/split_res_probe.txt
( time timeout 900 python bench.py > gpurun_out/r06a/bench_default.json 2> gpurun_out/r06a/bench_default.err ) 2>> gpurun_out/r06a/bench_time.txt
tail -3 gpurun_out/r06a/bench_default.err; cat gpurun_out/r06a/bench_time.txt
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06a/bench_default.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','value_resident','value_f16','value_f32_split','value_hrnet32','collectives','warmup_serialized'): print(k, j.get(k))
print('roofline', {k:j['roofline'][k] for k in ('frac','achieved','conv_ms_per_step','algorithmic_tflop_per_step','exact_fp32_ceiling_frames_per_s')})
print('hbm', j['roofline_hbm']['frac'], j['roofline_hbm']['units_per_launch'])
print('lat', [(l['n_streams'],l['frames_per_step'],round(l['fps'],1)) for l in j['latency']])
print('lat16', [(l['n_streams'],l['frames_per_step'],round(l['fps'],1)) for l in j['latency_f16']])
print('lat16o', j['latency_f16_overlap'] if not isinstance(j['latency_f16_overlap'],list) else [(l['n_streams'],l['frames_per_step'],round(l['fps'],1)) for l in j['latency_f16_overlap']])
print('parity', j['parity']['track_ids_equal_oracle'], j['hrnet32_leg'])
PY
