set -x
mkdir -p gpurun_out/r06q
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_zz_stage_overlap.py -m gpu -q -s 2>&1 | grep -E "passed|failed|FAILED|^   \(|Error" | tail
for i in 1 2 3; do ( time timeout 900 python bench.py > gpurun_out/r06q/bench_default_$i.json 2> gpurun_out/r06q/bench_default_$i.err ) 2>> gpurun_out/r06q/bench_time.txt; done
timeout 600 python bench.py --workload config2 --steps 6 --warmup 2 --no-cpu-baseline --no-latency-leg --no-f32-leg --no-live-traffic --check-frames 96 > gpurun_out/r06q/bench_config2.json 2>/dev/null
python - <<'PY'
import json
for f in ('bench_default_1','bench_default_2','bench_default_3','bench_config2'):
    j=json.loads([l for l in open(f'gpurun_out/r06q/{f}.json') if l.startswith('{')][-1])
    print(f, round(j['value'],1), round(j['value_resident'],1), j.get('value_f16'), j.get('value_f32_split'), j.get('value_hrnet32'))
    if j.get('latency_f16'): print('  lat16', [(l['n_streams'],l['frames_per_step'],round(l['fps'],1)) for l in j['latency_f16']], 'auto', [(l['n_streams'],l['frames_per_step'],round(l['fps'],1),(l.get('overlap_trial') or {}).get('chosen','')[:12]) for l in j['latency_f16_overlap']], 'fp32', [(l['n_streams'],l['frames_per_step'],round(l['fps'],1)) for l in j['latency']])
PY
