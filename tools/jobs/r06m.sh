set -x
mkdir -p gpurun_out/r06m
export TMPDIR=/tmp
for i in 1 2; do ( time timeout 900 python bench.py > gpurun_out/r06m/bench_default_$i.json 2> gpurun_out/r06m/bench_default_$i.err ) 2>> gpurun_out/r06m/bench_time.txt; done
cat gpurun_out/r06m/bench_time.txt
timeout 600 python bench.py --workload config2 --steps 6 --warmup 2 --no-cpu-baseline --no-latency-leg --no-f32-leg --no-live-traffic --check-frames 96 > gpurun_out/r06m/bench_config2.json 2>/dev/null
timeout 300 tools/micro/conv16_probe 2400 f16 auto "+res" > gpurun_out/r06m/f16_res_auto.txt 2>&1
python - <<'PY'
import json
for f in ('bench_default_1','bench_default_2','bench_config2'):
    j=json.loads([l for l in open(f'gpurun_out/r06m/{f}.json') if l.startswith('{')][-1])
    print(f, round(j['value'],1), round(j['value_resident'],1), j.get('value_f16'), j.get('value_f32_split'), j.get('value_hrnet32'), j['collectives'][:40], (j.get('hota_allreduce') or {}).get('equal_to_local'))
    if j.get('latency_f16'): print('  lat16', [(l['n_streams'],l['frames_per_step'],round(l['fps'],1)) for l in j['latency_f16']], 'auto', [(l['n_streams'],l['frames_per_step'],round(l['fps'],1),(l.get('overlap_trial') or {}).get('chosen','')[:12]) for l in j['latency_f16_overlap']], 'fp32', [(l['n_streams'],l['frames_per_step'],round(l['fps'],1)) for l in j['latency']])
PY
