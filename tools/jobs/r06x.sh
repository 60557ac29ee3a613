set -x
mkdir -p gpurun_out/r06x
export TMPDIR=/tmp
for i in 1 2; do timeout 900 python bench.py > gpurun_out/r06x/bench_default_$i.json 2> gpurun_out/r06x/bench_default_$i.err; done
python - <<'PY'
import json
for f in ('bench_default_1','bench_default_2'):
    j=json.loads([l for l in open(f'gpurun_out/r06x/{f}.json') if l.startswith('{')][-1])
    print(f, round(j['value'],1), round(j['value_resident'],1), j.get('value_f16'), j.get('value_f32_split'), j.get('value_hrnet32'), j['roofline']['frac'])
    print('  fp32 lat', [(l['n_streams'],l['frames_per_step'],round(l['fps'],1)) for l in j['latency']])
    print('  per inst', [(p['kernel'][:48], p['launches_per_step'], round(p['avg_launch_ms'],3), round(p['tflops'],1)) for p in j['roofline']['per_instantiation'][:8]])
PY
