set -x
mkdir -p gpurun_out/r06ah
export TMPDIR=/tmp
( time timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > gpurun_out/r06ah/gpu_tests.txt 2>&1
cat gpurun_out/r06ah/gpu_tests.txt
python -c "
import __graft_entry__ as g
g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r06ah/bench_default.json 2> gpurun_out/r06ah/bench_default.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r06ah/bench_default.json') if l.startswith('{')][-1])
print(round(j['value'],1), round(j['value_resident'],1), j.get('value_f16'), j.get('value_f32_split'), j.get('value_hrnet32'), j['roofline']['frac'])
PY
