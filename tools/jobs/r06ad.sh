set -x
mkdir -p gpurun_out/r06ad
export TMPDIR=/tmp
( time timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > gpurun_out/r06ad/gpu_tests.txt 2>&1
cat gpurun_out/r06ad/gpu_tests.txt
bash tools/make_profiles_r06.sh "bench trace" > gpurun_out/r06ad/make_profiles.log 2>&1
tail -3 gpurun_out/r06ad/make_profiles.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/round6/bench_*.json')):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], round(j['value'],1), j.get('value_resident'), j.get('value_f16'), j.get('value_f32_split'), j.get('value_hrnet32'), (j.get('roofline') or {}).get('frac'))
    except Exception as e:
        print(f, 'failed', e)
PY
