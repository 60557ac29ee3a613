set -x
mkdir -p gpurun_out/r06ab
export TMPDIR=/tmp
for spec in "f16 reid-hrnet32 2211" "f16 rtmpose-m 2211" "f16 yolox-m 24" "f16 reid 2211" "f16 yolox-m 1" "f16 reid 100"; do
  set -- $spec
  timeout 600 python tools/sweep_conv16.py $1 $2 $3 > gpurun_out/r06ab/sweep16_$1_$2_$3.txt 2>&1
  grep -c "<--" gpurun_out/r06ab/sweep16_$1_$2_$3.txt; tail -1 gpurun_out/r06ab/sweep16_$1_$2_$3.txt | cut -c1-300
done
timeout 1200 python -m pytest tests/test_gpu_conv16.py tests/test_gpu_precision.py -m gpu -q -x 2>&1 | tail -4
SHORT="--no-cpu-baseline --no-latency-leg --no-f32-leg --no-live-traffic"
for wl in config3h config4 config3; do
timeout 600 python bench.py --workload $wl --dtype f16 --steps 10 --warmup 3 $SHORT --check-frames 96 > gpurun_out/r06ab/bench_${wl}_f16.json 2>/dev/null
python -c "
import json
j=json.loads([l for l in open('gpurun_out/r06ab/bench_${wl}_f16.json') if l.startswith('{')][-1]); print('$wl', round(j['value'],1), j.get('value_resident'))"
done
