set -x
mkdir -p gpurun_out/r06e
export TMPDIR=/tmp
for cfg in "1 1" "0 1" "0 0"; do set -- $cfg; TLK_SPLIT_SCALES=$2 timeout 600 python tools/probe_split_leg.py $1 8 2>&1 | grep "split leg" | tee -a gpurun_out/r06e/summary.txt; done
timeout 300 tools/micro/conv16_probe 2400 split -1,0 > gpurun_out/r06e/split_probe.txt 2>&1; tail -2 gpurun_out/r06e/split_probe.txt
timeout 900 python -m pytest tests/test_gpu_split_scales.py tests/test_gpu_zz_stage_overlap.py tests/test_gpu_heads.py tests/test_gpu_precision.py -m gpu -q > gpurun_out/r06e/gpu_tests.txt 2>&1; tail -4 gpurun_out/r06e/gpu_tests.txt
( time timeout 900 python bench.py > gpurun_out/r06e/bench_default.json 2> gpurun_out/r06e/bench_default.err ) 2>> gpurun_out/r06e/bench_time.txt
tail -3 gpurun_out/r06e/bench_default.err; cat gpurun_out/r06e/bench_time.txt
