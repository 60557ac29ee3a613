set -x
mkdir -p gpurun_out/r06l
export TMPDIR=/tmp
timeout 900 python tools/fuzz_gpu_conv.py 240 6 2>&1 | tail -3 | tee gpurun_out/r06l/fuzz_conv.txt
timeout 900 python tools/fuzz_gpu.py 120 2>&1 | tail -4 | tee gpurun_out/r06l/fuzz_gpu.txt
( time timeout 900 python bench.py > gpurun_out/r06l/bench_default.json 2> gpurun_out/r06l/bench_default.err ) 2>> gpurun_out/r06l/bench_time.txt; cat gpurun_out/r06l/bench_time.txt
