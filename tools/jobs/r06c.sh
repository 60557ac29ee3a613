set -x
mkdir -p gpurun_out/r06c
export TMPDIR=/tmp
( for v in "" "--dist"; do echo "== $v"; timeout 300 python tools/probe_overlap.py -1 serial $v; done ) > gpurun_out/r06c/overlap_probe.txt 2>&1
grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|destroy_process" gpurun_out/r06c/overlap_probe.txt
timeout 1800 python -m pytest tests/test_gpu_heads.py tests/test_gpu_split_scales.py tests/test_gpu_precision.py tests/test_gpu_conv16.py tests/test_gpu_zz_stage_overlap.py tests/test_gpu_pipeline_configs.py tests/test_gpu_spp.py tests/test_gpu_bench_launch.py -m gpu -q -s > gpurun_out/r06c/gpu_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06c/gpu_tests.txt
grep -E "passed|failed|FAILED|Error|precision envelope|frames/s" gpurun_out/r06c/gpu_tests.txt | tail -30
timeout 300 tools/micro/conv16_probe 2400 split -1,0 > gpurun_out/r06c/split_probe.txt 2>&1; tail -3 gpurun_out/r06c/split_probe.txt
( time timeout 900 python bench.py > gpurun_out/r06c/bench_default.json 2> gpurun_out/r06c/bench_default.err ) 2>> gpurun_out/r06c/bench_time.txt
tail -3 gpurun_out/r06c/bench_default.err; cat gpurun_out/r06c/bench_time.txt
