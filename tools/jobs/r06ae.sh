set -x
mkdir -p gpurun_out/r06ae
export TMPDIR=/tmp
timeout 600 python tools/fuzz_gpu_conv.py 300 7 > gpurun_out/r06ae/fuzz_conv.txt 2>&1
tail -3 gpurun_out/r06ae/fuzz_conv.txt | cut -c1-400
timeout 900 python tools/fuzz_gpu.py > gpurun_out/r06ae/fuzz_gpu.txt 2>&1
tail -8 gpurun_out/r06ae/fuzz_gpu.txt | cut -c1-300
