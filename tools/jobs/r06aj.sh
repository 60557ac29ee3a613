set -x
mkdir -p gpurun_out/r06aj
export TMPDIR=/tmp
( time timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > gpurun_out/r06aj/gpu_tests.txt 2>&1
cat gpurun_out/r06aj/gpu_tests.txt
