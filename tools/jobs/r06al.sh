set -x
mkdir -p gpurun_out/r06al
export TMPDIR=/tmp
for i in 1 2 3 4 5; do timeout 900 python -m pytest tests/test_gpu_zz_stage_overlap.py tests/test_gpu_under_load.py -m gpu -q -x 2>&1 | tail -1; done
( time timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 ) > gpurun_out/r06al/gpu_tests.txt 2>&1
cat gpurun_out/r06al/gpu_tests.txt
