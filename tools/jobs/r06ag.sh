set -x
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_conv16.py -m gpu -q -x 2>&1 | tail -6
