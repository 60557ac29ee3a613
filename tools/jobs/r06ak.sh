set -x
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_determinism.py -m gpu -q -x 2>&1 | tail -6
