set -x
export TMPDIR=/tmp
timeout 600 python tools/probe_determinism.py rtmpose-t 400 f16 2>&1 | tail -5
timeout 600 python tools/probe_determinism.py rtmpose-m 200 f16 2>&1 | tail -5
timeout 600 python tools/probe_determinism.py rtmpose-t 200 f32 2>&1 | tail -5
for i in 1 2 3 4 5 6; do timeout 600 python -m pytest tests/test_gpu_modules.py -m gpu -q -x -k weight_derived 2>&1 | tail -1; done
