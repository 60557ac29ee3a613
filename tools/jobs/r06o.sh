set -x
mkdir -p gpurun_out/r06o
export TMPDIR=/tmp
( timeout 600 python tools/probe_overlap2.py 0 1 2 3 5; timeout 600 python tools/probe_overlap2.py --dist 0 1 2 3 ) 2>&1 | grep -E "extra streams|nccl group|==" | tee gpurun_out/r06o/overlap2.txt
timeout 900 python -m pytest tests/test_gpu_zz_stage_overlap.py -m gpu -q -s 2>&1 | grep -E "passed|failed|FAILED|^   \(|Error" | tail
