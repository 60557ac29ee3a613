set -x
mkdir -p gpurun_out/r06h
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -k "patch" 2>&1 | tail -4
timeout 600 tools/micro/conv32_probe 2211 hrnet 5 2>&1 | grep -E "64>64|64 @" > gpurun_out/r06h/patch64_probe.txt; cat gpurun_out/r06h/patch64_probe.txt
timeout 900 python -m pytest tests/test_gpu_zz_stage_overlap.py tests/test_gpu_heads.py -m gpu -q -s 2>&1 | grep -E "passed|failed|FAILED|^   \(|one frame" | tail
