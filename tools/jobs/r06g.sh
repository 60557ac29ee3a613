set -x
mkdir -p gpurun_out/r06g
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -s > gpurun_out/r06g/gpu_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06g/gpu_tests.txt
grep -E "passed|failed|FAILED|Error|one frame per step|^   \(" gpurun_out/r06g/gpu_tests.txt | tail -30
( echo "== r05 library"; LD_LIBRARY_PATH=$PWD/tools/micro/oldlib timeout 300 tools/micro/conv16_probe 2400 split -1,0 | tail -1
  echo "== this tree"; timeout 300 tools/micro/conv16_probe 2400 split -1,0 | tail -1 ) 2>&1 | tee gpurun_out/r06g/split_ab.txt
timeout 300 python tools/probe_split_leg.py 1 8 2>&1 | grep "split leg"
( time timeout 900 python bench.py > gpurun_out/r06g/bench_default.json 2> gpurun_out/r06g/bench_default.err ) 2>> gpurun_out/r06g/bench_time.txt
tail -3 gpurun_out/r06g/bench_default.err; cat gpurun_out/r06g/bench_time.txt
