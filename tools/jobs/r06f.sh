set -x
mkdir -p gpurun_out/r06f
export TMPDIR=/tmp
( timeout 600 python tools/probe_overlap2.py 0 1 2 3 5; timeout 600 python tools/probe_overlap2.py --dist 0 1 2 3 ) 2>&1 | grep -E "extra streams|nccl group" | tee gpurun_out/r06f/overlap2.txt
# A/B of the split kernels: r05 library against this tree's, same box
( echo "== r05 library"; LD_LIBRARY_PATH=$PWD/tools/micro/oldlib timeout 300 tools/micro/conv16_probe 2400 split -1,0 | tail -1
  echo "== this tree"; timeout 300 tools/micro/conv16_probe 2400 split -1,0 | tail -1
  echo "== r05 library"; LD_LIBRARY_PATH=$PWD/tools/micro/oldlib timeout 300 tools/micro/conv16_probe 2400 split -1,0 | tail -1
  echo "== this tree"; timeout 300 tools/micro/conv16_probe 2400 split -1,0 | tail -1 ) 2>&1 | tee gpurun_out/r06f/split_ab.txt
timeout 300 tools/micro/conv16_probe 2400 split -1,0,4 "l1" 2>&1 | tee gpurun_out/r06f/split_l1_cfg4.txt | tail -12
timeout 900 python -m pytest tests/test_gpu_conv16.py -m gpu -q -k "half_step or full_step" 2>&1 | tail -3
TLK_PIPE_OVERLAP=0 bash tools/prof_lat.sh r06f 2>&1 | tail -45
