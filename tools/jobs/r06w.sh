set -x
mkdir -p gpurun_out/r06w
export TMPDIR=/tmp
for spec in "yolox-m 24" "yolox-m 1" "reid 2211" "reid 100" "reid-hrnet32 2211" "rtmpose-m 2211" "yolox-l 24" "yolox-s 32"; do
  set -- $spec
  timeout 600 python tools/sweep_conv_f32.py $1 $2 > gpurun_out/r06w/sweep_$1_$2.txt 2>&1
  grep -c "<--" gpurun_out/r06w/sweep_$1_$2.txt; tail -1 gpurun_out/r06w/sweep_$1_$2.txt
done
timeout 1200 python -m pytest tests/test_gpu_conv.py -m gpu -q -x 2>&1 | tail -4
