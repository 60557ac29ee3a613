set -x
mkdir -p gpurun_out/r06y
export TMPDIR=/tmp
for spec in "f16 reid 2211" "split reid 2211" "f16 yolox-m 24" "f16 reid 100" "f16 yolox-m 1" "split yolox-m 24"; do
  set -- $spec
  timeout 500 python tools/sweep_conv16.py $1 $2 $3 > gpurun_out/r06y/sweep16_$1_$2_$3.txt 2>&1
  grep -c "<--" gpurun_out/r06y/sweep16_$1_$2_$3.txt; tail -2 gpurun_out/r06y/sweep16_$1_$2_$3.txt | cut -c1-300
done
