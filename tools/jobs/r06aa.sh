set -x
mkdir -p gpurun_out/r06aa
export TMPDIR=/tmp
for spec in "f16 reid-hrnet32 2211" "f16 rtmpose-m 2211" "f16 yolox-l 24" "f16 yolox-s 32"; do
  set -- $spec
  timeout 600 python tools/sweep_conv16.py $1 $2 $3 > gpurun_out/r06aa/sweep16_$1_$2_$3.txt 2>&1
  grep -c "<--" gpurun_out/r06aa/sweep16_$1_$2_$3.txt; tail -1 gpurun_out/r06aa/sweep16_$1_$2_$3.txt | cut -c1-300
done
