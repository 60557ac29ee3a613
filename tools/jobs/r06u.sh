set -x
mkdir -p gpurun_out/r06u
export TMPDIR=/tmp
timeout 1200 python tools/sweep_conv_f32.py yolox-m 24 > gpurun_out/r06u/sweep_yolox_m_24.txt 2>&1
tail -60 gpurun_out/r06u/sweep_yolox_m_24.txt
timeout 1500 python tools/sweep_conv_f32.py reid 2211 > gpurun_out/r06u/sweep_reid_2211.txt 2>&1
tail -40 gpurun_out/r06u/sweep_reid_2211.txt
