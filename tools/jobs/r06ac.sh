set -x
mkdir -p gpurun_out/r06ac
export TMPDIR=/tmp
timeout 900 python tools/sweep_conv_f32.py reid 2211 > gpurun_out/r06ac/sweep_reid_2211.txt 2>&1
grep -E "38:|39:" gpurun_out/r06ac/sweep_reid_2211.txt | cut -c1-220
tail -1 gpurun_out/r06ac/sweep_reid_2211.txt
timeout 600 python tools/sweep_conv_f32.py yolox-m 24 > gpurun_out/r06ac/sweep_yolox-m_24.txt 2>&1
grep -E "38:|39:" gpurun_out/r06ac/sweep_yolox-m_24.txt | cut -c1-220
tail -1 gpurun_out/r06ac/sweep_yolox-m_24.txt
