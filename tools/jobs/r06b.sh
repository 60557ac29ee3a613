set -x
mkdir -p gpurun_out/r06b
export TMPDIR=/tmp
( for v in "" "--dist"; do echo "== $v"; timeout 300 python tools/probe_overlap.py -1 0 serial $v; done
  echo "== GPU_MAX_HW_QUEUES=8 --dist"; GPU_MAX_HW_QUEUES=8 timeout 300 python tools/probe_overlap.py -1 serial --dist
  echo "== TLK_HEADS=0 --dist"; TLK_HEADS=0 timeout 300 python tools/probe_overlap.py -1 serial --dist
) > gpurun_out/r06b/overlap_probe.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06b/overlap_probe.txt
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r06b/gpu_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06b/gpu_tests.txt
tail -30 gpurun_out/r06b/gpu_tests.txt
