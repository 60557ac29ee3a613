set -x
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/make_profiles_r06.sh "bench trace pmc probes" 2>&1 | tail -25
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/round6/gpu_tests_final.txt 2>&1; echo "pytest rc $?" >> gpurun_out/round6/gpu_tests_final.txt
tail -5 gpurun_out/round6/gpu_tests_final.txt
