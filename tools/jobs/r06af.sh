set -x
export TMPDIR=/tmp
rm -rf gpurun_out/round6/prof_split
bash tools/make_profiles_r06.sh "pmc probes" > gpurun_out/make_profiles_pmc_probes.log 2>&1
tail -5 gpurun_out/make_profiles_pmc_probes.log | cut -c1-300
