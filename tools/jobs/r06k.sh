set -x
export TMPDIR=/tmp
bash tools/make_profiles_r06.sh "bench" 2>&1 | tail -14
