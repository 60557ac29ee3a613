set -x
mkdir -p gpurun_out/r06d
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
for cfg in "1 1" "0 1" "0 0"; do
  set -- $cfg
  TLK_SPLIT_SCALES=$2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06d/prof_$1_$2 -- python $R/tools/probe_split_leg.py $1 5 2>&1 | grep "split leg" | tee -a $R/gpurun_out/r06d/summary.txt
  f=$(find $R/gpurun_out/r06d/prof_$1_$2 -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY' | tee -a $R/gpurun_out/r06d/summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {int(r["Calls"]):6d} calls {float(r["AverageNs"])/1e3:9.1f} us  {100*float(r["TotalDurationNs"])/tot:5.1f}%  {r["Name"][:150]}')
PY
  find $R/gpurun_out/r06d/prof_$1_$2 -name "*.csv" ! -name "*kernel_stats.csv" -delete
done
