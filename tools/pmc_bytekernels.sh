#!/usr/bin/env bash
# Runs ON THE GPU BOX: SQ counter passes (one counter group per pass, --kernel-trace only: never mixed with sys/hip tracing) over
# tools/probe_bytekernels.py for the three byte-moving kernels of r03, sources cold (PROBE_COLD=12, what the pipeline looks like) and, for the crop
# kernel, warm. Output: gpurun_out/$1/<kernel>_<cold|warm>_<pass>/...counter_collection.csv + summary.txt (mean per launch).
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-pmc_bytekernels}"
OUT="$R/gpurun_out/$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {   # kernel, cold count, name, counters...
  local k="$1" cold="$2" name="$3"; shift 3
  PROBE_COLD="$cold" rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python "$R/tools/probe_bytekernels.py" "$k" > "$OUT/$name.log" 2>&1
  find "$OUT/$name" -name '*kernel_trace.csv' -delete
}
for spec in "crop 12 cold" "crop 0 warm" "pil 12 cold" "letterbox 12 cold"; do
  set -- $spec
  pass "$1" "$2" "$1_$3_sq1" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
  pass "$1" "$2" "$1_$3_sq2" SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
done
python - "$OUT" <<'PY' > "$OUT/summary.txt"
import csv, glob, os, sys, collections
root = sys.argv[1]
for d in sorted(set(os.path.basename(p).rsplit("_", 1)[0] for p in glob.glob(root + "/*_sq1"))):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(f"{root}/{d}_sq*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"]
            if not any(s in k for s in ("crop_wave", "pil_wave", "letterbox_wave")):
                continue
            acc[k.split("(")[0][-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, c in acc.items():
        print(f"{d}: {k}")
        for name, v in sorted(c.items()):
            print(f"   {name:28s} mean={sum(v)/len(v):16.1f} n={len(v)}")
PY
cat "$OUT/summary.txt"
