#!/usr/bin/env bash
# Runs ON THE GPU BOX: SQ / TCC counter passes over tools/probe_crops.py (one counter group per pass, --kernel-trace only:
# never mixed with sys/hip tracing). Output: gpurun_out/$1/pmc_*/...counter_collection.csv + a per-kernel summary text.
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-pmc_crops}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {   # name, counters...
  local name="$1"; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python "$R/tools/probe_crops.py" > "$OUT/$name.log" 2>&1
  find "$OUT/$name" -name '*kernel_trace.csv' -delete
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
python - "$OUT" <<'PY' > "$OUT/summary.txt"
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if "crop" not in k:
            continue
        acc[k[:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} mean={sum(v)/len(v):16.1f} n={len(v)}")
PY
cat "$OUT/summary.txt"
