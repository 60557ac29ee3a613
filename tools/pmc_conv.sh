#!/usr/bin/env bash
# ON THE GPU BOX: PMC passes over tools/probe_conv_pmc.py (counters only with --kernel-trace, never with sys-trace)
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-pmc_conv}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/p1" -- python "$R/tools/probe_conv_pmc.py" > "$OUT/p1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_MFMA --output-format csv -d "$OUT/p2" -- python "$R/tools/probe_conv_pmc.py" > "$OUT/p2.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "conv" not in n: continue
        key = "split" if "Li1ELb0EEE" in n or ", 1, false>" in n and "conv16" in n else ("f16" if "conv16" in n else "f32")
        agg[n[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in agg.items():
    print(n)
    for c, v in sorted(d.items()): print(f"   {c:28s} {sum(v)/len(v):16.0f}")
PY
