#!/usr/bin/env bash
# ON THE GPU BOX (r05): PMC passes over the standalone probes -- counters only with --kernel-trace (never with sys-trace), one counter
# group per pass.  Usage: tools/pmc_probes.sh OUTDIR
#   pass sq1 / sq2 : SQ counters of the large-tile f16 kernel on one compute-bound layer (l4 3x3 512, configuration 1)
#   pass fetch / write : FETCH_SIZE / WRITE_SIZE (+ L2 hit / miss) of the fp32 expansions with residual and of the direct stem kernel
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-pmc_probes}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P16="$R/tools/micro/conv16_probe"; P32="$R/tools/micro/conv32_probe"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/sq1" -- "$P16" 2400 f16 1,6 "l4 3x3" > "$OUT/sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_MFMA --output-format csv -d "$OUT/sq2" -- "$P16" 2400 f16 1,6 "l4 3x3" > "$OUT/sq2.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d "$OUT/fetch" -- "$P32" 2400 all 2 > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d "$OUT/write" -- "$P32" 2400 all 2 > "$OUT/write.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
grid = {}
for path in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "conv" not in n or "ref_conv" in n: continue
        key = (n[:110], r.get("Grid_Size", "?"))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as f:
    for (n, g), d in sorted(agg.items()):
        f.write(f"{n}  grid {g}\n")
        for c, v in sorted(d.items()):
            f.write(f"   {c:28s} mean {sum(v)/len(v):18.0f}   launches {len(v)}\n")
print(open(out + "/summary.txt").read())
PY
