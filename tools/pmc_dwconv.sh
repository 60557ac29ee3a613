#!/usr/bin/env bash
# Runs ON THE GPU BOX: HBM traffic of dwconv_kernel / spp_kernel per launch -- FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes
# (--kernel-trace only, never mixed with sys / hip tracing), over tools/probe_dwconv.py.  gfx950 correction as in bench.py / the guide:
# bytes = (FETCH_SIZE x 2 + WRITE_SIZE) x 1024.  Output: gpurun_out/$1/summary.txt
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-pmc_dwconv}"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/$ctr" -- python "$R/tools/probe_dwconv.py" > "$OUT/$ctr.log" 2>&1
  find "$OUT/$ctr" -name '*kernel_trace.csv' -delete
done
python - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for path in glob.glob(f"{root}/{ctr}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"]
            if ("dwconv_kernel" in k or "spp_kernel" in k) and row["Counter_Name"] == ctr:
                # one group per (kernel instantiation, grid): the probe launches each shape 12 times in a row
                acc[(k.split("(")[0][-70:], row.get("Grid_Size", "?"))][ctr].append(float(row["Counter_Value"]))
print("kernel | grid | launches | FETCH_SIZE KB | WRITE_SIZE KB | HBM bytes per launch = (FETCH x 2 + WRITE) x 1024")
for (k, g), d in acc.items():
    f, w = d.get("FETCH_SIZE", [0]), d.get("WRITE_SIZE", [0])
    fm, wm = sum(f) / len(f), sum(w) / len(w)
    print(f"{k} | {g} | {len(f)} | {fm:.0f} | {wm:.0f} | {(fm * 2 + wm) * 1024 / 1e6:.1f} MB")
PY
