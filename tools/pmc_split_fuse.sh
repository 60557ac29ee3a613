#!/usr/bin/env bash
# Runs ON THE GPU BOX: HBM traffic of split_fuse_sum_kernel per launch -- FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (--kernel-trace only,
# never mixed with sys / hip tracing), over tools/probe_split_fuse.py.  gfx950 correction as in bench.py / the guide:
# bytes = (FETCH_SIZE x 2 + WRITE_SIZE) x 1024.  Output: gpurun_out/$1/summary.txt
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-pmc_split_fuse}"
rm -rf "$OUT"; mkdir -p "$OUT"
python "$R/tools/probe_split_fuse.py" 2>&1 | grep -v amdgpu.ids | tee "$OUT/probe.txt"
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/$ctr" -- python "$R/tools/probe_split_fuse.py" > "$OUT/$ctr.log" 2>&1
  find "$OUT/$ctr" -name '*kernel_trace.csv' -delete
done
python - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.OrderedDict()
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for path in glob.glob(f"{root}/{ctr}/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(path)) if "split_fuse_sum_kernel" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
        rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
        for i, row in enumerate(rows):                       # the probe launches each of its four shapes 9 times in a row (1 + 8)
            acc.setdefault(i // 9, collections.defaultdict(list))[ctr].append(float(row["Counter_Value"]))
print("shape (order of tools/probe_split_fuse.py) | launches | FETCH_SIZE KB | WRITE_SIZE KB | HBM bytes per launch = (FETCH x 2 + WRITE) x 1024")
for k, d in acc.items():
    f, w = d.get("FETCH_SIZE", [0]), d.get("WRITE_SIZE", [0])
    fm, wm = sum(f) / len(f), sum(w) / len(w)
    print(f"{k} | {len(f)} | {fm:.0f} | {wm:.0f} | {(fm * 2 + wm) * 1024 / 1e9:.3f} GB")
PY
