"""GPU box: summarise the LAST `window_ms` of a rocprofv3 kernel trace (csv) -- the steady-state steps at the end of a bench run, without the
tuning / warm-up / graph-capture launches that dominate `--stats` of a short run.
usage: python tools/trace_tail_summary.py <dir with *_kernel_trace.csv> <window_ms> [top]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d, win = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
path = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)      # the newest run
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))))
# the window ends with the last CONVOLUTION launch of the trace, not with the trace: what follows the timed steps (the one-rank RCCL group's
# set-up, the table fetch, teardown fills and copies) is not a step -- the r06 summaries taken at the end of the trace held only those
conv_ends = [r[1] for r in rows if "conv_f32_mfma_kernel" in r[2] or "conv16x_kernel" in r[2] or "conv16_mfma_kernel" in r[2]]
end = max(conv_ends) if conv_ends else max(r[1] for r in rows)
rows = [r for r in rows if r[0] <= end]
t0 = end - int(win * 1e6)
sel = [r for r in rows if r[0] >= t0]
agg = defaultdict(lambda: [0, 0])
for s, e, n, q in sel:
    a = agg[n]
    a[0] += 1
    a[1] += e - s
busy = sum(v[1] for v in agg.values())
# union of busy intervals (kernels on different streams overlap)
iv = sorted((s, e) for s, e, _, _ in sel)
union, cs, ce = 0, None, None
for s, e in iv:
    if cs is None:
        cs, ce = s, e
    elif s <= ce:
        ce = max(ce, e)
    else:
        union += ce - cs
        cs, ce = s, e
if cs is not None:
    union += ce - cs
print(f"window {win:.0f} ms: {len(sel)} launches, sum of kernel durations {busy / 1e6:.1f} ms, GPU busy (union) {union / 1e6:.1f} ms = {union / (win * 1e6) * 100:.0f} % of the window")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{n[:110]:110s} {c:6d} {t / 1e6:8.2f} ms  avg {t / c / 1e3:8.1f} us  {t / busy * 100:5.1f}%")
