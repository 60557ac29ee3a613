"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into profiles/<name>.md + .json (top kernels by total time)."""
import json
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
tot = sum(r[2] for r in rows)
lines = ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
js = []
for name, calls, total, avg, pct in rows[:40]:
    short = name if len(name) < 110 else name[:107] + "..."
    lines.append(f"| `{short}` | {calls} | {total / 1e3:.3f} | {avg:.2f} | {pct:.2f} |")
    js.append({"kernel": name, "calls": calls, "total_ms": total / 1e3, "avg_us": avg, "pct": pct})
open(out + ".md", "w").write("\n".join(lines) + f"\n\ntotal kernel time: {tot / 1e3:.2f} ms\n")
json.dump(js, open(out + ".json", "w"), indent=1)
print("\n".join(lines[:24]))
