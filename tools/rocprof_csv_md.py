"""rocprofv3 --kernel-trace --stats (csv) -> profiles/<name>.md + <name>_kernel_stats.csv, with the libtlk convolution kernel's template
instantiations also summed into one row (the bench line's roofline figure is over all of them).
    python tools/rocprof_csv_md.py <dir with *_kernel_stats.csv> profiles/r04_config3_f32_rocprof "<command>" [bench line json]"""
import csv
import glob
import json
import shutil
import sys

src, dst, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
import os
path = max(glob.glob(src + "/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)      # the newest run: gpurun merges every call's files into the same directory
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = [f"# {dst.split('/')[-1]}", "", f"command: `{cmd}`", ""]
if len(sys.argv) > 4:
    d = json.loads([l for l in open(sys.argv[4]).read().strip().splitlines() if l.startswith("{")][-1])      # (RCCL prints its banner to stdout when the process ends)
    out += [f"bench line of this profiled run: value = {d['value']:.1f} frames/s ({d['dtype']}), ms_per_step = {d['ms_per_step']:.2f}; "
            f"roofline (HIP events): {d['roofline']['achieved']:.1f} {d['roofline']['unit']} = {d['roofline']['frac']:.3f} of peak, "
            f"avg launch {d['roofline']['avg_launch_ms']:.4f} ms", ""]
CONV_NAMES = ("conv_f32_mfma_kernel", "conv16x_kernel", "conv_stem3_kernel", "conv16_mfma_kernel", "conv_stem16_kernel")
conv = [r for r in rows if any(k in r["Name"] for k in CONV_NAMES)]
if conv:
    c_calls = sum(int(r["Calls"]) for r in conv)
    c_ns = sum(float(r["TotalDurationNs"]) for r in conv)
    out += [f"**libtlk's convolution kernels (conv_f32_mfma_kernel, conv16x_kernel, conv_stem3_kernel, conv16_mfma_kernel, conv_stem16_kernel), all template instantiations together: {c_calls} calls, {c_ns / 1e6:.2f} ms, average {c_ns / c_calls / 1e3:.2f} us, "
            f"{100 * c_ns / tot:.2f} % of the GPU time** (includes the warm-up, parity and roofline passes of the command, whose launches are the same)", ""]
if len(sys.argv) > 4 and (d.get("roofline") or {}).get("per_instantiation"):
    out += ["Per instantiation, HIP events of the bench line (its roofline pass) beside rocprofv3's average over the whole command -- the ReLU / linear",
            "instantiations (5th template argument 1 / 0) are launched by the ReID network only, with the same mix of layers in every step, so the two",
            "averages are over the same set of shapes; the SiLU ones (2) also run in the detector's extra graph captures, whose small launches pull",
            "rocprofv3's average down:", "",
            "| instantiation | launches / step | bench: avg ms (events) | bench: TFLOP/s | rocprofv3: calls | rocprofv3: avg ms | events / rocprofv3 |", "|---|---|---|---|---|---|---|"]
    for e in d["roofline"]["per_instantiation"]:
        key = e["kernel"].split(" (")[0]                    # "conv_stem3_kernel (direct RGB stem, ...)" -> the kernel's name
        if key.endswith(">"):
            key = key[:-1]                                      # prefix match: rocprofv3 prints the defaulted trailing template arguments too (r06: ROWB, CPP)
        match = [r for r in rows if key in r["Name"]]
        if match:
            r = match[0]
            ravg = float(r["AverageNs"]) / 1e6
            out.append(f"| `{e['kernel']}` | {e['launches_per_step']} | {e['avg_launch_ms']:.4f} | {e['tflops']:.1f} | {r['Calls']} | {ravg:.4f} | {e['avg_launch_ms'] / ravg:.3f} |")
        else:
            out.append(f"| `{e['kernel']}` | {e['launches_per_step']} | {e['avg_launch_ms']:.4f} | {e['tflops']:.1f} | - | - | - |")
    out.append("")
out += ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
for r in rows[:45]:
    n = r["Name"]
    n = n if len(n) < 100 else n[:97] + "..."
    out.append(f"| `{n}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['MinNs']) / 1e3:.2f} | "
               f"{float(r['MaxNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |")
out.append(f"\ntotal kernel time: {tot / 1e6:.2f} ms over {len(rows)} kernels")
open(dst + ".md", "w").write("\n".join(out) + "\n")
shutil.copy(path, dst + "_kernel_stats.csv")
print("\n".join(out[:30]))
