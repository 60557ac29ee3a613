"""HERE, after tools/make_profiles_r06.sh ran on the GPU box: copy what is judged from gpurun_out/round6 into profiles/ (r06_*)."""
import glob
import json
import os
import shutil
import subprocess
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/round6"
dst = "profiles"


def last_json(path):
    lines = [l for l in open(path).read().strip().splitlines() if l.startswith("{")]
    return lines[-1] if lines else None


for path in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
    name = os.path.basename(path)[len("bench_"):-len(".json")]
    out = "r06_bench_default_driver_form.json" if name == "default" else f"r06_bench_{name}.json"
    line = last_json(path)
    if line:
        json.loads(line)
        open(os.path.join(dst, out), "w").write(line + "\n")
SHORT = "--no-cpu-baseline --no-latency-leg --no-f32-leg --no-live-traffic --no-h2d-leg --check-frames 0"
for tag, out, cmd in (("prof_f32", "r06_config3_f32_rocprof", f"python bench.py --workload config3 --steps 10 --warmup 3 {SHORT}"),
                      ("prof_f16", "r06_config3_f16_rocprof", f"python bench.py --workload config3 --dtype f16 --steps 12 --warmup 3 {SHORT}"),
                      ("prof_f16_lat", "r06_latency1_f16_rocprof", f"python bench.py --workload config3 --dtype f16 --frames-per-step 1 --steps 150 --warmup 10 {SHORT}")):
    if glob.glob(os.path.join(src, tag, "**", "*kernel_stats.csv"), recursive=True):
        subprocess.run([sys.executable, "tools/rocprof_csv_md.py", os.path.join(src, tag), os.path.join(dst, out),
                        "rocprofv3 --kernel-trace --stats --output-format csv -- " + cmd, os.path.join(src, tag + ".json")], check=True, stdout=subprocess.DEVNULL)
if glob.glob(os.path.join(src, "prof_split", "**", "*kernel_stats.csv"), recursive=True):
    subprocess.run([sys.executable, "tools/rocprof_csv_md.py", os.path.join(src, "prof_split"), os.path.join(dst, "r06_split_leg_rocprof"),
                    "rocprofv3 --kernel-trace --stats --output-format csv -- python tools/probe_split_leg.py 1 6"], check=False, stdout=subprocess.DEVNULL)
for name, out, head in (
        ("tail_f32.txt", "r06_config3_f32_steady_state.txt", "tools/trace_tail_summary.py over the last 1200 ms of the fp32 kernel trace above (steady-state steps only: no warm-up, no graph capture)"),
        ("tail_f16.txt", "r06_config3_f16_steady_state.txt", "tools/trace_tail_summary.py over the last 400 ms of the f16 kernel trace (24 frames per step; steady-state steps only)"),
        ("tail_f16_lat.txt", "r06_latency1_f16_kernels.txt", "tools/trace_tail_summary.py over the last 400 ms of the one-frame-per-step f16 kernel trace (what the online step spends its time on)"),
        ("conv16x_reid2400.txt", "r06_conv16x_shapes_reid2400.txt", "tools/micro/conv16_probe 2400 f16 -1,0 -- every convolution shape of the ReID ResNet-50 at 2400 crops: cfg -1 = the r04 kernels, cfg 0 = what tlk_conv2d_nhwc_16 launches now (heuristic over csrc/tlk_conv16x.hip)"),
        ("conv16x_reid100.txt", "r06_conv16x_shapes_reid100.txt", "tools/micro/conv16_probe 100 f16 -1,0 -- the same at 100 crops (one frame: the online step)"),
        ("conv16x_yolox24.txt", "r06_conv16x_shapes_yolox24.txt", "tools/micro/conv16_probe 24 f16 -1,0 '' yolox -- YOLOX-m's convolution shapes at 24 frames"),
        ("conv16x_yolox1.txt", "r06_conv16x_shapes_yolox1.txt", "tools/micro/conv16_probe 1 f16 -1,0 '' yolox -- YOLOX-m's convolution shapes at one frame"),
        ("conv16x_split2400.txt", "r06_conv16x_shapes_split2400.txt", "tools/micro/conv16_probe 2400 split -1,0 -- split-precision mode, ReID ResNet-50 at 2400 crops"),
        ("conv_f32_resnet2400.txt", "r06_conv_f32_shapes_resnet2400.txt", "tools/micro/conv32_probe 2400 all -- fp32: the direct RGB stem kernel vs the padded implicit GEMM; the 1 x 1 layers of ResNet-50 under every tile configuration (21..33 = the direct-to-LDS kernels of tlk_conv16x.hip on fp32 tensors); 'bits differing' = against configuration 0"),
        ("conv_f32_hrnet2211.txt", "r06_conv_f32_shapes_hrnet2211.txt", "tools/micro/conv32_probe 2211 hrnet -- fp32: HRNet-W32's layer shapes (2211 crops = one config-3h step) under every tile configuration; 30..33 = the patch-resident 3 x 3 kernel"),
        ("stem16.txt", "r06_stem16.txt", "tools/probe_stem16.py -- the f16 RGB stem (7 x 7 / 2 + bias + ReLU + 3 x 3 / 2 max-pool) of the ReID network: fused kernel vs the library route it replaces"),
        ("prof_split.txt", "r06_split_leg.txt", "tools/probe_split_leg.py 1 6 under rocprofv3 --kernel-trace --stats -- the split-precision leg alone (both networks in split mode, scaled planes): its own frames/s line; per-kernel table in r06_split_leg_rocprof.md"),
        ("overlap2.txt", "r06_overlap_probe2.txt", "tools/probe_overlap2.py 0 1 2 3 -- one-frame f16 chain, serial vs overlap_stages=True as further idle streams are created in the process (see r06_overlap_autotune.md)"),
        ("precision_envelope.txt", "r06_precision_envelope.txt", "python -m pytest tests/test_gpu_precision.py -q -s -m gpu -k 'envelope or saturated' -- the f16 / split-precision legs against the exact-fp32 run as the backbone's largest activation is scaled towards and past float16's range"),
        ("ab_f16.txt", "r06_f16_route_ab.txt", "tools/r06_ab.sh f16 -- config 3 end to end, library route (MIOpen / CK / hipBLASLt + tlk_bias_act; r01-r04 default) vs libtlk's own 16-bit kernels (r06 default), ids checked against the oracle chain in every run")):
    if os.path.exists(os.path.join(src, name)):
        open(os.path.join(dst, out), "w").write(head + "\n" + open(os.path.join(src, name)).read())
# HBM traffic of the fp32 convolutions
p = os.path.join(src, "pmc_conv_f32.json")
if os.path.exists(p):
    tr = json.load(open(p))
    tr["workload"] = "config3"
    bl = last_json(os.path.join(src, "bench_default.json")) if os.path.exists(os.path.join(src, "bench_default.json")) else None
    alg = (json.loads(bl).get("roofline") or {}).get("algorithmic_bytes_per_launch") if bl else None
    tr["algorithmic_bytes_per_conv_launch"] = alg
    tr["traffic_over_algorithmic_bytes"] = tr["mean_traffic_bytes_per_conv_launch"] / alg if alg else None
    json.dump(tr, open(os.path.join(dst, "r06_conv_f32_traffic.json"), "w"), indent=1)
    md = ["# r06 -- HBM traffic of the fp32 convolution launches of one config-3 step (PMC)", "",
          "`rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in SEPARATE passes (tools/make_profiles_r06.sh pmc) over",
          "`python bench.py --workload config3 --steps 2 --warmup 1 --no-cpu-baseline --no-latency-leg --no-f32-leg --no-live-traffic --no-h2d-leg --check-frames 0`;",
          "FETCH_SIZE (KB) x 1024 x 2 (MI355X_MICROARCH.md: gfx950 reports half the bytes of a wide coalesced read), WRITE_SIZE (KB) x 1024.", "",
          f"mean over all convolution launches of the run: **{tr['mean_traffic_bytes_per_conv_launch'] / 1e6:.1f} MB per launch**"
          + (f"; algorithmic bytes (input + output (+ residual) + weights once, live crops only) {alg / 1e6:.1f} MB per launch: traffic / algorithmic = "
             f"**{tr['traffic_over_algorithmic_bytes']:.2f}**" if alg else ""), "",
          "| kernel instantiation | launches (fetch / write pass) | fetch MB / launch | write MB / launch |", "|---|---|---|---|"]
    for r in tr["per_instantiation"]:
        md.append(f"| `{r['kernel']}` | {r['launches_fetch_pass']} / {r['launches_write_pass']} | {r['fetch_bytes_per_launch'] / 1e6:.1f} | {r['write_bytes_per_launch'] / 1e6:.1f} |")
    open(os.path.join(dst, "r06_conv_f32_traffic.md"), "w").write("\n".join(md) + "\n")
if os.path.exists(os.path.join(src, "trackers.log")):
    import ast
    rows = [ast.literal_eval(l) for l in open(os.path.join(src, "trackers.log")) if l.startswith("{")]
    md = ["# r06 -- association kernels alone (tests/perf/bench_trackers.py 120 64): one 100-object stream and 64 streams, 120 frames each", "",
          "| tracker | streams | us per frame and launch | frames/s | C oracle frames/s (1 thread) |", "|---|---|---|---|---|"]
    for r in rows:
        md.append(f"| {r['tracker']} | {r['streams']} | {r['gpu_us_per_frame_per_launch']:.1f} | {r['gpu_frames_per_s']:.0f} | {r.get('cpu_oracle_frames_per_s', float('nan')):.1f} |")
    open(os.path.join(dst, "r06_trackers.md"), "w").write("\n".join(md).replace("| nan |", "| - |") + "\n")
print(sorted(f for f in os.listdir(dst) if f.startswith("r06_")))
