"""HERE, after tools/make_profiles_r04.sh ran on the GPU box: copy what is judged from gpurun_out/round4 into profiles/ (r04_*)."""
import glob
import json
import os
import shutil
import subprocess
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/round4"
dst = "profiles"
for path in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
    name = os.path.basename(path)[len("bench_"):-len(".json")]
    out = "r04_bench_default_driver_form.json" if name == "default" else f"r04_bench_{name}.json"
    line = open(path).read().strip().splitlines()[-1]
    json.loads(line)
    open(os.path.join(dst, out), "w").write(line + "\n")
cmd = ("rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --workload config3 --steps 10 --warmup 3 --no-cpu-baseline "
       "--no-latency-leg --no-f32-leg --no-live-traffic --check-frames 0")
subprocess.run([sys.executable, "tools/rocprof_csv_md.py", os.path.join(src, "prof_f32"), os.path.join(dst, "r04_config3_f32_rocprof"), cmd,
                os.path.join(src, "prof_f32.json")], check=True, stdout=subprocess.DEVNULL)
import ast
rows = [ast.literal_eval(l) for l in open(os.path.join(src, "trackers.log")) if l.startswith("{")]
md = ["# r04 -- association kernels alone (tests/perf/bench_trackers.py 120 64): one 100-object stream and 64 streams, 120 frames each", "",
      "| tracker | streams | us per frame and launch | frames/s | C oracle frames/s (1 thread) |", "|---|---|---|---|---|"]
for r in rows:
    md.append(f"| {r['tracker']} | {r['streams']} | {r['gpu_us_per_frame_per_launch']:.1f} | {r['gpu_frames_per_s']:.0f} | {r.get('cpu_oracle_frames_per_s', float('nan')):.1f} |")
open(os.path.join(dst, "r04_trackers.md"), "w").write("\n".join(md).replace("| nan |", "| – |") + "\n")
shutil.copy(os.path.join(src, "decode_nms.txt"), os.path.join(dst, "r04_decode_nms.txt"))
for name, out, head in (("pose_f16.txt", "r04_pose_forward_f16.txt", "python tools/probe_rtmpose.py 2400 f16 3 -- RTMPose-m forward of config 4 alone (2400 crops, f16), events around the modules (eager)"),
                        ("fuzz_gpu.txt", "r04_fuzz_gpu.txt", "python tools/fuzz_gpu.py 300; FUZZ_MAX_OBJECTS=320 FUZZ_BIG_CAPACITY=1 python tools/fuzz_gpu.py 20 ocsort bytetrack botsort deepocsort -- "
                                                              "every tracker bank vs the C oracle on random hyper-parameters x random crowded streams (second command: banks of 4096 x 512, scenes beyond the LDS tiers)")):
    if os.path.exists(os.path.join(src, name)):
        open(os.path.join(dst, out), "w").write(head + "\n" + open(os.path.join(src, name)).read())
print(open(os.path.join(dst, "r04_config3_f32_rocprof.md")).read()[:3000])
