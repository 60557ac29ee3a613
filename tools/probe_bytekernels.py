"""HIP-event timing of the three byte-moving libtlk kernels on one bench launch's shape (the workloads of tools/probe_traffic.py):
crop (24 frames x ~98 crops -> 384x128 nhwc f16), pil (-> 256x128, Pillow semantics), letterbox (32 frames 1080p -> 640 focus f16).
python tools/probe_bytekernels.py crop|pil|letterbox ; kernel variants through TLK_CROP_WAVE / TLK_PIL_WAVE / TLK_LETTERBOX_WAVE."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]] + (sys.argv[1:] or ["crop"])
which = sys.argv[1]
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_traffic.py")).read()
g = {"__name__": "probe", "__file__": os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_traffic.py")}
exec(compile(src, "probe_traffic.py", "exec"), g)                   # builds the workload and warms the kernel up: fn(), frames, counts, ...
fn = g["fn"]
# PROBE_COLD=K: cycle through K copies of the frames, so that a launch reads source bytes that are not in L2 / MALL any more -- what the kernel sees
# inside the pipeline, where a detector forward has run since the frames were uploaded (the warm loop flatters the crop kernels by ~15 %)
cold = int(os.environ.get("PROBE_COLD", "0"))
clones = [g["frames"].clone() for _ in range(cold)] if cold else None
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for rep in range(5):
    e0.record()
    for it in range(20):
        if clones:
            g["frames"] = clones[it % cold]
        fn()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20 * 1e3)
us = float(np.median(ts))
if which == "letterbox":
    from tracklab_amd import roofline as rl
    alg = rl.letterbox_bytes(1080, 1920, 640, 360, 640, elem_bytes=2) * 32
else:
    b = g["boxes"]; cnt = g["counts"]
    oh = 384 if which == "crop" else 256
    src_px = sum(float(np.sum(np.round(b[i, :cnt[i], 2]) * np.round(b[i, :cnt[i], 3]))) for i in range(len(cnt)))
    alg = src_px * 3 + float(cnt.sum()) * 3 * oh * 128 * 2
env = {k: v for k, v in os.environ.items() if k.startswith("TLK_")}
print(f"{which} {env}: {us:.1f} us/launch (runs {[round(t, 1) for t in ts]}), {alg / 1e6:.1f} MB algorithmic -> {alg / us / 1e6:.2f} TB/s = {alg / us / 1e6 / 8:.3f} of 8 TB/s")
