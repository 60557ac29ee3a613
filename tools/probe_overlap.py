"""GPU box: one-frame-per-step f16 chain with DetReidTrackPipeline(overlap_stages=True): frames/s with the detector stage's stream at high
priority (its own hardware queue) and at normal priority (may share the ReID stage's queue).  usage: python tools/probe_overlap.py [prio ...]"""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracklab_amd import gpu_pipeline as gp
from tracklab_amd.synth import SyntheticStream, render_frame, synth_yolox_head

args = [a for a in sys.argv[1:] if not a.startswith("--")]
if "--dist" in sys.argv:          # r06: with a one-rank RCCL process group alive in the process (what bench.py's N = 1 line has since r06)
    from tracklab_amd import dist as tdist
    d = tdist.init_single("nccl")
    t = torch.ones(1, device="cuda")
    d.all_reduce(t); d.barrier()
    print("one-rank nccl group initialised, all_reduce + barrier done", flush=True)
rng = np.random.default_rng(0)
T = 6
heads, frames = [], []
ratio = min(640 / 1080, 640 / 1920)
for fr in SyntheticStream(0, 100, T):
    heads.append(synth_yolox_head(rng, fr["dets"][:, :4], ratio=ratio))
    frames.append(render_frame(rng, fr["gt_boxes"]))
dh = torch.from_numpy(np.stack(heads)).cuda()
fr1 = torch.from_numpy(frames[0][None]).cuda()
for prio in (args or ["-1", "0"]):
    os.environ["TLK_DET_PRIO"] = prio if prio != "serial" else "0"
    pipe = gp.DetReidTrackPipeline("m", n_streams=1, frames_per_step=1, max_dets=104, dtype=torch.float16, overlap_stages=prio != "serial")
    for j in range(8):
        pipe.step(fr1, dh[j % T:j % T + 1], fetch=False)
    pipe.synchronize()
    t0 = time.perf_counter()
    for j in range(100):
        pipe.step(fr1, dh[j % T:j % T + 1], fetch=False)
    pipe.synchronize()
    print(f"detector-stage stream priority {prio}: {100 / (time.perf_counter() - t0):.1f} frames/s (overlap={pipe.overlap}; {pipe.overlap_note})", flush=True)
    pipe.close()
