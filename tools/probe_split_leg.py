"""GPU box: the split-precision leg of bench.py alone (config 3: YOLOX-m + ReID R50, 24 frames per step, 100 objects), for rocprofv3 passes.
usage: python tools/probe_split_leg.py [det_split 0|1] [steps]      env TLK_SPLIT_SCALES=0: unscaled planes (r05)"""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracklab_amd import gpu_pipeline as gp
from tracklab_amd.synth import SyntheticStream, render_frame, synth_yolox_head

det_split = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
F = 24
rng = np.random.default_rng(0)
ratio = min(640 / 1080, 640 / 1920)
stream = list(SyntheticStream(0, 100, F))
heads = torch.from_numpy(np.stack([synth_yolox_head(rng, fr["dets"][:, :4], ratio=ratio) for fr in stream])).cuda()
frames = torch.from_numpy(np.stack([render_frame(rng, fr["gt_boxes"]) for fr in stream])).cuda()
pipe = gp.DetReidTrackPipeline("m", n_streams=1, frames_per_step=F, max_dets=104, dtype=torch.float32, reid_split_precision=True, detector_split_precision=det_split)
for _ in range(2):
    pipe.step(frames, heads, fetch=False)
pipe.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    pipe.step(frames, heads, fetch=False)
pipe.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"split leg (detector split: {det_split}, plane scales: {os.environ.get('TLK_SPLIT_SCALES', '1')}): {dt * 1e3:.1f} ms / step = {F / dt:.1f} frames/s", flush=True)
pipe.close()
