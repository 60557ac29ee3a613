"""Letterbox kernel timing: 32 frames (H x W, default 1080p) -> 640 focus_nhwc f16 (= one config2 bench launch), HIP events around 200 launches.
TLK_LETTERBOX_WAVE=0 selects letterbox_lds_kernel (r01/r02), default letterbox_wave_kernel (r03).  python tools/probe_letterbox.py [H W [B]]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tracklab_amd import _lib, roofline as rl                             # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
rng = np.random.default_rng(0)
frames = torch.from_numpy(rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)).cuda()
out = None
for _ in range(5):
    out, ratio = _lib.letterbox(frames, 640, "focus_nhwc", torch.float16)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for rep in range(5):
    e0.record()
    for _ in range(40):
        _lib.letterbox(frames, 640, "focus_nhwc", torch.float16)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 40 * 1e3)
rh, rw = int(H * ratio), int(W * ratio)
alg = rl.letterbox_bytes(H, W, 640, rh, rw, elem_bytes=2) * B
us = float(np.median(ts))
print(f"letterbox {B}x{H}x{W} wave={os.environ.get('TLK_LETTERBOX_WAVE', '1')}: {us:.1f} us/launch (runs {[round(t, 1) for t in ts]}), "
      f"{alg / 1e6:.1f} MB algorithmic -> {alg / us / 1e6:.2f} TB/s = {alg / us / 1e6 / 8:.3f} of 8 TB/s")
