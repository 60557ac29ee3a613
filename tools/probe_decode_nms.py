"""Probe (not product): tlk_yolox_decode_nms on the bench's launch shape (24 frames x 100 objects + duplicates), HIP events, and the bytes it must read."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracklab_amd import _lib
from tracklab_amd.synth import SyntheticStream, synth_yolox_head

B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for nobj, dup in ((100, 3), (50, 3), (100, 8)):
    rng = np.random.default_rng(0)
    heads = np.stack([synth_yolox_head(rng, SyntheticStream(s, nobj, 1).step()["dets"][:, :4], dup=dup) for s in range(B)])
    d = torch.from_numpy(heads).cuda()
    out = _lib.yolox_decode_nms(d, 640, float(np.float32(640 / 1920)), 1920, 1080, 128)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        _lib.yolox_decode_nms(d, 640, float(np.float32(640 / 1920)), 1920, 1080, 128)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    cand = int(((heads[..., 4] * heads[..., 5]) > 0.7).sum() / B)
    print(f"decode+NMS {B} frames x {nobj} objects (dup {dup}, ~{cand} candidates/frame, kept {float(out['counts'].float().mean()):.0f}): {us:.1f} us per launch; "
          f"{heads.nbytes / 1e6:.1f} MB head -> {heads.nbytes / us / 1e3:.1f} GB/s")
