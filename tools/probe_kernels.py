"""Probe (not product): launch the byte-moving libtlk kernels in isolation for rocprofv3 --pmc passes.
Shapes = one bench launch: letterbox of 32 (config2) frames, crop of 24 frames x 104 slots (config3; Pillow-semantics crop: config3s/3b/3d).
Calibration launches with a known byte count in the SAME access pattern: letterbox 640x640 -> 640 (every source byte
read exactly once by the same byte loads) and a 1 GiB torch copy (wide coalesced)."""
import numpy as np
import torch

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracklab_amd import _lib
from tracklab_amd.synth import SyntheticStream

torch.cuda.set_device(0)
rng = np.random.default_rng(0)
f1080 = torch.from_numpy(rng.integers(0, 255, (32, 1080, 1920, 3), dtype=np.uint8)).cuda()
f640 = torch.from_numpy(rng.integers(0, 255, (32, 640, 640, 3), dtype=np.uint8)).cuda()
boxes = np.zeros((24, 104, 4), dtype=np.float32)
counts = np.zeros(24, dtype=np.int32)
for b in range(24):
    d = SyntheticStream(b, 100, 1).step()["dets"]
    n = len(d)
    boxes[b, :n] = np.column_stack([d[:, 0], d[:, 1], d[:, 2] - d[:, 0], d[:, 3] - d[:, 1]])
    counts[b] = n
db, dc = torch.from_numpy(boxes).cuda(), torch.from_numpy(counts).cuda()
big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")     # 1 GiB
big2 = torch.empty_like(big)
out_lb = torch.empty((32, 320, 320, 12), dtype=torch.float16, device="cuda")
out_cr = torch.empty((24 * 104, 384, 128, 3), dtype=torch.float16, device="cuda")
out_pil = torch.empty((24 * 104, 256, 128, 3), dtype=torch.float16, device="cuda")
xyxy = np.zeros((24, 104, 7))                                      # tracker input rows [x1, y1, x2, y2, conf, cls, id] (config3s/3b/3d)
xyxy[..., 0], xyxy[..., 1] = boxes[..., 0], boxes[..., 1]
xyxy[..., 2], xyxy[..., 3] = boxes[..., 0] + boxes[..., 2], boxes[..., 1] + boxes[..., 3]
dx = torch.from_numpy(xyxy).cuda()
src_bytes = float(sum((np.clip(np.rint(boxes[b, :counts[b], 2]), 1, None) * np.clip(np.rint(boxes[b, :counts[b], 3]), 1, None)).sum() for b in range(24)) * 3)
print("crop source bytes (approx)", src_bytes, "crop out bytes", out_cr.numel() * 2)
for it in range(5):
    big2.copy_(big)
    _lib.letterbox(f640, 640, "focus_nhwc", torch.float16, out=out_lb)
    _lib.letterbox(f1080, 640, "focus_nhwc", torch.float16, out=out_lb)
    _lib.roi_crop_resize_norm(f1080[:24], db, dc, 384, 128, "nhwc", torch.float16, out=out_cr)
    _lib.roi_crop_pil_resize_norm(f1080[:24], dx, dc, 256, 128, "nhwc", torch.float16, out=out_pil)
torch.cuda.synchronize()
print("done")
