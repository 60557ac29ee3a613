"""Launch ONE byte-moving libtlk kernel on one bench launch's shape, a few times, for bench.py's live rocprofv3 --pmc passes
(roofline.traffic).  python tools/probe_traffic.py crop|pil|letterbox"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracklab_amd import _lib
from tracklab_amd.synth import SyntheticStream, render_frame

which = sys.argv[1] if len(sys.argv) > 1 else "crop"
torch.cuda.set_device(0)
rng = np.random.default_rng(0)
B, MAXD = (32, 0) if which == "letterbox" else (24, 104)
frames = torch.from_numpy(np.stack([render_frame(rng, SyntheticStream(b, 100, 1).step()["gt_boxes"]) for b in range(B)])).cuda()
if which == "letterbox":
    out = torch.empty((B, 320, 320, 12), dtype=torch.float16, device="cuda")
    fn = lambda: _lib.letterbox(frames, 640, "focus_nhwc", torch.float16, out=out, swap_rb=True)          # noqa: E731
else:
    boxes = np.zeros((B, MAXD, 4), dtype=np.float32)
    counts = np.zeros(B, dtype=np.int32)
    for b in range(B):
        d = SyntheticStream(b, 100, 1).step()["dets"]
        n = len(d)
        boxes[b, :n] = np.column_stack([d[:, 0], d[:, 1], d[:, 2] - d[:, 0], d[:, 3] - d[:, 1]])
        counts[b] = n
    dc = torch.from_numpy(counts).cuda()
    if which == "crop":
        db = torch.from_numpy(boxes).cuda()
        out = torch.empty((B * MAXD, 384, 128, 3), dtype=torch.float16, device="cuda")
        fn = lambda: _lib.roi_crop_resize_norm(frames, db, dc, 384, 128, "nhwc", torch.float16, out=out)       # noqa: E731
    else:
        xyxy = np.zeros((B, MAXD, 7))
        xyxy[..., 0], xyxy[..., 1] = boxes[..., 0], boxes[..., 1]
        xyxy[..., 2], xyxy[..., 3] = boxes[..., 0] + boxes[..., 2], boxes[..., 1] + boxes[..., 3]
        dx = torch.from_numpy(xyxy).cuda()
        out = torch.empty((B * MAXD, 256, 128, 3), dtype=torch.float16, device="cuda")
        fn = lambda: _lib.roi_crop_pil_resize_norm(frames, dx, dc, 256, 128, "nhwc", torch.float16, out=out)   # noqa: E731
for _ in range(4):
    fn()
torch.cuda.synchronize()
