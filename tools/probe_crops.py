"""Probe (not product): time the two ReID crop kernels in isolation with HIP events on one bench launch's shapes
(24 frames x 104 slots, ~98 real crops per frame of the synthetic 100-object stream) and print us / algorithmic GB/s.
(r04: one kernel per output type; the TLK_CROP_KERNEL switch is gone.)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracklab_amd import _lib
from tracklab_amd.synth import SyntheticStream, render_frame

torch.cuda.set_device(0)
rng = np.random.default_rng(0)
B, MAXD = 24, 104
frames = np.stack([render_frame(rng, SyntheticStream(b, 100, 1).step()["gt_boxes"]) for b in range(B)])
f1080 = torch.from_numpy(frames).cuda()
boxes = np.zeros((B, MAXD, 4), dtype=np.float32)
counts = np.zeros(B, dtype=np.int32)
for b in range(B):
    d = SyntheticStream(b, 100, 1).step()["dets"]
    n = len(d)
    boxes[b, :n] = np.column_stack([d[:, 0], d[:, 1], d[:, 2] - d[:, 0], d[:, 3] - d[:, 1]])
    counts[b] = n
db, dc = torch.from_numpy(boxes).cuda(), torch.from_numpy(counts).cuda()
xyxy = np.zeros((B, MAXD, 7))
xyxy[..., 0], xyxy[..., 1] = boxes[..., 0], boxes[..., 1]
xyxy[..., 2], xyxy[..., 3] = boxes[..., 0] + boxes[..., 2], boxes[..., 1] + boxes[..., 3]
dx = torch.from_numpy(xyxy).cuda()
ncrops = int(counts.sum())
src = float(sum((np.rint(boxes[b, :counts[b], 2]) * np.rint(boxes[b, :counts[b], 3])).sum() for b in range(B)) * 3)
res = {"variant": os.environ.get("TLK_CROP_KERNEL", "2"), "crops": ncrops, "source_bytes": src}
for name, fn, hw, dt in (("crop 384x128 f16", lambda o: _lib.roi_crop_resize_norm(f1080, db, dc, 384, 128, "nhwc", torch.float16, out=o), (384, 128), torch.float16),
                         ("crop 384x128 f32", lambda o: _lib.roi_crop_resize_norm(f1080, db, dc, 384, 128, "nhwc", torch.float32, out=o), (384, 128), torch.float32),
                         ("pil  256x128 f16", lambda o: _lib.roi_crop_pil_resize_norm(f1080, dx, dc, 256, 128, "nhwc", torch.float16, out=o), (256, 128), torch.float16)):
    out = torch.zeros((B * MAXD, hw[0], hw[1], 3), dtype=dt, device="cuda")
    for _ in range(5):
        fn(out)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for e0, e1 in ev:
        e0.record(); fn(out); e1.record()
    torch.cuda.synchronize()
    ms = float(np.median([e0.elapsed_time(e1) for e0, e1 in ev]))
    alg = src + ncrops * 3 * hw[0] * hw[1] * out.element_size()
    res[name] = {"us": ms * 1e3, "alg_GBps": alg / (ms * 1e-3) / 1e9, "frac_of_8TBps": alg / (ms * 1e-3) / 8e12}
# write roof on this box: a plain fill of the crop tensor (write-only, perfectly coalesced) and a copy (read + write)
for name, shape, dt in (("fill 708MB f16", (B * MAXD, 384, 128, 3), torch.float16), ("fill 1.4GB f32", (B * MAXD, 384, 128, 3), torch.float32)):
    t = torch.empty(shape, dtype=dt, device="cuda"); u = torch.empty_like(t)
    for _ in range(3):
        t.fill_(1.0); u.copy_(t)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for e0, e1 in ev:
        e0.record(); t.fill_(1.0); e1.record()
    torch.cuda.synchronize()
    ms = float(np.median([e0.elapsed_time(e1) for e0, e1 in ev]))
    for e0, e1 in ev:
        e0.record(); u.copy_(t); e1.record()
    torch.cuda.synchronize()
    ms2 = float(np.median([e0.elapsed_time(e1) for e0, e1 in ev]))
    nbytes = t.numel() * t.element_size()
    res[name] = {"fill_us": ms * 1e3, "fill_GBps": nbytes / (ms * 1e-3) / 1e9, "copy_us": ms2 * 1e3, "copy_GBps_rw": 2 * nbytes / (ms2 * 1e-3) / 1e9}
    del t, u
print(json.dumps(res))
