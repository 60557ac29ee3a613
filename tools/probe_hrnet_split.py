"""Probe (not product): the part-based ReID forward on HRNet-W32 (bpbreid.yaml:53) at exact fp32 and in split-precision mode -- ms per forward,
and the per-kernel share of the split forward's joints (tlk_split_fuse_sum).    python tools/probe_hrnet_split.py [crops] [exact fp32 | split]      TLK_FUSE32=0: the exact route on torch's interpolate / add / relu / cat passes"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracklab_amd.backbones.reid import part_based_reid  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2211
dev = torch.device("cuda:0")
x = (torch.rand(B, 3, 384, 128, device=dev) * 255).contiguous(memory_format=torch.channels_last)
res = {}
with torch.no_grad():
    only = sys.argv[2] if len(sys.argv) > 2 else None
    for tag, split in (("exact fp32", False), ("split", True)):
        if only is not None and tag != only:
            continue
        net = part_based_reid(6, 256, device=dev, dtype=torch.float32, arch="hrnet32", split_precision=split)
        for _ in range(3):
            f = net.features(x)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            f = net.features(x)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        res[tag] = f
        print(f"HRNet-W32 x {B} crops, {tag:10s}: min {min(ts):8.3f} ms  median {sorted(ts)[2]:8.3f} ms", flush=True)
        del net
if len(res) == 2:
    a, b = res["exact fp32"], res["split"]
    print(f"max |exact - split| = {float((a - b).abs().max()):.3e}   (largest |feature| {float(a.abs().max()):.3e})")
