"""Probe (not product): is the split-precision route's distance from the exact-fp32 network arithmetic error or the network's own conditioning?
For ResNet-50 and HRNet-W32 (random init, as in the bench): features and part embeddings of (a) the split route and (b) the EXACT route on inputs
perturbed by a relative 2^-22 per pixel (the precision of one (hi, lo) pair), both against the exact route on the clean inputs.
    python tools/probe_split_conditioning.py [crops]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracklab_amd.backbones.reid import part_based_reid  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B, 3, 384, 128, device=dev, generator=g).contiguous(memory_format=torch.channels_last)        # normalised crops (mean 0, std 1)
xp = (x * (1 + 2.0 ** -22 * torch.randn(x.shape, device=dev, generator=g).contiguous(memory_format=torch.channels_last)))


def rel(a, b):
    return float((a - b).abs().max() / a.abs().max())


def cosd(a, b):
    c = torch.nn.functional.cosine_similarity(a.double().flatten(1), b.double().flatten(1), dim=1)
    return float((1 - c).max())


with torch.no_grad():
    for arch in ("resnet50", "hrnet32"):
        exact = part_based_reid(6, 256, device=dev, dtype=torch.float32, arch=arch)
        split = part_based_reid(6, 256, device=dev, dtype=torch.float32, arch=arch, split_precision=True)
        f0 = exact.features(x); e0, _ = exact.head(f0)
        f1 = split.features(x); e1, _ = split.head(f1)
        f2 = exact.features(xp); e2, _ = exact.head(f2)
        print(f"{arch:9s} x {B}: largest |feature| {float(f0.abs().max()):.3e}, largest |embedding| {float(e0.abs().max()):.3e}")
        print(f"   split route      vs exact: features {rel(f0, f1):.2e} of the largest, embeddings {rel(e0, e1):.2e} of the largest, 1 - cos {cosd(e0, e1):.1e}")
        print(f"   exact, inputs perturbed by 2^-22: features {rel(f0, f2):.2e}, embeddings {rel(e0, e2):.2e}, 1 - cos {cosd(e0, e2):.1e}", flush=True)
        del exact, split
