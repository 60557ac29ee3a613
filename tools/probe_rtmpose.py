"""GPU box: the RTMPose-m forward of config 4 alone (2400 crops of 256 x 192), eager, timed with events; run under rocprofv3 for the kernel mix.
usage: python tools/probe_rtmpose.py [crops] [dtype f16|f32] [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracklab_amd.backbones.rtmpose import rtmpose  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
dt = {"f16": torch.float16, "f32": torch.float32}[sys.argv[2] if len(sys.argv) > 2 else "f16"]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
net = rtmpose("m", "cuda", dt)
x = torch.randn(n, 3, 256, 192, device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(2):
        y = net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        y = net(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
print(f"rtmpose-m {n} crops {dt}: {ms:.2f} ms per forward ({n / ms * 1e3:.0f} crops/s)", flush=True)

# per-module breakdown (events around every leaf convolution / fused module; eager, so launch gaps are included in `total` only)
import collections
import torch.nn as nn
from tracklab_amd.backbones.common import ConvBiasAct
from tracklab_amd.backbones.rtmpose import ChannelAttention, GAU, DWConvBiasAct, CSPNeXtBlock, CSPLayer, SPPBottleneck

recs = []


def hook_pre(m, inp):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    m._e0 = e


def hook_post(m, inp, out):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    xs = inp[0].shape
    if isinstance(m, ConvBiasAct):
        c = m.conv
        key = f"ConvBiasAct k{c.kernel_size[0]} s{c.stride[0]} {c.in_channels}>{c.out_channels} @{xs[2]}x{xs[3]}"
    else:
        key = type(m).__name__ + f" {tuple(xs[1:])}"
    recs.append((key, m._e0, e))


for m in net.modules():
    if isinstance(m, (ConvBiasAct, DWConvBiasAct, CSPNeXtBlock, CSPLayer, SPPBottleneck, ChannelAttention, GAU)) or m is net.final_layer:
        m.register_forward_pre_hook(hook_pre)
        m.register_forward_hook(hook_post)
with torch.no_grad():
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); net(x); e1.record()
    torch.cuda.synchronize()
tot = e0.elapsed_time(e1)
agg = collections.OrderedDict()
for k, a, b in recs:
    c = agg.setdefault(k, [0, 0.0])
    c[0] += 1
    c[1] += a.elapsed_time(b)
s = 0.0
for k, (c, t) in agg.items():
    print(f"{k:50s} x{c:2d} {t:8.3f} ms")
    s += t
print(f"total {tot:.2f} ms (nested: DWConvBiasAct contains its pointwise ConvBiasAct, CSPNeXtBlock its two convolutions + the residual add, CSPLayer "
      f"everything of a stage but the strided convolution)")
