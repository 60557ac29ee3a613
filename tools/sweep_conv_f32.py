"""Probe (not product): every distinct fp32 convolution of a network, timed under the tile configuration the heuristic picks and under every
configuration tlk_conv2d_set_config can force (0-9: conv_f32_mfma_kernel tiles, 21-39: the direct-to-LDS kernels on fp32 tensors) -- where a
forced configuration beats the heuristic, the heuristic is wrong for that shape.

    python tools/sweep_conv_f32.py yolox-m 24          # detector, 24 frames of 640 x 640
    python tools/sweep_conv_f32.py reid 2211           # part-based ReID ResNet-50, 2211 crops of 384 x 128 (reid-hrnet32: the HRNet-W32 backbone)
    python tools/sweep_conv_f32.py rtmpose-m 2211      # RTMPose-m, 2211 crops of 256 x 192
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracklab_amd import _lib  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "yolox-m"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 24
MIN_GAIN = float(os.environ.get("SWEEP_MIN_GAIN", "0.03"))
dev = torch.device("cuda:0")

# 1. record the convolutions of one forward pass
calls = []
real = _lib.conv2d_nhwc_f32


def recorder(x, weight, bias=None, act=None, residual=None, stride=1, pad=None, out=None, residual_after_act=False):
    y = real(x, weight, bias, act, residual, stride, pad, out=out, residual_after_act=residual_after_act)
    calls.append((tuple(x.shape), tuple(weight.shape), act, residual is not None, stride, pad, bool(residual_after_act), _lib.lib().tlk_conv2d_last_config()))
    return y


_lib.conv2d_nhwc_f32 = recorder
with torch.no_grad():
    if what.startswith("yolox"):
        from tracklab_amd.backbones.yolox import yolox
        net = yolox(what.split("-")[1], device=dev, dtype=torch.float32)
        net(torch.rand(batch, 3, 640, 640, device=dev).contiguous(memory_format=torch.channels_last))
    elif what.startswith("rtmpose"):
        from tracklab_amd.backbones.rtmpose import rtmpose
        net = rtmpose(what.split("-")[1], device=dev, dtype=torch.float32)
        net(torch.rand(batch, 3, 256, 192, device=dev).contiguous(memory_format=torch.channels_last))
    else:
        from tracklab_amd.backbones.reid import part_based_reid
        net = part_based_reid(6, 512, device=dev, dtype=torch.float32, arch="hrnet32" if "hrnet" in what else "resnet50")
        net(torch.rand(batch, 3, 384, 128, device=dev).contiguous(memory_format=torch.channels_last))
torch.cuda.synchronize()
_lib.conv2d_nhwc_f32 = real
shapes = {}
for c in calls:
    shapes.setdefault(c[:7], [0, c[7]])[0] += 1
print(f"{what} x {batch}: {len(calls)} convolutions, {len(shapes)} distinct shapes")


def timed(fn, n):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.02:              # warm clocks: GPU time, not host time -- synchronise inside the loop
        fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


L = _lib.lib()
CFGS = list(range(0, 10)) + list(range(21, 40))
tot_default = tot_best = 0.0
print(f"{'n':>3s} {'x (N,C,H,W)':>22s} {'w':>16s} s act res | {'heuristic':>9s} {'ms':>8s} {'TF/s':>6s} | best forced")
for key, (count, cfg_default) in sorted(shapes.items(), key=lambda kv: -kv[1][0]):
    xs, ws, act, res, stride, pad, raa = key
    x = torch.randn(xs, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(ws, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(ws[0], device=dev)
    y = real(x, w, b, act, None, stride, pad)
    r = torch.randn_like(y) if res else None
    fn = lambda: real(x, w, b, act, r, stride, pad, out=y, residual_after_act=raa)          # noqa: E731
    flop = 2.0 * y.shape[0] * y.shape[2] * y.shape[3] * ws[0] * ws[1] * ws[2] * ws[3]
    n = max(3, min(50, int(20.0 / max(flop / 100e12 * 1e3, 0.02))))
    L.tlk_conv2d_set_config(-1)
    t0 = timed(fn, n)
    ran = L.tlk_conv2d_last_config()
    best = (t0, "heuristic")
    row = []
    for cfg in CFGS:
        if L.tlk_conv2d_set_config(cfg) != 0:
            continue
        try:
            fn()
        except _lib.TlkError:
            continue
        if L.tlk_conv2d_last_config() != cfg:
            continue
        t = timed(fn, n)
        row.append((cfg, t))
        if t < best[0]:
            best = (t, cfg)
    L.tlk_conv2d_set_config(-1)
    tot_default += count * t0
    tot_best += count * best[0]
    flag = "" if best[1] == "heuristic" or best[0] > t0 * (1 - MIN_GAIN) else f"  <-- cfg {best[1]}: {best[0]:.4f} ms ({(t0 / best[0] - 1) * 100:.0f} % faster)"
    print(f"{count:3d} {str(xs):>22s} {str(ws):>16s} {stride} {str(act):>4s} {int(res)} | {ran:9d} {t0:8.4f} {flop / t0 / 1e9:6.1f} | "
          + " ".join(f"{c}:{t:.3f}" for c, t in sorted(row, key=lambda ct: ct[1])[:4]) + flag, flush=True)
print(f"sum over the network: heuristic {tot_default:.2f} ms, best forced per shape {tot_best:.2f} ms")
