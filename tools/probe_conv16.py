"""Probe (not product): tlk_conv2d_nhwc_16 per ReID layer shape -- split mode (fp32-class, TFLOP/s counted on the ALGORITHMIC flops of the fp32
convolution it stands for) and f16 mode, beside the exact-fp32 kernel.  python tools/probe_conv16.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tracklab_amd.backbones  # noqa: F401
from tracklab_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2400


def timed(fn, n=5):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


R50 = [("conv1 7x7s2 (8ch)", 8, 64, 7, 2, 384, 128, 1, False), ("l1 1x1 64>64", 64, 64, 1, 1, 96, 32, 1, False), ("l1 1x1 256>64", 256, 64, 1, 1, 96, 32, 2, False),
       ("l1 3x3 64", 64, 64, 3, 1, 96, 32, 3, False), ("l1 1x1 64>256 +res", 64, 256, 1, 1, 96, 32, 4, True),
       ("l2 1x1 256>128", 256, 128, 1, 1, 96, 32, 1, False), ("l2 3x3 128 s2", 128, 128, 3, 2, 96, 32, 1, False),
       ("l2 down 256>512 s2", 256, 512, 1, 2, 96, 32, 1, False), ("l2 1x1 512>128", 512, 128, 1, 1, 48, 16, 3, False),
       ("l2 3x3 128", 128, 128, 3, 1, 48, 16, 3, False), ("l2 1x1 128>512 +res", 128, 512, 1, 1, 48, 16, 4, True),
       ("l3 1x1 512>256", 512, 256, 1, 1, 48, 16, 1, False), ("l3 3x3 256 s2", 256, 256, 3, 2, 48, 16, 1, False),
       ("l3 down 512>1024 s2", 512, 1024, 1, 2, 48, 16, 1, False), ("l3 1x1 1024>256", 1024, 256, 1, 1, 24, 8, 5, False),
       ("l3 3x3 256", 256, 256, 3, 1, 24, 8, 5, False), ("l3 1x1 256>1024 +res", 256, 1024, 1, 1, 24, 8, 6, True),
       ("l4 1x1 1024>512", 1024, 512, 1, 1, 24, 8, 1, False), ("l4 down 1024>2048", 1024, 2048, 1, 1, 24, 8, 1, False),
       ("l4 1x1 2048>512", 2048, 512, 1, 1, 24, 8, 2, False), ("l4 3x3 512", 512, 512, 3, 1, 24, 8, 3, False),
       ("l4 1x1 512>2048 +res", 512, 2048, 1, 1, 24, 8, 3, True), ("reduce 2048>256", 2048, 256, 1, 1, 24, 8, 1, False)]
tot = {"split": 0.0, "f16": 0.0, "f32": 0.0, "flop": 0.0}
print(f"ReID ResNet-50, {B} crops: TFLOP/s on the algorithmic flops (ms per call)")
for name, cin, cout, k, s, H, W, cnt, res in R50:
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    flop = 2.0 * B * Ho * Wo * cout * (3 if cin == 8 else cin) * k * k
    x = torch.randn(B, cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device="cuda")
    r = torch.randn(B, cout, Ho, Wo, device="cuda").contiguous(memory_format=torch.channels_last) if res else None
    xh, xl = _lib.split_planes(x); wh, wl = _lib.split_planes(w)
    rh, rl = _lib.split_planes(r) if res else (None, None)
    only32 = os.environ.get("PROBE_ONLY") == "f32"          # A/B runs of the exact kernel alone
    t_split = float("nan") if only32 else timed(lambda: _lib.conv2d_nhwc_16(xh, wh, b, "relu", rh, stride=s, x_lo=xl, weight_lo=wl, residual_lo=rl))
    x16, w16, r16 = x.half(), w.half(), (r.half() if res else None)
    t_f16 = float("nan") if only32 else timed(lambda: _lib.conv2d_nhwc_16(x16, w16, b, "relu", r16, stride=s))
    t_f32 = timed(lambda: _lib.conv2d_nhwc_f32(x, w, b, "relu", r, stride=s)) if cin % 4 == 0 else float("nan")
    for key, t in (("split", t_split), ("f16", t_f16), ("f32", t_f32)):
        tot[key] += t * cnt
    tot["flop"] += flop * cnt
    print(f"{name:22s} x{cnt} {flop / 1e9:8.1f} GF: split {flop / t_split / 1e12:6.1f} ({t_split * 1e3:7.3f})  f16 {flop / t_f16 / 1e12:6.1f} ({t_f16 * 1e3:7.3f})  "
          f"exact f32 {flop / t_f32 / 1e12:6.1f} ({t_f32 * 1e3:7.3f})", flush=True)
    del x, w, r, xh, xl, wh, wl, rh, rl, x16, w16, r16
fl = tot.pop("flop")
print("sum over the forward, ms:", {k_: round(v * 1e3, 2) for k_, v in tot.items()}, " TFLOP/s:", {k_: round(fl / v / 1e12, 1) for k_, v in tot.items()}, f"({fl / 1e12:.2f} TFLOP)")
