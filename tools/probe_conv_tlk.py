"""Probe (not product): tlk_conv2d_nhwc_f32 per layer shape of the config-3 step and per tile configuration, TFLOP/s against the fp32 MFMA
peak (157.3), beside MIOpen's fp32 convolution of the same shape.  python tools/probe_conv_tlk.py [B_reid] [B_det] [cfgs e.g. -1,0,2]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tracklab_amd.backbones  # noqa: F401  (MIOpen env + benchmark mode)
from tracklab_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
BD = int(sys.argv[2]) if len(sys.argv) > 2 else 24
CFGS = [int(c) for c in sys.argv[3].split(",")] if len(sys.argv) > 3 else [-1, 0, 1, 2, 3, 5]
MIOPEN = os.environ.get("PROBE_MIOPEN", "1") == "1"


def timed(fn, n=5):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


# (name, batch, cin, cout, k, stride, H, W, count per forward, residual)
R50 = [("r50 conv1 7x7s2 (4ch)", B, 4, 64, 7, 2, 384, 128, 1, False),
       ("l1 1x1 64>64", B, 64, 64, 1, 1, 96, 32, 1, False), ("l1 1x1 256>64", B, 256, 64, 1, 1, 96, 32, 2, False),
       ("l1 3x3 64", B, 64, 64, 3, 1, 96, 32, 3, False), ("l1 1x1 64>256 +res", B, 64, 256, 1, 1, 96, 32, 4, True),
       ("l2 1x1 256>128", B, 256, 128, 1, 1, 96, 32, 1, False), ("l2 3x3 128 s2", B, 128, 128, 3, 2, 96, 32, 1, False),
       ("l2 down 256>512 s2", B, 256, 512, 1, 2, 96, 32, 1, False), ("l2 1x1 512>128", B, 512, 128, 1, 1, 48, 16, 3, False),
       ("l2 3x3 128", B, 128, 128, 3, 1, 48, 16, 3, False), ("l2 1x1 128>512 +res", B, 128, 512, 1, 1, 48, 16, 4, True),
       ("l3 1x1 512>256", B, 512, 256, 1, 1, 48, 16, 1, False), ("l3 3x3 256 s2", B, 256, 256, 3, 2, 48, 16, 1, False),
       ("l3 down 512>1024 s2", B, 512, 1024, 1, 2, 48, 16, 1, False), ("l3 1x1 1024>256", B, 1024, 256, 1, 1, 24, 8, 5, False),
       ("l3 3x3 256", B, 256, 256, 3, 1, 24, 8, 5, False), ("l3 1x1 256>1024 +res", B, 256, 1024, 1, 1, 24, 8, 6, True),
       ("l4 1x1 1024>512", B, 1024, 512, 1, 1, 24, 8, 1, False), ("l4 down 1024>2048", B, 1024, 2048, 1, 1, 24, 8, 1, False),
       ("l4 1x1 2048>512", B, 2048, 512, 1, 1, 24, 8, 2, False), ("l4 3x3 512", B, 512, 512, 3, 1, 24, 8, 3, False),
       ("l4 1x1 512>2048 +res", B, 512, 2048, 1, 1, 24, 8, 3, True), ("reduce 2048>256", B, 2048, 256, 1, 1, 24, 8, 1, False)]
YM = [("yolox-m stem 12>48", BD, 12, 48, 3, 1, 320, 320, 1, False), ("ym dark2 3x3s2 48>96", BD, 48, 96, 3, 2, 320, 320, 1, False),
      ("ym 1x1 96>48 @160", BD, 96, 48, 1, 1, 160, 160, 4, False), ("ym 3x3 48 @160", BD, 48, 48, 3, 1, 160, 160, 2, False),
      ("ym 3x3s2 96>192", BD, 96, 192, 3, 2, 160, 160, 1, False), ("ym 3x3 96 @80", BD, 96, 96, 3, 1, 80, 80, 8, False),
      ("ym 1x1 192>96 @80", BD, 192, 96, 1, 1, 80, 80, 6, False), ("ym 3x3s2 192>384", BD, 192, 384, 3, 2, 80, 80, 1, False),
      ("ym 3x3 192 @40", BD, 192, 192, 3, 1, 40, 40, 8, False), ("ym head 3x3 192 @80", BD, 192, 192, 3, 1, 80, 80, 4, False),
      ("ym 3x3 384 @20", BD, 384, 384, 3, 1, 20, 20, 4, False), ("ym 1x1 768>384 @20", BD, 768, 384, 1, 1, 20, 20, 3, False)]

for table, title in ((R50, f"ReID ResNet-50, {B} crops of 384x128"), (YM, f"YOLOX-m (main shapes), {BD} frames of 640x640")):
    print(f"== {title}: TFLOP/s (ms per call) per tile configuration; cfg -1 = heuristic ==")
    tot = {}
    for name, nb, cin, cout, k, s, H, W, cnt, res in table:
        Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
        flop = 2.0 * nb * Ho * Wo * cout * cin * k * k
        x = torch.randn(nb, cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
        b = torch.randn(cout, device="cuda")
        r = torch.randn(nb, cout, Ho, Wo, device="cuda").contiguous(memory_format=torch.channels_last) if res else None
        y = torch.empty(nb, cout, Ho, Wo, device="cuda").contiguous(memory_format=torch.channels_last)
        row = {}
        _lib.conv2d_nhwc_f32(x[:0], w, None)
        for cfg in CFGS:
            _lib.check(_lib.lib().tlk_conv2d_set_config(cfg))
            row[f"cfg{cfg}"] = timed(lambda: _lib.conv2d_nhwc_f32(x, w, b, "relu", r, stride=s, out=y))
        _lib.lib().tlk_conv2d_set_config(-1)
        if MIOPEN:
            def lib_route():
                t = F.conv2d(x, w, None, s, k // 2) + b.view(1, -1, 1, 1)
                if r is not None:
                    t = t + r
                return F.relu(t, inplace=True)
            row["miopen+epi"] = timed(lib_route, 3)
        for t, v in row.items():
            tot[t] = tot.get(t, 0.0) + v * cnt
        tot["flop"] = tot.get("flop", 0.0) + flop * cnt
        print(f"{name:24s} x{cnt} {flop / 1e9:8.1f} GF: " + "  ".join(f"{t} {flop / v / 1e12:6.1f} ({v * 1e3:7.3f})" for t, v in row.items()), flush=True)
        del x, w, y, r
    fl = tot.pop("flop")
    print("   sum over the forward (these shapes x count), ms:", {t: round(v * 1e3, 2) for t, v in tot.items()},
          " TFLOP/s:", {t: round(fl / v / 1e12, 1) for t, v in tot.items()}, f" ({fl / 1e12:.2f} TFLOP)")
