"""Probe (not product): tlk_dwconv2d_nhwc and tlk_spp_maxpool_nhwc alone on the config-4 / config-3 shapes, HIP events, achieved GB/s on the
algorithmic bytes (every input element once + every output element once).  python tools/probe_dwconv.py [crops] [frames]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracklab_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
FR = int(sys.argv[2]) if len(sys.argv) > 2 else 24


def timed(fn, n=20):
    # warm: the first milliseconds of a process run at idle clocks (r04's table was taken with two warm-up calls only and reads ~25 % slow on its
    # first rows for that reason; its "variants" rows further down the same file were warm)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:              # (GPU time, not host time: synchronise inside the loop)
        fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


print(f"depthwise 5 x 5 + bias + SiLU (RTMPose-m CSPNeXt blocks, {B} crops of 256 x 192): algorithmic bytes = input + output once")
for dt, es in ((torch.float16, 2), (torch.float32, 4)):
    for c, h, w, cnt in ((48, 64, 48, 2), (96, 32, 24, 4), (192, 16, 12, 4), (384, 8, 6, 2)):
        x = torch.randn(B, c, h, w, device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last)
        wk = torch.randn(5, 5, c, device="cuda", dtype=dt) * 0.2
        b = torch.randn(c, device="cuda")
        out = torch.empty_like(x)
        t = timed(lambda: _lib.dwconv2d_nhwc(x, wk, b, "silu", out=out))
        by = 2.0 * x.numel() * es
        print(f"  {str(dt)[6:]:8s} {c:4d} ch @ {h:2d} x {w:2d}  x{cnt}: {t * 1e6:8.1f} us  {by / 1e6:8.1f} MB -> {by / t / 1e9:7.1f} GB/s = {by / t / 8e12:.2f} of 8 TB/s")
if os.environ.get("PROBE_DW_SHAPES"):
    print("lane shapes (tlk_dwconv_set_config: 1..4 = (columns per lane, rows ahead) = (1, 1), (2, 1), (1, 2), (2, 2)), 5 x 5 + SiLU / ReLU, us")
    for dt in (torch.float16, torch.float32):
        for c, h, w in ((48, 64, 48), (192, 16, 12), (384, 8, 6)):
            x = torch.randn(B, c, h, w, device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last)
            wk = torch.randn(5, 5, c, device="cuda", dtype=dt) * 0.2
            b = torch.randn(c, device="cuda")
            out = torch.empty_like(x)
            row = []
            for cfg in (1, 2, 3, 4):
                _lib.dwconv_set_config(cfg)
                row.append("%7.1f / %7.1f" % (timed(lambda: _lib.dwconv2d_nhwc(x, wk, b, "silu", out=out)) * 1e6, timed(lambda: _lib.dwconv2d_nhwc(x, wk, b, "relu", out=out)) * 1e6))
            _lib.dwconv_set_config(0)
            print(f"  {str(dt)[6:]:8s} {c:4d} ch @ {h:2d} x {w:2d}: " + "   ".join(row))
if os.environ.get("PROBE_DW_VARIANTS"):
    x = torch.randn(B, 48, 64, 48, device="cuda", dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    out = torch.empty_like(x)
    b = torch.randn(48, device="cuda")
    for k in (5, 3):
        wk = torch.randn(k, k, 48, device="cuda", dtype=torch.float16) * 0.2
        for act in ("silu", "relu", None):
            t = timed(lambda: _lib.dwconv2d_nhwc(x, wk, b, act, out=out))
            print(f"  variant f16 48 ch @ 64 x 48 k{k} act {act}: {t * 1e6:8.1f} us")
print("SPP pooling + concatenation (5 / 9 / 13): algorithmic bytes = input once + 4 x output")
for name, n, c, h, w in ((f"RTMPose-m {B} crops", B, 384, 8, 6), (f"YOLOX-m {FR} frames 640", FR, 384, 20, 20), (f"YOLOX-l {FR} frames 640", FR, 512, 20, 20)):
    for dt, es in ((torch.float16, 2), (torch.float32, 4)):
        x = torch.randn(n, c, h, w, device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last)
        out = torch.empty(n, 4 * c, h, w, device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last)
        t = timed(lambda: _lib.spp_maxpool_nhwc(x, out=out))
        by = 5.0 * x.numel() * es
        t_lib = timed(lambda: torch.cat([x] + [torch.nn.functional.max_pool2d(x, k, 1, k // 2) for k in (5, 9, 13)], 1))
        print(f"  {name:28s} {str(dt)[6:]:8s}: {t * 1e6:8.1f} us  {by / 1e6:7.1f} MB -> {by / t / 1e9:7.1f} GB/s   (three max_pool2d + cat: {t_lib * 1e6:8.1f} us)")

print("bias + activation (+ residual) epilogue pass of the f16 library route (tlk_bias_act_nhwc, in place): algorithmic bytes = read + write (+ residual read)")
for name, shape in (("ReID layer1 3x3 out", (B, 64, 96, 32)), ("ReID layer1 expansion out + residual", (B, 256, 96, 32)), ("ReID layer4 3x3 out", (B, 512, 24, 8)),
                    ("YOLOX-m stem out", (FR, 48, 320, 320))):
    x = torch.randn(*shape, device="cuda", dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    b = torch.randn(shape[1], device="cuda", dtype=torch.float16)
    r = torch.randn_like(x) if "residual" in name else None
    t = timed(lambda: _lib.bias_act_(x, b, "relu", r))
    by = (3.0 if r is not None else 2.0) * x.numel() * 2
    print(f"  {name:38s}: {t * 1e6:8.1f} us  {by / 1e6:8.1f} MB -> {by / t / 1e9:7.1f} GB/s = {by / t / 8e12:.2f} of 8 TB/s")
