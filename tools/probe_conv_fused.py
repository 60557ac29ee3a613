"""Probe (not product): 3x3 conv + bias + ReLU -- MIOpen conv + tlk_bias_act_nhwc (current) vs aten::miopen_convolution_relu."""
import os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tracklab_amd.backbones  # noqa: F401  (MIOpen env + benchmark mode)
from tracklab_amd import _lib

def bench(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        y = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, y

for (B, C, H, W, s) in [(2496, 64, 96, 32, 1), (2496, 128, 48, 16, 1), (2496, 256, 24, 8, 1), (2496, 512, 24, 8, 1), (24, 96, 160, 160, 1), (24, 192, 80, 80, 1)]:
    x = torch.randn(B, C, H, W, device="cuda", dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, C, 3, 3, device="cuda", dtype=torch.float16) * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(C, device="cuda", dtype=torch.float16)
    t_split, y0 = bench(lambda: _lib.bias_act_(F.conv2d(x, w, None, s, 1), b, "relu"))
    t_conv, _ = bench(lambda: F.conv2d(x, w, None, s, 1))
    try:
        t_fused, y1 = bench(lambda: torch.ops.aten.miopen_convolution_relu(x, w, b, [s, s], [1, 1], [1, 1], 1))
        err = (y0.float() - y1.float()).abs().max().item()
    except Exception as e:
        t_fused, err = float("nan"), str(e)[:80]
    print(f"B={B} C={C} {H}x{W}: conv alone {t_conv*1e3:.3f} ms, conv+epilogue {t_split*1e3:.3f} ms, miopen_convolution_relu {t_fused*1e3:.3f} ms, err {err}", flush=True)
