"""Probe (not product): ONE large layer (l4 3x3 512, 2400 crops) on the three convolution kernels, a few launches each -- the target of
`rocprofv3 --kernel-trace --pmc ...` passes (tools/pmc_conv.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracklab_amd import _lib

B, cin, cout, k, H, W = 2400, 512, 512, 3, 24, 8
x = torch.randn(B, cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
w = (torch.randn(cout, cin, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
b = torch.randn(cout, device="cuda")
xh, xl = _lib.split_planes(x); wh, wl = _lib.split_planes(w)
x16, w16 = x.half(), w.half()
for _ in range(4):
    _lib.conv2d_nhwc_16(xh, wh, b, "relu", x_lo=xl, weight_lo=wl)
    _lib.conv2d_nhwc_16(x16, w16, b, "relu")
    _lib.conv2d_nhwc_f32(x, w, b, "relu")
torch.cuda.synchronize()
