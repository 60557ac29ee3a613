"""Probe (not product): tlk_split_fuse_sum on the shapes of HRNet-W32's joints at 2211 crops -- time per launch, algorithmic bytes, TB/s.
The PMC passes of tools/pmc_split_fuse.sh run over this script (each shape is launched 8 times in a row).    python tools/probe_split_fuse.py [crops]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracklab_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2211
dev = torch.device("cuda:0")
cl = lambda t: t.contiguous(memory_format=torch.channels_last)          # noqa: E731
H, W = 96, 32
# (name, channels, shift of the output grid, own planes?, shifts of the fp32 terms relative to the output)
CASES = [("stage 4, branch 0: 32 ch @ 96 x 32 = own planes + 3 up-sampled fp32 terms", 32, 0, True, (1, 2, 3)),
         ("stage 4, branch 1: 64 ch @ 48 x 16 = fp32 (stride-2 path) + own planes + 2 up-sampled", 64, 1, True, (0, 1, 2)),
         ("stage 4, branch 3: 256 ch @ 12 x 4 = 3 fp32 terms (stride-2 chains) + own planes", 256, 3, True, (0, 0, 0)),
         ("reduce: 256 ch @ 96 x 32 = 3 up-sampled fp32 terms -> the residual planes of the branch-0 convolution", 256, 0, False, (1, 2, 3))]
for name, c, s0, own, shifts in CASES:
    h, w = H >> s0, W >> s0
    terms, nbytes = [], 4.0 * B * h * w * c
    if own:
        hi, lo = _lib.split_planes(cl(torch.randn(B, c, h, w, device=dev)))
        terms.append((hi, lo, None))
        nbytes += 4.0 * B * h * w * c
    for s in shifts:
        terms.append(cl(torch.randn(B, c, h >> s, w >> s, device=dev)))
        nbytes += 4.0 * B * (h >> s) * (w >> s) * c
    if own and shifts[0] == 0:
        terms = terms[1:2] + terms[0:1] + terms[2:]        # the fp32 route's term order: lower-index branches first
    st = torch.tensor([1.0, 0.0], device=dev)
    out = tuple(torch.empty((B, c, h, w), dtype=torch.float16, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(2))
    _lib.split_fuse_sum(terms, relu=own, out=out, out_state=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        _lib.split_fuse_sum(terms, relu=own, out=out, out_state=st)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 8
    print(f"{name}\n    {ms * 1e3:8.1f} us per launch, {nbytes / 1e9:6.3f} GB algorithmic, {nbytes / ms / 1e9:6.2f} TB/s = {nbytes / ms / 1e9 / 8:.2f} of 8 TB/s", flush=True)
    del terms, out
