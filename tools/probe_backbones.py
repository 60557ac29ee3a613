"""Probe (not product): backbone forward times on the GPU box for the design notes."""
import sys, time, torch
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracklab_amd.backbones.yolox import yolox
from tracklab_amd.backbones.reid import part_based_reid

def timeit(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3

torch.backends.cudnn.benchmark = bool(int(sys.argv[1])) if len(sys.argv) > 1 else False
for size in ("s", "m"):
    for cl in (True, False):
        for dt in (torch.float16, torch.bfloat16):
            m = yolox(size, dtype=dt, channels_last=cl)
            for B in (1, 8, 16):
                x = torch.rand(B, 3, 640, 640, device="cuda", dtype=dt) * 255
                if cl: x = x.contiguous(memory_format=torch.channels_last)
                with torch.no_grad():
                    t0=time.perf_counter(); m(x); torch.cuda.synchronize(); first=(time.perf_counter()-t0)*1e3
                    ms = timeit(lambda: m(x))
                print(f"yolox-{size} cl={cl} {dt} B={B}: {ms:.2f} ms/batch {ms/B:.3f} ms/frame (first {first:.0f} ms)", flush=True)
for cl in (True, False):
    for dt in (torch.float16, torch.bfloat16):
        m = part_based_reid(dtype=dt, channels_last=cl)
        for B in (100, 400, 800):
            x = torch.randn(B, 3, 384, 128, device="cuda", dtype=dt)
            if cl: x = x.contiguous(memory_format=torch.channels_last)
            with torch.no_grad():
                t0=time.perf_counter(); m(x); torch.cuda.synchronize(); first=(time.perf_counter()-t0)*1e3
                ms = timeit(lambda: m(x), n=5, w=2)
            print(f"reid-r50 cl={cl} {dt} B={B}: {ms:.2f} ms/batch {ms/B*100:.3f} ms/100crops (first {first:.0f} ms)", flush=True)
