"""Probe (not product): does running the front of the fp32 ReID ResNet-50 (stem, pool, layer1, layer2 -- where the convolutions are HBM-bound on
7.5 GB activation tensors) in CHUNKS of crops keep the inter-layer activations in the 256 MB infinity cache?  Whole batch per layer vs a loop over
chunks, both replayed from a hipGraph.  python tools/probe_reid_chunks.py [crops]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracklab_amd.backbones.reid import part_based_reid

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
net = part_based_reid(dtype=torch.float32)
bb = net.backbone
x = torch.randn(B, 3, 384, 128, device="cuda").contiguous(memory_format=torch.channels_last)


def front(t, upto):
    t = bb.pool(bb.conv1(t))
    t = bb.layer1(t)
    if upto >= 2:
        t = bb.layer2(t)
    if upto >= 3:
        t = bb.layer3(t)
    return t


def graph_time(fn, n=3):
    with torch.no_grad():
        fn(); fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for upto in (1, 2, 3):
    with torch.no_grad():
        shp = front(x[:2], upto).shape
    out = torch.empty((B,) + tuple(shp[1:]), device="cuda").contiguous(memory_format=torch.channels_last)

    def whole():
        out.copy_(front(x, upto))
    base = graph_time(whole)
    line = f"stem + pool + layers 1..{upto}, {B} crops fp32: whole batch {base:.1f} ms;"
    for ch in (16, 32, 64, 128, 300):
        def chunked():
            for i in range(0, B, ch):
                out[i:i + ch].copy_(front(x[i:i + ch], upto))
        line += f"  chunks of {ch}: {graph_time(chunked):.1f} ms;"
    print(line, flush=True)
