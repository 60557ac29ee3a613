"""Probe (not product): the two MFMA kernels at canonical sizes, for rocprofv3 timing / MFMA counters."""
import numpy as np
import torch
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracklab_amd import _lib
rng = np.random.default_rng(0)
T, G, N, D = 100, 100, 100, 512
gal = torch.from_numpy(rng.normal(0, 1, (T * G, D)).astype(np.float32)).cuda()
offs = torch.arange(0, T * G + 1, G, dtype=torch.int32).cuda()
dets = torch.from_numpy(rng.normal(0, 1, (N, D)).astype(np.float32)).cuda()
q = torch.from_numpy(rng.normal(0, 1, (110, 6, 256)).astype(np.float32)).cuda()
g = torch.from_numpy(rng.normal(0, 1, (100, 6, 256)).astype(np.float32)).cuda()
qv = torch.ones((110, 6), dtype=torch.uint8).cuda()
gv = torch.ones((100, 6), dtype=torch.uint8).cuda()
for _ in range(20):
    _lib.cosine_gallery_min(gal, offs, dets)
    _lib.partdist(q, qv, g, gv)
torch.cuda.synchronize()
print("flops cosine", 2 * T * G * N * D, "partdist", 2 * 110 * 100 * 6 * 256)
