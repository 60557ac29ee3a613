"""Probe (not product): Deep-OC-SORT kernel time per frame + per-phase breakdown (TLK_DEEPOCSORT_PROF=1)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

os.environ.setdefault("TLK_DEEPOCSORT_PROF", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracklab_amd import _lib
from tracklab_amd.synth import SyntheticStream

nobj = int(sys.argv[1]) if len(sys.argv) > 1 else 100
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
S, F, MAXD = 1, 200, 128
hyper = dict(det_thresh=0, max_age=50, min_hits=1, iou_threshold=0.22136877277096445, delta_t=1, asso_func="giou", inertia=0.3941737016672115,
             w_association_emb=0.75, alpha_fixed_emb=0.95, aw_param=0.5, cmc_off=True)
bank = _lib.DeepOCSortBank(D, **hyper, min_confidence=0.4, wrapper_mode=True, n_streams=S, max_dets=MAXD)
dets = np.zeros((S, F, MAXD, 7)); embs = np.zeros((S, F, MAXD, D), np.float32); counts = np.zeros((S, F), dtype=np.int32)
for s in range(S):
    for f, fr in enumerate(SyntheticStream(s, nobj, F, parts=1, dim=D, with_embeddings=True)):
        d = fr["dets"]
        dets[s, f, :len(d)] = d; embs[s, f, :len(d)] = fr["embeddings"][:, 0, :]; counts[s, f] = len(d)
d_dets, d_embs, d_counts = torch.from_numpy(dets).cuda(), torch.from_numpy(embs).cuda(), torch.from_numpy(counts).cuda()
d_out = torch.zeros((S, F, 256, 8), dtype=torch.float64, device="cuda")
d_oc = torch.zeros((S, F), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
bank.update_dev(d_dets.data_ptr(), d_embs.data_ptr(), d_counts.data_ptr(), F, d_out.data_ptr(), 256, d_oc.data_ptr(), None)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"nobj={nobj} D={D}: {dt / F * 1e6:.1f} us/frame (whole launch {dt * 1e3:.2f} ms), rows/frame {d_oc.float().mean().item():.1f}")
L = _lib.lib()
L.tlk_deepocsort_get_profile.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]
buf = (C.c_longlong * 16)()
_lib.check(L.tlk_deepocsort_get_profile(bank._h, 0, buf))
names = ["split", "predict", "nan-drop", "vel/kobs", "iou+1to1 test", "emb dots", "aw+cost fill", "lsa", "um lists", "upd matched+emb", "2nd round",
         "upd none+birth", "emit+death"]
tot = sum(buf[:13])
for i, n in enumerate(names):
    print(f"  {n:16s} {buf[i] * 10 / F / 1e3:8.2f} us/frame  {100.0 * buf[i] / max(tot, 1):5.1f}%")
