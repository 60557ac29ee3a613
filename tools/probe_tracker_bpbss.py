"""Probe (not product): BPBReID-StrongSORT association time per frame + per-phase breakdown (TLK_BPBSS_PROF=1) at the bench's shape
(100 objects, K = 6 parts, D = 256) through the batched device entry point, one stream."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

PROF = os.environ.get("TLK_BPBSS_PROF")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracklab_amd import _lib
from tracklab_amd.synth import SyntheticStream, ltrb_to_ltwh_rows

nobj = int(sys.argv[1]) if len(sys.argv) > 1 else 100
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
K, S, F, MAXD = 6, 1, 200, 128
cfg = dict(ema_alpha=0.9, mc_lambda=0.995, max_dist=0.5, motion_criterium="iou", max_iou_distance=0.8, max_oks_distance=0.7, max_age=300, n_init=0,
           nn_budget=100, min_bbox_confidence=0.0, only_position_for_kf_gating=False, max_kalman_prediction_without_update=7,
           matching_strategy="strong_sort_matching", gating_thres_factor=1, w_kfgd=1, w_reid=1, w_st=1)
bank = _lib.BpbssBank(K, D, **cfg, wrapper_mode=True, n_streams=S, max_dets=MAXD, max_tracks=512)
ids = np.zeros((S, F, MAXD), np.int64); ltwh = np.zeros((S, F, MAXD, 4)); emb = np.zeros((S, F, MAXD, K, D), np.float32)
vis = np.zeros((S, F, MAXD, K), np.uint8); conf = np.ones((S, F, MAXD)); cnt = np.zeros((S, F), np.int32)
for f, fr in enumerate(SyntheticStream(0, nobj, F, parts=K, dim=D, with_embeddings=True)):
    d = fr["dets"]; n = len(d)
    ids[0, f, :n] = d[:, 6]; ltwh[0, f, :n] = ltrb_to_ltwh_rows(d[:, :4]); emb[0, f, :n] = fr["embeddings"]; vis[0, f, :n] = fr["visibility"]; cnt[0, f] = n
t = [torch.from_numpy(a).cuda() for a in (ids, ltwh, emb, vis, conf, cnt)]
rows = torch.zeros((S, F, MAXD, _lib.BPBSS_ROW.itemsize), dtype=torch.uint8, device="cuda"); oc = torch.zeros((S, F), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
bank.update_dev(*(x.data_ptr() for x in t), F, rows.data_ptr(), MAXD, oc.data_ptr())
torch.cuda.synchronize()
dt = time.perf_counter() - t0
out = {"nobj": nobj, "D": D, "us_per_frame_all_three_kernels": dt / F * 1e6, "rows_per_frame": oc.float().mean().item(), "profiled": bool(PROF)}
if PROF:
    L = _lib.lib()
    L.tlk_bpbss_get_profile.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]
    buf = (C.c_longlong * 16)()
    _lib.check(L.tlk_bpbss_get_profile(bank._h, 0, buf))
    names = ["filter+predict", "det+gate prep", "appearance cost fill", "LSA A + lists", "set order + motion fill", "LSA B + lists", "KF update matched",
             "embedding EMA", "misses + births", "deaths + rows"]
    out["phases_us_per_frame"] = {n: buf[i] * 10 / F / 1e3 for i, n in enumerate(names)}
    out["shader_clock_MHz"] = buf[14] / max(buf[15], 1) * 100.0
print(json.dumps(out))
