import sys, numpy as np
sys.path.insert(0, 'tests')
mode = sys.argv[1]
from tracklab_amd import _lib
from tracklab_amd.synth import SyntheticStream, ltrb_to_ltwh_rows
YAML = dict(ema_alpha=0.9, mc_lambda=0.995, max_dist=0.5, max_iou_distance=0.8, max_age=300, n_init=0, min_bbox_confidence=0.0,
            max_kalman_prediction_without_update=7, gating_thres_factor=1)
K, D = 6, 64
if mode in ("create", "update"):
    bank = _lib.BpbssBank(K, D, **YAML)
    if mode == "update":
        for fr in SyntheticStream(1, 20, 3, parts=K, dim=D, with_embeddings=True):
            d = fr["dets"]
            bank.update(d[:, 6].astype(np.int64), ltrb_to_ltwh_rows(d[:, :4]), fr["embeddings"], fr["visibility"], d[:, 4])
    bank.close()
if mode == "ocsort":
    b = _lib.OCSortBank(0.0); b.update(np.zeros((0, 7))); b.close()
import ctypes
L = _lib.lib()
import torch
try:
    torch.zeros(1).cuda(); print(mode, "torch ok")
except Exception as e:
    print(mode, "torch FAIL", e)
