"""Measurement script kept under tests/ because it times the C oracle next to the GPU banks (only tests/, smoke() and bench.py's
cpu_baseline leg may touch oracle/): association-only throughput of every tracker bank, device entry points (`*_update_dev`), synthetic
1080p streams with 100 objects, next to the C oracle on one host core. Writes gpurun_out/trackers.json; tools/collect_profiles.py
turns it into profiles/<tag>_trackers.md.  usage: python tests/perf/bench_trackers.py [frames] [streams]"""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import oracle  # noqa: E402  (the oracle is the timed CPU port here)
from tracklab_amd import _lib  # noqa: E402
from tracklab_amd.synth import SyntheticStream, ltrb_to_ltwh_rows  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 120
S_MANY = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ONLY = set(sys.argv[3:])                                    # optional tracker names: run only these
NOBJ, MAXD = 100, 128
oracle.build()


def streams(n, **kw):
    out = []
    for s in range(n):
        out.append(list(SyntheticStream(100 + s, NOBJ, F, **kw)))
    return out


def timed(fn, sync=True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    if sync:
        torch.cuda.synchronize()
    return time.perf_counter() - t0


results = []


def run(name, make_bank, make_oracle, pack, step_dev, step_cpu, row_dtype, **skw):
    if ONLY and name not in ONLY:
        return
    for S in (1, S_MANY):
        data = streams(S, **skw)
        bank = make_bank(S)
        bufs = pack(data, S)
        rows = torch.zeros((S, F, MAXD * 2, row_dtype.itemsize), dtype=torch.uint8, device="cuda")
        ocnt = torch.zeros((S, F), dtype=torch.int32, device="cuda")
        step_dev(bank, bufs, rows, ocnt)                 # warm-up (also the correctness run)
        torch.cuda.synchronize()
        got = rows.cpu().numpy().view(row_dtype).reshape(S, F, -1)
        cnt = ocnt.cpu().numpy()
        assert (cnt >= 0).all(), name
        bank.reset(-1)
        dt = timed(lambda: step_dev(bank, bufs, rows, ocnt))
        entry = {"tracker": name, "streams": S, "frames_per_stream": F, "objects": NOBJ, "gpu_frames_per_s": S * F / dt,
                 "gpu_us_per_frame_per_launch": dt / F * 1e6}
        if S == 1:
            ref = make_oracle()
            t0 = time.perf_counter()
            ids_ok = True
            for f, fr in enumerate(data[0]):
                exp = step_cpu(ref, fr)
                ids_ok &= len(exp) == cnt[0, f]
            entry["cpu_oracle_frames_per_s"] = F / (time.perf_counter() - t0)
            entry["row_counts_equal_oracle"] = bool(ids_ok)
        results.append(entry)
        print(entry, flush=True)
        bank.close()


# ---- OC-SORT ----
HY = dict(asso_func="giou", delta_t=1, det_thresh=0, inertia=0.3941737016672115, iou_threshold=0.22136877277096445, max_age=50, min_hits=1, use_byte=False)


def pack_dets(data, S):
    d = np.zeros((S, F, MAXD, 7)); c = np.zeros((S, F), np.int32)
    for s in range(S):
        for f, fr in enumerate(data[s]):
            n = len(fr["dets"]); d[s, f, :n] = fr["dets"]; c[s, f] = n
    return {"dets": torch.from_numpy(d).cuda(), "counts": torch.from_numpy(c).cuda()}


OC_ROW = np.dtype([("r", "<f8", (8,))])
run("oc_sort", lambda S: _lib.OCSortBank(**HY, min_confidence=0.4, wrapper_mode=True, n_streams=S, max_dets=MAXD, max_tracks=256),
    lambda: oracle.OCSort(**HY), pack_dets,
    lambda b, B, rows, oc: b.update_dev(B["dets"].data_ptr(), B["counts"].data_ptr(), F, rows.data_ptr(), MAXD * 2, oc.data_ptr()),
    lambda ref, fr: oracle.ocsort_wrapper_step(ref, fr["dets"], 0.4), OC_ROW)

# ---- ByteTrack ----
BY = dict(track_thresh=0.6, track_buffer=30, match_thresh=0.8, frame_rate=30)
run("byte_track", lambda S: _lib.ByteTrackBank(**BY, min_confidence=0.4, wrapper_mode=True, n_streams=S, max_dets=MAXD, max_tracks=256),
    lambda: oracle.ByteTrack(**BY), pack_dets,
    lambda b, B, rows, oc: b.update_dev(B["dets"].data_ptr(), B["counts"].data_ptr(), F, rows.data_ptr(), MAXD * 2, oc.data_ptr()),
    lambda ref, fr: ref.update(fr["dets"][fr["dets"][:, 4] > 0.4]), _lib.BYTETRACK_ROW, low_conf_frac=0.2)

# ---- plain StrongSORT (D = 512, budget 100) ----
D = 512
SS = dict(max_dist=0.2, max_iou_dist=0.7, max_age=70, max_unmatched_preds=7, n_init=3, nn_budget=100, mc_lambda=0.995, ema_alpha=0.9)


def pack_ss(data, S):
    B = pack_dets(data, S)
    e = np.zeros((S, F, MAXD, D), np.float32)
    for s in range(S):
        for f, fr in enumerate(data[s]):
            e[s, f, :len(fr["dets"])] = fr["embeddings"][:, 0, :]
    B["feat"] = torch.from_numpy(e).cuda()
    return B


run("strong_sort", lambda S: _lib.SsortBank(D, **SS, min_confidence=0.4, wrapper_mode=True, n_streams=S, max_dets=MAXD, max_tracks=256),
    lambda: oracle.PlainStrongSORT(D, **SS), pack_ss,
    lambda b, B, rows, oc: b.update_dev(B["dets"].data_ptr(), B["feat"].data_ptr(), B["counts"].data_ptr(), F, rows.data_ptr(), MAXD * 2, oc.data_ptr()),
    lambda ref, fr: ref.update(fr["dets"][fr["dets"][:, 4] > 0.4], fr["embeddings"][:, 0, :][fr["dets"][:, 4] > 0.4]), _lib.SSORT_ROW,
    parts=1, dim=D, with_embeddings=True)

# ---- BoT-SORT (D = 512, cmc none) ----
BO = dict(track_high_thresh=0.6, new_track_thresh=0.7, track_buffer=30, match_thresh=0.8, proximity_thresh=0.5, appearance_thresh=0.25,
          frame_rate=30, lambda_=0.985)
run("bot_sort", lambda S: _lib.BoTSORTBank(D, **BO, min_confidence=0.4, wrapper_mode=True, n_streams=S, max_dets=MAXD, max_tracks=256),
    lambda: oracle.BoTSORT(D, **BO), pack_ss,
    lambda b, B, rows, oc: b.update_dev(B["dets"].data_ptr(), B["feat"].data_ptr(), B["counts"].data_ptr(), F, rows.data_ptr(), MAXD * 2, oc.data_ptr()),
    lambda ref, fr: ref.update(fr["dets"][fr["dets"][:, 4] > 0.4], fr["embeddings"][:, 0, :][fr["dets"][:, 4] > 0.4]), _lib.BOTSORT_ROW,
    parts=1, dim=D, with_embeddings=True, low_conf_frac=0.2)

# ---- Deep-OC-SORT (D = 512, cmc off; unit-norm detector embeddings) ----
DOC = dict(det_thresh=0, max_age=50, min_hits=1, iou_threshold=0.22136877277096445, delta_t=1, asso_func="giou", inertia=0.3941737016672115,
           w_association_emb=0.75, alpha_fixed_emb=0.95, aw_param=0.5, embedding_off=False, cmc_off=True, aw_off=False, new_kf_off=False)
DOC_ROW = np.dtype([("r", "<f8", (8,))])
run("deep_oc_sort", lambda S: _lib.DeepOCSortBank(D, **DOC, min_confidence=0.4, wrapper_mode=True, n_streams=S, max_dets=MAXD, max_tracks=256),
    lambda: oracle.DeepOCSort(D, **DOC), pack_ss,
    lambda b, B, rows, oc: b.update_dev(B["dets"].data_ptr(), B["feat"].data_ptr(), B["counts"].data_ptr(), F, rows.data_ptr(), MAXD * 2, oc.data_ptr()),
    lambda ref, fr: ref.update(fr["dets"][fr["dets"][:, 4] > 0.4], fr["embeddings"][:, 0, :][fr["dets"][:, 4] > 0.4]), DOC_ROW,
    parts=1, dim=D, with_embeddings=True)

# ---- BPBReID-StrongSORT (K = 6, D = 256) ----
K, DP = 6, 256
BP = dict(ema_alpha=0.9, mc_lambda=0.995, max_dist=0.5, motion_criterium="iou", max_iou_distance=0.8, max_age=300, n_init=0, nn_budget=100,
          min_bbox_confidence=0.0, only_position_for_kf_gating=False, max_kalman_prediction_without_update=7,
          matching_strategy="strong_sort_matching", gating_thres_factor=1, w_kfgd=1, w_reid=1, w_st=1)


def pack_bp(data, S):
    ids = np.zeros((S, F, MAXD), np.int64); ltwh = np.zeros((S, F, MAXD, 4)); emb = np.zeros((S, F, MAXD, K, DP), np.float32)
    vis = np.zeros((S, F, MAXD, K), np.uint8); conf = np.ones((S, F, MAXD)); c = np.zeros((S, F), np.int32)
    for s in range(S):
        for f, fr in enumerate(data[s]):
            d = fr["dets"]; n = len(d)
            ids[s, f, :n] = d[:, 6] + f * 1000; ltwh[s, f, :n] = ltrb_to_ltwh_rows(d[:, :4]); emb[s, f, :n] = fr["embeddings"]
            vis[s, f, :n] = fr["visibility"]; conf[s, f, :n] = d[:, 4]; c[s, f] = n
    return {k: torch.from_numpy(v).cuda() for k, v in dict(ids=ids, ltwh=ltwh, emb=emb, vis=vis, conf=conf, counts=c).items()}


run("bpbreid_strong_sort", lambda S: _lib.BpbssBank(K, DP, **BP, wrapper_mode=True, n_streams=S, max_dets=MAXD, max_tracks=256),
    lambda: oracle.StrongSORT(K, DP, **BP), pack_bp,
    lambda b, B, rows, oc: b.update_dev(B["ids"].data_ptr(), B["ltwh"].data_ptr(), B["emb"].data_ptr(), B["vis"].data_ptr(), B["conf"].data_ptr(),
                                        B["counts"].data_ptr(), F, rows.data_ptr(), MAXD * 2, oc.data_ptr()),
    lambda ref, fr: ref.update(fr["dets"][:, 6].astype(np.int64) + fr["frame"] * 1000, ltrb_to_ltwh_rows(fr["dets"][:, :4]), fr["embeddings"],
                               fr["visibility"], fr["dets"][:, 4]), _lib.BPBSS_ROW, parts=K, dim=DP, with_embeddings=True)

os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(results, open(os.path.join(REPO, "gpurun_out", "trackers.json"), "w"), indent=1)
