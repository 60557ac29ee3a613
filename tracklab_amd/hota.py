"""HOTA (Luiten et al.) as TrackEval computes it -- the accuracy half of BASELINE.json's metric.

Restates ``HOTA.eval_sequence`` / ``combine_sequences`` / ``_compute_final_fields`` of the TrackEval copy
vendored by the reference (plugins/eval/PoseTrack21/posetrack21/posetrack21/trackeval/metrics/hota.py:30-176;
the official path is pip ``trackeval`` through tracklab/wrappers/eval/trackeval_evaluator.py). Host-side numpy:
this is the metric definition, not a hot-path kernel. Sequences combine by summing TP/FN/FP and TP-weighting
AssA/AssRe/AssPr/LocA, which is what makes the per-epoch multi-GPU reduction a plain SUM all-reduce of
``pack(stats)`` (19 alphas x 7 sums + 2 scalars, ~1.1 KB).
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import linear_sum_assignment

ALPHAS = np.arange(0.05, 0.99, 0.05)
EPS = np.finfo("float").eps


def box_iou_matrix(a_ltrb: np.ndarray, b_ltrb: np.ndarray) -> np.ndarray:
    """TrackEval _calculate_box_ious (x0y0x1y1 format)."""
    if len(a_ltrb) == 0 or len(b_ltrb) == 0:
        return np.zeros((len(a_ltrb), len(b_ltrb)))
    a, b = a_ltrb[:, None, :], b_ltrb[None, :, :]
    min_ = np.minimum(a, b)
    max_ = np.maximum(a, b)
    inter = np.maximum(min_[..., 2] - max_[..., 0], 0) * np.maximum(min_[..., 3] - max_[..., 1], 0)
    area_a = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
    area_b = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    union = area_a + area_b - inter
    iou = np.zeros_like(inter)
    m = union > 0 + EPS
    iou[m] = inter[m] / union[m]
    return iou


def hota_sequence(gt_ids, tracker_ids, similarity):
    """gt_ids / tracker_ids: per-frame int arrays (ids already 0..n-1 contiguous); similarity: per-frame (g, t) IoU.
    Returns the per-sequence sufficient statistics (arrays over the 19 alphas)."""
    nA = len(ALPHAS)
    n_gt = int(max((int(g.max()) + 1 if len(g) else 0) for g in gt_ids)) if len(gt_ids) else 0
    n_tr = int(max((int(t.max()) + 1 if len(t) else 0) for t in tracker_ids)) if len(tracker_ids) else 0
    res = {k: np.zeros(nA) for k in ("HOTA_TP", "HOTA_FN", "HOTA_FP", "LocA_sum", "AssA", "AssRe", "AssPr")}
    num_gt_dets = sum(len(g) for g in gt_ids)
    num_tr_dets = sum(len(t) for t in tracker_ids)
    if num_tr_dets == 0:
        res["HOTA_FN"] += num_gt_dets
        return res
    if num_gt_dets == 0:
        res["HOTA_FP"] += num_tr_dets
        return res
    potential = np.zeros((n_gt, n_tr))
    gt_cnt = np.zeros((n_gt, 1))
    tr_cnt = np.zeros((1, n_tr))
    for g, t, sim in zip(gt_ids, tracker_ids, similarity):
        denom = sim.sum(0)[None, :] + sim.sum(1)[:, None] - sim
        sim_iou = np.zeros_like(sim)
        m = denom > 0 + EPS
        sim_iou[m] = sim[m] / denom[m]
        potential[g[:, None], t[None, :]] += sim_iou
        gt_cnt[g] += 1
        tr_cnt[0, t] += 1
    with np.errstate(invalid="ignore", divide="ignore"):
        gas = potential / (gt_cnt + tr_cnt - potential)
    matches = [np.zeros_like(potential) for _ in ALPHAS]
    for g, t, sim in zip(gt_ids, tracker_ids, similarity):
        if len(g) == 0:
            res["HOTA_FP"] += len(t)
            continue
        if len(t) == 0:
            res["HOTA_FN"] += len(g)
            continue
        score = gas[g[:, None], t[None, :]] * sim
        rows, cols = linear_sum_assignment(-score)
        for a, alpha in enumerate(ALPHAS):
            ok = sim[rows, cols] >= alpha - EPS
            r, c = rows[ok], cols[ok]
            n = len(r)
            res["HOTA_TP"][a] += n
            res["HOTA_FN"][a] += len(g) - n
            res["HOTA_FP"][a] += len(t) - n
            if n > 0:
                res["LocA_sum"][a] += sum(sim[r, c])
                matches[a][g[r], t[c]] += 1
    for a in range(nA):
        mc = matches[a]
        tp = np.maximum(1, res["HOTA_TP"][a])
        res["AssA"][a] = np.sum(mc * (mc / np.maximum(1, gt_cnt + tr_cnt - mc))) / tp
        res["AssRe"][a] = np.sum(mc * (mc / np.maximum(1, gt_cnt))) / tp
        res["AssPr"][a] = np.sum(mc * (mc / np.maximum(1, tr_cnt))) / tp
    return res


def pack(res, frames: float = 0.0, seconds: float = 0.0) -> np.ndarray:
    """Sufficient statistics as one float64 vector that SUM-reduces across sequences / ranks:
    TP, FN, FP, LocA_sum and the TP-weighted AssA/AssRe/AssPr (combine_sequences, hota.py:157-169)."""
    tp = res["HOTA_TP"]
    return np.concatenate([tp, res["HOTA_FN"], res["HOTA_FP"], res["LocA_sum"],
                           res["AssA"] * tp, res["AssRe"] * tp, res["AssPr"] * tp, [frames, seconds]])


def finalize(vec: np.ndarray) -> dict:
    """Summed `pack` vector -> HOTA fields (_compute_final_fields, hota.py:171-186; scalars = mean over alphas)."""
    n = len(ALPHAS)
    tp, fn, fp, loc, assa, assre, asspr = (vec[i * n:(i + 1) * n] for i in range(7))
    one = np.maximum(1e-10, tp)
    out = {"HOTA_TP": tp, "HOTA_FN": fn, "HOTA_FP": fp,
           "AssA": assa / one, "AssRe": assre / one, "AssPr": asspr / one,
           "LocA": np.maximum(1e-10, loc) / one}
    out["DetRe"] = tp / np.maximum(1, tp + fn)
    out["DetPr"] = tp / np.maximum(1, tp + fp)
    out["DetA"] = tp / np.maximum(1, tp + fn + fp)
    out["HOTA"] = np.sqrt(out["DetA"] * out["AssA"])
    out["frames"], out["seconds"] = float(vec[7 * n]), float(vec[7 * n + 1])
    out["summary"] = {k: float(np.mean(out[k])) for k in ("HOTA", "DetA", "AssA", "DetRe", "DetPr", "AssRe", "AssPr", "LocA")}
    return out


def sequence_from_rows(gt_frames, tracker_frames):
    """gt_frames: list of (ids (n,), ltrb (n,4)); tracker_frames: list of (track_ids (m,), ltrb (m,4)).
    Re-labels ids to 0..n-1 like TrackEval's preprocessing and builds the IoU similarity per frame."""
    gmap, tmap = {}, {}
    gi, ti, sims = [], [], []
    for (g, gb), (t, tb) in zip(gt_frames, tracker_frames):
        gi.append(np.array([gmap.setdefault(int(x), len(gmap)) for x in g], dtype=int))
        ti.append(np.array([tmap.setdefault(int(x), len(tmap)) for x in t], dtype=int))
        sims.append(box_iou_matrix(np.asarray(gb, dtype=float).reshape(-1, 4), np.asarray(tb, dtype=float).reshape(-1, 4)))
    return gi, ti, sims


def hota_sequence_gpu(gt_frames, tracker_frames):
    """`hota_sequence(*sequence_from_rows(...))` on the device (tlk_hota_sequence_f64: similarity, global alignment, per-frame Hungarian
    matching and the 19-threshold counts as four kernel launches). Same dict of per-threshold arrays; no CPU fallback."""
    import ctypes as C
    from . import _lib
    L = _lib.lib()
    gmap, tmap = {}, {}
    gid, tid, gb, tb, goff, toff = [], [], [], [], [0], [0]
    for (g, gbox), (t, tbox) in zip(gt_frames, tracker_frames):
        gid.extend(gmap.setdefault(int(x), len(gmap)) for x in g)
        tid.extend(tmap.setdefault(int(x), len(tmap)) for x in t)
        gb.append(np.asarray(gbox, dtype=np.float64).reshape(-1, 4)); tb.append(np.asarray(tbox, dtype=np.float64).reshape(-1, 4))
        goff.append(len(gid)); toff.append(len(tid))
    gid, tid = np.asarray(gid, dtype=np.int32), np.asarray(tid, dtype=np.int32)
    gb = np.ascontiguousarray(np.concatenate(gb)) if gb else np.zeros((0, 4))
    tb = np.ascontiguousarray(np.concatenate(tb)) if tb else np.zeros((0, 4))
    goff, toff = np.asarray(goff, dtype=np.int64), np.asarray(toff, dtype=np.int64)
    stats = np.zeros((7, len(ALPHAS)))
    alphas = np.ascontiguousarray(ALPHAS, dtype=np.float64)
    vp = C.c_void_p
    L.tlk_hota_sequence_f64.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]
    _lib.check(L.tlk_hota_sequence_f64(gid.ctypes.data, gb.ctypes.data, goff.ctypes.data, tid.ctypes.data, tb.ctypes.data, toff.ctypes.data,
                                       len(goff) - 1, len(gmap), len(tmap), alphas.ctypes.data, stats.ctypes.data))
    return {k: stats[i].copy() for i, k in enumerate(("HOTA_TP", "HOTA_FN", "HOTA_FP", "LocA_sum", "AssA", "AssRe", "AssPr"))}
