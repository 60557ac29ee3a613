"""TrackEval's CLEAR and Identity metric families and TrackEval's result layout, for the evaluator plugin (wrappers/eval.py).

The reference evaluates through pip ``trackeval`` (tracklab/wrappers/eval/trackeval_evaluator.py:80-104: ``cfg.metrics`` = CLEAR, HOTA, Identity of
``trackeval.metrics``, configs/eval/trackeval.yaml:11-14, THRESHOLD 0.5).  HOTA is restated in tracklab_amd/hota.py and pinned against the copy of
TrackEval's hota.py the reference vendors (plugins/eval/PoseTrack21/.../trackeval/metrics/hota.py).  CLEAR and Identity are NOT in that vendored
copy and pip ``trackeval`` is absent from this image: what follows restates the published algorithm of TrackEval 1.0.dev1
(``trackeval/metrics/clear.py::CLEAR.eval_sequence / combine_sequences / _compute_final_fields`` and ``identity.py::Identity``) -- PARITY UNPINNED
against TrackEval itself; pinned here by brute-force optimality on small sequences and by agreement with the py-motmetrics restatement
(tracklab_amd/clearmot.py, which IS pinned against the motmetrics copy the reference vendors) wherever the two definitions coincide
(tests/test_trackeval_metrics.py).

How TrackEval's CLEAR differs from py-motmetrics' (why tracklab_amd.clearmot is not a substitute):
  * per frame ONE Hungarian assignment over score = 1000 * [tracker id == the id this gt was matched to in the PREVIOUS frame] + IoU, entries
    with IoU < 0.5 zeroed, pairs with score 0 dropped -- motmetrics first keeps every still-valid previous pair, then solves the rest;
  * an id switch is counted against the last id the gt was EVER matched to (``prev_tracker_id``), continuity preference only looks one frame back;
  * MT / ML thresholds are > 0.8 and < 0.2 of the gt's frames (motmetrics: >= 0.8, <= 0.2); Frag counts tracked -> untracked -> tracked resumptions.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import linear_sum_assignment

from . import hota as _hota

EPS = np.finfo("float").eps
CLEAR_INT = ("CLR_TP", "CLR_FN", "CLR_FP", "IDSW", "MT", "PT", "ML", "Frag", "CLR_Frames")
CLEAR_SUMMED = CLEAR_INT + ("MOTP_sum",)
ID_INT = ("IDTP", "IDFN", "IDFP")


def clear_eval_sequence(gt_ids, tracker_ids, similarity, threshold: float = 0.5) -> dict:
    """gt_ids / tracker_ids: per-frame int arrays with ids 0..n-1 (hota.sequence_from_rows); similarity: per-frame (g, t) IoU.
    -> the summed fields of CLEAR.eval_sequence (integers + MOTP_sum)."""
    res = {k: 0 for k in CLEAR_SUMMED}
    res["MOTP_sum"] = 0.0
    n_frames = len(gt_ids)
    num_gt_dets = sum(len(g) for g in gt_ids)
    num_tr_dets = sum(len(t) for t in tracker_ids)
    num_gt_ids = int(max([int(g.max()) + 1 for g in gt_ids if len(g)] + [0]))
    if num_tr_dets == 0:                                      # (TrackEval's early returns leave CLR_Frames at 0: only the full path sets it)
        res["CLR_FN"] = num_gt_dets
        res["ML"] = num_gt_ids
        return res
    if num_gt_dets == 0:
        res["CLR_FP"] = num_tr_dets
        return res
    gt_id_count = np.zeros(num_gt_ids)
    gt_matched_count = np.zeros(num_gt_ids)
    gt_frag_count = np.zeros(num_gt_ids)
    prev_tracker_id = np.full(num_gt_ids, np.nan)             # for scoring IDSW: the last tracker id this gt was ever matched to
    prev_timestep_tracker_id = np.full(num_gt_ids, np.nan)    # for matching: the tracker id of the previous frame only
    for g, t, sim in zip(gt_ids, tracker_ids, similarity):
        if len(g) == 0:
            res["CLR_FP"] += len(t)
            continue
        if len(t) == 0:
            res["CLR_FN"] += len(g)
            gt_id_count[g] += 1
            continue
        score = (t[np.newaxis, :] == prev_timestep_tracker_id[g[:, np.newaxis]])
        score = 1000 * score + sim
        score[sim < threshold - EPS] = 0
        rows, cols = linear_sum_assignment(-score)
        keep = score[rows, cols] > 0 + EPS
        rows, cols = rows[keep], cols[keep]
        mg, mt = g[rows], t[cols]
        prev = prev_tracker_id[mg]
        res["IDSW"] += int(np.sum(np.logical_not(np.isnan(prev)) & np.not_equal(mt, prev)))
        gt_id_count[g] += 1
        gt_matched_count[mg] += 1
        not_previously_tracked = np.isnan(prev_timestep_tracker_id)
        prev_tracker_id[mg] = mt
        prev_timestep_tracker_id[:] = np.nan
        prev_timestep_tracker_id[mg] = mt
        currently_tracked = np.logical_not(np.isnan(prev_timestep_tracker_id))
        gt_frag_count += np.logical_and(not_previously_tracked, currently_tracked)
        n = len(mg)
        res["CLR_TP"] += n
        res["CLR_FN"] += len(g) - n
        res["CLR_FP"] += len(t) - n
        if n > 0:
            res["MOTP_sum"] += float(sum(sim[rows, cols]))
    seen = gt_id_count > 0
    ratio = gt_matched_count[seen] / gt_id_count[seen]
    res["MT"] = int(np.sum(np.greater(ratio, 0.8)))
    res["PT"] = int(np.sum(np.greater_equal(ratio, 0.2))) - res["MT"]
    res["ML"] = num_gt_ids - res["MT"] - res["PT"]
    res["Frag"] = int(np.sum(np.subtract(gt_frag_count[gt_frag_count > 0], 1)))
    res["CLR_Frames"] = n_frames
    return res


def clear_final(res: dict) -> dict:
    """CLEAR._compute_final_fields on summed fields (one sequence or COMBINED_SEQ)."""
    out = dict(res)
    num_gt_ids = res["MT"] + res["ML"] + res["PT"]
    tp, fn, fp, idsw = res["CLR_TP"], res["CLR_FN"], res["CLR_FP"], res["IDSW"]
    out["MTR"] = res["MT"] / np.maximum(1.0, num_gt_ids)
    out["MLR"] = res["ML"] / np.maximum(1.0, num_gt_ids)
    out["PTR"] = res["PT"] / np.maximum(1.0, num_gt_ids)
    out["CLR_Re"] = tp / np.maximum(1.0, tp + fn)
    out["CLR_Pr"] = tp / np.maximum(1.0, tp + fp)
    out["MODA"] = (tp - fp) / np.maximum(1.0, tp + fn)
    out["MOTA"] = (tp - fp - idsw) / np.maximum(1.0, tp + fn)
    out["MOTP"] = res["MOTP_sum"] / np.maximum(1.0, tp)
    out["sMOTA"] = (res["MOTP_sum"] - fp - idsw) / np.maximum(1.0, tp + fn)
    out["CLR_F1"] = tp / np.maximum(1.0, tp + 0.5 * fn + 0.5 * fp)
    out["FP_per_frame"] = fp / np.maximum(1.0, res["CLR_Frames"])
    safe_log_idsw = np.log10(idsw) if idsw > 0 else idsw
    out["MOTAL"] = (tp - fp - safe_log_idsw) / np.maximum(1.0, tp + fn)
    return {k: (float(v) if not isinstance(v, (int, np.integer)) else int(v)) for k, v in out.items()}


def identity_eval_sequence(gt_ids, tracker_ids, similarity, threshold: float = 0.5) -> dict:
    """Identity.eval_sequence: the global one-to-one id assignment that minimises IDFN + IDFP."""
    num_gt_dets = sum(len(g) for g in gt_ids)
    num_tr_dets = sum(len(t) for t in tracker_ids)
    if num_tr_dets == 0:
        return {"IDTP": 0, "IDFN": num_gt_dets, "IDFP": 0}
    if num_gt_dets == 0:
        return {"IDTP": 0, "IDFN": 0, "IDFP": num_tr_dets}
    ng = int(max(int(g.max()) + 1 for g in gt_ids if len(g)))
    nt = int(max(int(t.max()) + 1 for t in tracker_ids if len(t)))
    potential = np.zeros((ng, nt))
    gt_cnt, tr_cnt = np.zeros(ng), np.zeros(nt)
    for g, t, sim in zip(gt_ids, tracker_ids, similarity):
        gi, ti = np.nonzero(np.greater_equal(sim, threshold)) if len(g) and len(t) else (np.zeros(0, int), np.zeros(0, int))
        potential[g[gi], t[ti]] += 1
        gt_cnt[g] += 1
        tr_cnt[t] += 1
    fp_mat, fn_mat = np.zeros((ng + nt, ng + nt)), np.zeros((ng + nt, ng + nt))
    fp_mat[ng:, :nt] = 1e10
    fn_mat[:ng, nt:] = 1e10
    for i in range(ng):
        fn_mat[i, :nt] = gt_cnt[i]
        fn_mat[i, nt + i] = gt_cnt[i]
    for j in range(nt):
        fp_mat[:ng, j] = tr_cnt[j]
        fp_mat[j + ng, j] = tr_cnt[j]
    fn_mat[:ng, :nt] -= potential
    fp_mat[:ng, :nt] -= potential
    rows, cols = linear_sum_assignment(fn_mat + fp_mat)
    idfn = int(fn_mat[rows, cols].sum())
    idfp = int(fp_mat[rows, cols].sum())
    return {"IDTP": int(gt_cnt.sum() - idfn), "IDFN": idfn, "IDFP": idfp}


def identity_final(res: dict) -> dict:
    tp, fn, fp = res["IDTP"], res["IDFN"], res["IDFP"]
    return {"IDTP": int(tp), "IDFN": int(fn), "IDFP": int(fp), "IDR": float(tp / np.maximum(1.0, tp + fn)), "IDP": float(tp / np.maximum(1.0, tp + fp)),
            "IDF1": float(tp / np.maximum(1.0, tp + 0.5 * fp + 0.5 * fn))}


def hota_fields(packed: np.ndarray) -> dict:
    """tracklab_amd.hota's packed statistics -> TrackEval's HOTA field dict (arrays over the 19 alphas + the (0) fields)."""
    f = _hota.finalize(packed)
    out = {k: np.asarray(f[k]) for k in ("HOTA", "DetA", "AssA", "DetRe", "DetPr", "AssRe", "AssPr", "LocA", "HOTA_TP", "HOTA_FN", "HOTA_FP")}
    out["OWTA"] = np.sqrt(out["DetRe"] * out["AssA"])
    out["HOTA(0)"], out["LocA(0)"] = float(out["HOTA"][0]), float(out["LocA"][0])
    out["HOTALocA(0)"] = out["HOTA(0)"] * out["LocA(0)"]
    return out


# TrackEval's `summary_fields` per metric family, in its order (CLEAR: main float + main integer fields -- the "extra" fields CLR_F1, FP_per_frame,
# MOTAL, MOTP_sum, CLR_Frames are per-sequence detail, not summary; HOTA: the float-array fields + the (0) fields; Identity: all six)
SUMMARY_FIELDS = {
    "CLEAR": ("MOTA", "MOTP", "MODA", "CLR_Re", "CLR_Pr", "MTR", "PTR", "MLR", "sMOTA", "CLR_TP", "CLR_FN", "CLR_FP", "IDSW", "MT", "PT", "ML", "Frag"),
    "HOTA": ("HOTA", "DetA", "AssA", "DetRe", "DetPr", "AssRe", "AssPr", "LocA", "OWTA", "HOTA(0)", "LocA(0)", "HOTALocA(0)"),
    "Identity": ("IDF1", "IDR", "IDP", "IDTP", "IDFN", "IDFP"),
}


def _summary(family: str, fields: dict) -> dict:
    """_BaseMetric.summary_results over the family's summary_fields: floats as percentages "{0:1.5g}", integers "%d", float arrays by their
    mean over the alphas."""
    out = {}
    for k in SUMMARY_FIELDS.get(family, tuple(fields)):
        if k not in fields:
            continue
        v = fields[k]
        if isinstance(v, np.ndarray):
            out[k] = "{0:1.5g}".format(100 * float(np.mean(v)))
        elif isinstance(v, (int, np.integer)):
            out[k] = "%d" % v
        else:
            out[k] = "{0:1.5g}".format(100 * float(v))
    return out


def evaluate_sequence_frames(gt_frames, tracker_frames, threshold: float = 0.5) -> dict:
    """gt_frames / tracker_frames: per-frame (ids, ltrb) as evaluate.evaluate_sequence builds them -> summed CLEAR + Identity fields."""
    g, t, sims = _hota.sequence_from_rows(gt_frames, tracker_frames)
    return {"CLEAR": clear_eval_sequence(g, t, sims, threshold), "Identity": identity_eval_sequence(g, t, sims, threshold)}


def trackeval_layout(per_seq: dict, cls: str = "pedestrian", metrics=("CLEAR", "HOTA", "Identity")) -> dict:
    """{sequence: {"hota": packed vector, "CLEAR": summed fields, "Identity": summed fields}} -> TrackEval's
    ``output_res[dataset][tracker]`` = {sequence: {cls: {family: fields}}, "COMBINED_SEQ": {...}, "SUMMARIES": {cls: {family: {field: str}}}}
    (the layout tracklab's ``process_trackeval_results`` hooks read: wrappers/dataset/mot_like/common.py:242-258)."""
    def families(h, c, i):
        out = {}
        if "HOTA" in metrics:
            out["HOTA"] = hota_fields(h)
        if "CLEAR" in metrics:
            out["CLEAR"] = clear_final(c)
        if "Identity" in metrics:
            out["Identity"] = identity_final(i)
        return out
    res = {name: {cls: families(r["hota"], r["CLEAR"], r["Identity"])} for name, r in per_seq.items()}
    if per_seq:
        hsum = np.sum([r["hota"] for r in per_seq.values()], axis=0)
        csum = {k: sum(r["CLEAR"][k] for r in per_seq.values()) for k in CLEAR_SUMMED}
        isum = {k: sum(r["Identity"][k] for r in per_seq.values()) for k in ID_INT}
        comb = families(hsum, csum, isum)
        res["COMBINED_SEQ"] = {cls: comb}
        res["SUMMARIES"] = {cls: {fam: _summary(fam, fields) for fam, fields in comb.items()}}
    return res
