"""CLEAR-MOT and ID measures from in-memory tables, restated from the py-motmetrics copy the reference vendors for its
PoseTrack21 MOT evaluator (plugins/eval/PoseTrack21/posetrack21_mot/posetrack21_mot/motmetrics: mot.py:134-345 the event
accumulator, metrics.py:342-728 the measures, distances.py:52-129 the IoU distance, lap.py:79-130 the NaN-edge handling around
scipy's linear_sum_assignment). No pandas event frame: a sequence is reduced on the fly to

* summable counts (frames, matches, switches, misses, false positives, transfers / ascends / migrates, objects, predictions,
  sum of matched distances, mostly / partially tracked / mostly lost, fragmentations, IDTP / IDFP / IDFN), and
* ratios derived from the sums (MOTA, MOTP, precision, recall, IDP, IDR, IDF1),

so that several sequences -- or the per-rank partial results of a multi-GPU run, after one SUM all-reduce of ``pack()`` -- combine
exactly like motmetrics' ``compute_many(generate_overall=True)`` (metrics.py ``*_m`` rules).
Host-side numpy: the evaluator is not on the hot path; per-frame work is one IoU matrix and one LSA.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import linear_sum_assignment as _scipy_lsa

SUM_FIELDS = ["num_frames", "num_matches", "num_switches", "num_transfer", "num_ascend", "num_migrate", "num_false_positives", "num_misses",
              "num_objects", "num_predictions", "num_unique_objects", "mostly_tracked", "partially_tracked", "mostly_lost", "num_fragmentations",
              "sum_distance", "idtp", "idfp", "idfn"]


def iou_distance_matrix(objs_ltwh, hyps_ltwh, max_iou: float = 1.0) -> np.ndarray:
    """distances.iou_matrix (distances.py:83-129): 1 - IoU of (x, y, w, h) rectangles, NaN above ``max_iou``."""
    if np.size(objs_ltwh) == 0 or np.size(hyps_ltwh) == 0:
        return np.empty((0, 0))
    a = np.asarray(objs_ltwh, dtype=float)[:, None, :]
    b = np.asarray(hyps_ltwh, dtype=float)[None, :, :]
    a_min, a_max = a[..., :2], a[..., :2] + a[..., 2:]
    b_min, b_max = b[..., :2], b[..., :2] + b[..., 2:]
    i_vol = np.prod(np.maximum(np.minimum(a_max, b_max) - np.maximum(a_min, b_min), 0), axis=-1)
    a_vol = np.prod(np.maximum(a_max - a_min, 0), axis=-1)
    b_vol = np.prod(np.maximum(b_max - b_min, 0), axis=-1)
    u_vol = a_vol + b_vol - i_vol
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = np.where(i_vol == 0, 0.0, np.true_divide(i_vol, u_vol))
    dist = 1.0 - iou
    return np.where(dist > max_iou, np.nan, dist)


def lsa_with_missing_edges(costs: np.ndarray):
    """lap.lsa_solve_scipy (lap.py:79-130): non-finite entries become 2 r c + 1, assigned pairs on such entries are dropped."""
    costs = np.asarray(costs, dtype=float)
    if not costs.size:
        return np.array([], dtype=int), np.array([], dtype=int)
    valid = np.isfinite(costs)
    if valid.all():
        finite = costs
    elif not valid.any():
        finite = np.zeros_like(costs)
    else:
        c = np.abs(costs[valid]).max() + 1
        finite = np.where(valid, costs, 2 * min(costs.shape) * c + 1)
    r, c = _scipy_lsa(finite)
    keep = valid[r, c]
    return r[keep], c[keep]


class MOTAccumulator:
    """mot.MOTAccumulator(auto_id=True).update restated (mot.py:134-345); ``update`` takes what the reference's takes."""

    def __init__(self, max_switch_time: float = float("inf")):
        self.max_switch_time = max_switch_time
        self.m, self.res_m, self.last_occurrence, self.last_match, self.hyp_history = {}, {}, {}, {}, {}
        self.frame = -1
        self.c = dict.fromkeys(SUM_FIELDS, 0)
        self.c["sum_distance"] = 0.0
        self._obj_events = {}              # oid -> list of 0 (matched / switched) or 1 (missed), in time order
        self._ocs, self._hcs, self._tps = {}, {}, {}

    def update(self, oids, hids, dists):
        self.frame += 1
        fid = self.frame
        oids, hids = np.asarray(oids), np.asarray(hids)
        no, nh = len(oids), len(hids)
        dists = np.atleast_2d(np.asarray(dists, dtype=float)).astype(float).reshape(no, nh).copy()
        c = self.c
        c["num_frames"] += 1
        for o in oids.tolist():            # RAW events: every present object / hypothesis once per frame, every finite pair
            self._ocs[o] = self._ocs.get(o, 0) + 1
        for h in hids.tolist():
            self._hcs[h] = self._hcs.get(h, 0) + 1
        vi, vj = np.where(np.isfinite(dists))
        for i, j in zip(vi.tolist(), vj.tolist()):
            k = (oids[i].item(), hids[j].item())
            self._tps[k] = self._tps.get(k, 0) + 1
        om, hm = np.zeros(no, bool), np.zeros(nh, bool)

        def matched(o, d):
            c["sum_distance"] += d
            self._obj_events.setdefault(o, []).append(0)

        if no * nh > 0:
            for i in range(no):            # 1. carry established correspondences forward
                o = oids[i].item()
                if o not in self.m:
                    continue
                j = np.where(~hm & (hids == self.m[o]))[0]
                if j.shape[0] == 0:
                    continue
                j = int(j[0])
                if np.isfinite(dists[i, j]):
                    h = hids[j].item()
                    om[i], hm[j] = True, True
                    self.m[o] = h
                    c["num_matches"] += 1
                    matched(o, dists[i, j])
                    self.last_match[o] = fid
                    self.hyp_history[h] = fid
            dists[om, :] = np.nan          # 2. minimum-cost assignment of the rest
            dists[:, hm] = np.nan
            rids, cids = lsa_with_missing_edges(dists)
            for i, j in zip(rids.tolist(), cids.tolist()):
                if not np.isfinite(dists[i, j]):
                    continue
                o, h = oids[i].item(), hids[j].item()
                is_switch = o in self.m and self.m[o] != h and abs(fid - self.last_occurrence[o]) <= self.max_switch_time
                if is_switch and h not in self.hyp_history:
                    c["num_ascend"] += 1
                if h in self.res_m and self.res_m[h] != o:
                    if o not in self.last_match:
                        c["num_migrate"] += 1
                    c["num_transfer"] += 1
                self.hyp_history[h] = fid
                self.last_match[o] = fid
                c["num_switches" if is_switch else "num_matches"] += 1
                matched(o, dists[i, j])
                om[i], hm[j] = True, True
                self.m[o] = h
                self.res_m[h] = o
        for o in oids[~om].tolist():       # 3. misses
            c["num_misses"] += 1
            self._obj_events.setdefault(o, []).append(1)
        c["num_false_positives"] += int((~hm).sum())          # 4. false alarms
        for o in oids.tolist():            # 5. occurrence state
            self.last_occurrence[o] = fid

    def update_boxes(self, oids, gt_ltwh, hids, hyp_ltwh, max_iou: float = 0.5):
        """One frame from boxes: evaluate_mot.py:130 (IoU distance, pairs above ``max_iou`` may not be matched)."""
        self.update(oids, hids, iou_distance_matrix(gt_ltwh, hyp_ltwh, max_iou=max_iou))

    def counts(self) -> dict:
        """The summable counts of the sequence so far (track ratios, fragmentations and the ID assignment are per sequence)."""
        c = dict(self.c)
        c["num_objects"] = sum(self._ocs.values())
        c["num_predictions"] = sum(self._hcs.values())
        c["num_unique_objects"] = len(self._ocs)
        mt = pt = ml = frag = 0
        for o, n in self._ocs.items():
            ev = np.asarray(self._obj_events.get(o, []))
            ratio = float((ev == 0).sum()) / n
            mt += ratio >= 0.8
            pt += 0.2 <= ratio < 0.8
            ml += ratio < 0.2
            hit = np.where(ev == 0)[0]
            if len(hit):                   # switches from tracked to missed inside [first hit, last hit] (metrics.py:492-506)
                span = ev[hit[0]:hit[-1] + 1]
                frag += int((np.diff(span) == 1).sum())
        c["mostly_tracked"], c["partially_tracked"], c["mostly_lost"], c["num_fragmentations"] = int(mt), int(pt), int(ml), frag
        # id_global_assignment (metrics.py:610-653)
        oids, hids = sorted(self._ocs), sorted(self._hcs)
        no, nh = len(oids), len(hids)
        fp = np.zeros((no + nh, no + nh)); fn = np.zeros((no + nh, no + nh))
        fp[no:, :nh] = np.nan; fn[:no, nh:] = np.nan
        oi = {o: i for i, o in enumerate(oids)}; hi = {h: i for i, h in enumerate(hids)}
        for o, oc in self._ocs.items():
            fn[oi[o], :nh] = oc; fn[oi[o], nh + oi[o]] = oc
        for h, hc in self._hcs.items():
            fp[:no, hi[h]] = hc; fp[hi[h] + no, hi[h]] = hc
        for (o, h), ex in self._tps.items():
            fp[oi[o], hi[h]] -= ex; fn[oi[o], hi[h]] -= ex
        r, cc = lsa_with_missing_edges(fp + fn)
        c["idfp"], c["idfn"] = float(fp[r, cc].sum()), float(fn[r, cc].sum())
        c["idtp"] = c["num_objects"] - c["idfn"]
        return c

    def metrics(self) -> dict:
        return finalize(self.counts())


def _div(a, b):
    with np.errstate(divide="ignore", invalid="ignore"):
        return float(np.true_divide(a, b))


def finalize(c: dict) -> dict:
    """Counts (of one sequence, or summed over sequences / ranks) -> counts + the ratio measures (metrics.py:512-569, :685-716)."""
    out = dict(c)
    det = c["num_matches"] + c["num_switches"]
    out["num_detections"] = det
    out["motp"] = _div(c["sum_distance"], det)
    out["mota"] = 1.0 - _div(c["num_misses"] + c["num_switches"] + c["num_false_positives"], c["num_objects"])
    out["precision"] = _div(det, c["num_false_positives"] + det)
    out["recall"] = _div(det, c["num_objects"])
    out["idp"] = _div(c["idtp"], c["idtp"] + c["idfp"])
    out["idr"] = _div(c["idtp"], c["idtp"] + c["idfn"])
    out["idf1"] = _div(2 * c["idtp"], c["num_objects"] + c["num_predictions"])
    return out


def sequence_counts_gpu(frames, max_iou: float = 0.5) -> dict:
    """The counts of one sequence on the device (tlk_clear_sequence_f64: one workgroup walks the frames -- the accumulator is sequential --
    and solves the global ID assignment at the end). frames: iterable of (gt_ids, gt_ltwh, hyp_ids, hyp_ltwh), what ``update_boxes``
    takes. Same dict as ``MOTAccumulator.counts()``; ``finalize`` / ``merge`` / ``pack`` apply unchanged. No CPU fallback."""
    import ctypes as C
    from . import _lib
    gi, hi, gb, hb, goff, hoff = [], [], [], [], [0], [0]
    for g, gbox, h, hbox in frames:
        gi.extend(int(x) for x in g); hi.extend(int(x) for x in h)
        gb.append(np.asarray(gbox, dtype=np.float64).reshape(-1, 4)); hb.append(np.asarray(hbox, dtype=np.float64).reshape(-1, 4))
        goff.append(len(gi)); hoff.append(len(hi))
    # dense ids in the SORTED order of the original ids: the ID assignment's matrix is ordered by id (metrics.py:616-617)
    gu, gd = np.unique(np.asarray(gi, dtype=np.int64), return_inverse=True) if gi else (np.zeros(0, np.int64), np.zeros(0, np.int64))
    hu, hd = np.unique(np.asarray(hi, dtype=np.int64), return_inverse=True) if hi else (np.zeros(0, np.int64), np.zeros(0, np.int64))
    gd, hd = np.ascontiguousarray(gd, dtype=np.int32), np.ascontiguousarray(hd, dtype=np.int32)
    gb = np.ascontiguousarray(np.concatenate(gb)) if gb else np.zeros((0, 4))
    hb = np.ascontiguousarray(np.concatenate(hb)) if hb else np.zeros((0, 4))
    goff, hoff = np.asarray(goff, dtype=np.int64), np.asarray(hoff, dtype=np.int64)
    out = np.zeros(len(SUM_FIELDS))
    vp = C.c_void_p
    L = _lib.lib()
    L.tlk_clear_sequence_f64.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, vp]
    _lib.check(L.tlk_clear_sequence_f64(gd.ctypes.data, gb.ctypes.data, goff.ctypes.data, hd.ctypes.data, hb.ctypes.data, hoff.ctypes.data,
                                        len(goff) - 1, len(gu), len(hu), float(max_iou), out.ctypes.data))
    c = {k: (float(v) if k in ("sum_distance", "idtp", "idfp", "idfn") else int(v)) for k, v in zip(SUM_FIELDS, out)}
    return c


def pack(c: dict) -> np.ndarray:
    """Counts as a float64 vector for a SUM all-reduce over ranks (tracklab_amd.dist)."""
    return np.array([float(c[k]) for k in SUM_FIELDS], dtype=np.float64)


def unpack(v) -> dict:
    return {k: float(x) for k, x in zip(SUM_FIELDS, np.asarray(v, dtype=np.float64))}


def merge(counts_list) -> dict:
    """compute_many(..., generate_overall=True): sum the counts, then the ratios from the sums."""
    tot = np.sum([pack(c) for c in counts_list], axis=0) if len(counts_list) else np.zeros(len(SUM_FIELDS))
    return finalize(unpack(tot))
