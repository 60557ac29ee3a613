#!/usr/bin/env python3
"""Differential fuzz of tracklab_amd.hota / tracklab_amd.clearmot (the host restatements the device evaluators are pinned on) against the copies of
TrackEval's HOTA and py-motmetrics the reference vendors (plugins/eval/PoseTrack21/...), on random sequences: ground truth vs jittered, dropped,
id-swapped, spurious hypotheses. Runs only where /root/reference exists (like fuzz_reference.py); results go to profiles/.
    python tests/golden/fuzz_eval_reference.py [trials]"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = "/root/reference"
from tracklab_amd import clearmot, hota                                   # noqa: E402
from tracklab_amd.synth import SyntheticStream                            # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100

base = os.path.join(REF, "plugins", "eval", "PoseTrack21", "posetrack21", "posetrack21", "trackeval")
pkg = types.ModuleType("trackeval"); pkg.__path__ = [base]; sys.modules["trackeval"] = pkg
mpkg = types.ModuleType("trackeval.metrics"); mpkg.__path__ = [os.path.join(base, "metrics")]; sys.modules["trackeval.metrics"] = mpkg


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m


load("trackeval._timing", os.path.join(base, "_timing.py"))
load("trackeval.utils", os.path.join(base, "utils.py"))
load("trackeval.metrics._base_metric", os.path.join(base, "metrics", "_base_metric.py"))
H = load("trackeval.metrics.hota", os.path.join(base, "metrics", "hota.py")).HOTA()
sys.path.insert(0, os.path.join(REF, "plugins", "eval", "PoseTrack21", "posetrack21_mot", "posetrack21_mot"))
if not hasattr(np, "asfarray"):
    np.asfarray = lambda a, dtype=float: np.asarray(a, dtype=dtype)
sys.modules.setdefault("xmltodict", types.ModuleType("xmltodict"))
import motmetrics as mm                                                   # noqa: E402
mm.lap.default_solver = "scipy"
CM = ["num_frames", "num_matches", "num_switches", "num_transfer", "num_ascend", "num_migrate", "num_false_positives", "num_misses", "num_objects",
      "num_predictions", "num_unique_objects", "mostly_tracked", "partially_tracked", "mostly_lost", "num_fragmentations", "idtp", "idfp", "idfn",
      "motp", "mota", "precision", "recall", "idp", "idr", "idf1"]

ok_h = ok_c = 0
worst_h = worst_c = 0.0
for t in range(N):
    rng = np.random.default_rng(31000 + t)
    nobj, nfr = int(rng.integers(2, 40)), int(rng.integers(5, 90))
    jitter, drop, swap, spur = float(rng.choice([0.5, 3.0, 10.0, 25.0])), float(rng.choice([0.0, 0.1, 0.4])), float(rng.choice([0.0, 0.03, 0.1])), float(rng.choice([0.0, 0.5, 2.0]))
    perm, gmap, tmap = {}, {}, {}
    gt_ids, tr_ids, sims, gt_fr, tr_fr = [], [], [], [], []
    acc_ref, acc = mm.MOTAccumulator(auto_id=True), clearmot.MOTAccumulator()
    for fr in SyntheticStream(700 + t, nobj, nfr, miss_prob=0.0, churn_period=int(rng.choice([7, 30, 1000]))):
        gid = fr["gt_all_ids"].astype(np.int64); gb = np.asarray(fr["gt_boxes"], dtype=np.float64)
        if rng.random() < 0.05:
            gid, gb = gid[:0], gb[:0]                                  # a frame without ground truth
        keep = rng.random(len(gb)) >= drop
        hb = gb[keep] + rng.normal(0, jitter, (int(keep.sum()), 4))
        hb[:, 2:] = np.maximum(hb[:, 2:], hb[:, :2] + 1.0)
        hid = gid[keep].copy()
        for k in range(len(hid)):
            if rng.random() < swap:
                perm[hid[k]] = 1000 + int(rng.integers(0, 50))
            hid[k] = perm.get(hid[k], hid[k])
        nsp = rng.poisson(spur)
        if nsp:
            x, y = rng.uniform(0, 1800, nsp), rng.uniform(0, 1000, nsp)
            hb = np.concatenate([hb, np.column_stack([x, y, x + rng.uniform(30, 120, nsp), y + rng.uniform(60, 250, nsp)])]); hid = np.concatenate([hid, 5000 + rng.integers(0, 20, nsp)])
        _, first = np.unique(hid, return_index=True)                   # ids unique within a frame
        hb, hid = hb[np.sort(first)], hid[np.sort(first)]
        # HOTA inputs (TrackEval's preprocessing: dense ids, similarity = IoU)
        gt_ids.append(np.array([gmap.setdefault(int(v), len(gmap)) for v in gid], dtype=int))
        tr_ids.append(np.array([tmap.setdefault(int(v), len(tmap)) for v in hid], dtype=int))
        sims.append(hota.box_iou_matrix(gb, hb))
        gt_fr.append((gid, gb)); tr_fr.append((hid, hb))
        # CLEAR inputs (ltwh)
        gw = np.column_stack([gb[:, 0], gb[:, 1], gb[:, 2] - gb[:, 0], gb[:, 3] - gb[:, 1]]).reshape(-1, 4)
        hw = np.column_stack([hb[:, 0], hb[:, 1], hb[:, 2] - hb[:, 0], hb[:, 3] - hb[:, 1]]).reshape(-1, 4)
        acc_ref.update(gid, hid, mm.distances.iou_matrix(gw, hw, max_iou=0.5))
        acc.update_boxes(gid, gw, hid, hw, max_iou=0.5)
    data = {"num_tracker_dets": sum(len(v) for v in tr_ids), "num_gt_dets": sum(len(v) for v in gt_ids), "num_gt_ids": len(gmap), "num_tracker_ids": len(tmap),
            "num_timesteps": nfr, "gt_ids": gt_ids, "tracker_ids": tr_ids, "similarity_scores": sims}
    ref = H.eval_sequence(data)
    got = hota.hota_sequence(*hota.sequence_from_rows(gt_fr, tr_fr))
    dh = max(float(np.max(np.abs(np.asarray(got[k], dtype=np.float64) - np.asarray(ref[k], dtype=np.float64)))) for k in ("HOTA_TP", "HOTA_FN", "HOTA_FP", "AssA", "AssRe", "AssPr"))
    loc = float(np.max(np.abs(np.asarray(got["LocA_sum"]) / np.maximum(1e-10, np.asarray(ref["HOTA_TP"], dtype=np.float64)) - np.asarray(ref["LocA"])) * (np.asarray(ref["HOTA_TP"]) > 0)))
    dh = max(dh, loc)
    worst_h = max(worst_h, dh); ok_h += dh <= 1e-9
    if dh > 1e-9:
        print(f"HOTA DIVERGENCE trial {t}: {dh}")
    summ = mm.metrics.create().compute(acc_ref, metrics=CM, name="s")
    mine = clearmot.finalize(acc.counts())
    dc = 0.0
    for k in CM:
        a, b = float(mine[k]), float(summ[k].iloc[0])
        if np.isnan(a) and np.isnan(b):
            continue
        dc = max(dc, abs(a - b))
    worst_c = max(worst_c, dc); ok_c += dc <= 1e-9
    if dc > 1e-9:
        print(f"CLEAR DIVERGENCE trial {t}: {dc}")
print(f"hota: {ok_h}/{N} sequences equal to the vendored TrackEval HOTA (max abs difference {worst_h:.2e})")
print(f"clearmot: {ok_c}/{N} sequences equal to the vendored py-motmetrics on {len(CM)} measures (max abs difference {worst_c:.2e})")
