"""Evaluation of MOTChallenge-format results from memory or from files: HOTA (tracklab_amd.hota, TrackEval's
definition) and the CLEAR-MOT / ID measures (tracklab_amd.clearmot, py-motmetrics' definition) for a set of sequences, per sequence
and combined the way the two libraries combine them (sums of sufficient statistics) -- the same vectors a multi-GPU run
all-reduces. Mirrors what tracklab/wrappers/eval/trackeval_evaluator.py does with pip `trackeval` on the files
TrackingDataset.save_for_eval wrote: here the rows can also come straight from tables.

    python -m tracklab_amd.evaluate GT_DIR PRED_DIR [--gpu]     # <name>.txt in both; prints one JSON object; --gpu: both evaluators on the device
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

from . import clearmot, hota, mot_io


def _by_frame(rows: dict):
    """MOT rows (frame 1-based, track_id, ltwh) -> {frame: (ids, ltwh)} with frames in ascending order."""
    out = {}
    if len(rows["frame"]):
        order = np.argsort(rows["frame"], kind="stable")
        fr, ids, box = rows["frame"][order], rows["track_id"][order], np.asarray(rows["ltwh"], dtype=np.float64)[order]
        cut = np.flatnonzero(np.diff(fr)) + 1
        for f, i, b in zip(fr[np.r_[0, cut]], np.split(ids, cut), np.split(box, cut)):
            out[int(f)] = (i, b)
    return out


def evaluate_sequence(gt: dict, pred: dict, n_frames: int | None = None, max_iou: float = 0.5, device: str = "cpu") -> dict:
    """gt / pred: dicts with 'frame' (1-based), 'track_id', 'ltwh' arrays (what mot_io.load_mot returns).
    -> {'hota': packed HOTA statistics, 'clear': CLEAR-MOT / ID counts} of this sequence (summable across sequences).
    device "gpu": both evaluators run on the MI355X (tlk_hota_sequence_f64, tlk_clear_sequence_f64; at most 512 boxes per frame and side) --
    same statistics (HOTA's floating-point sums to ~1e-15, every count exactly); there is no silent fall-back to the host path."""
    if device not in ("cpu", "gpu"):
        raise ValueError(f"evaluate_sequence: device must be 'cpu' or 'gpu', not {device!r}")
    g, p = _by_frame(gt), _by_frame(pred)
    last = n_frames if n_frames is not None else max([0] + list(g) + list(p))
    empty = (np.zeros(0, np.int64), np.zeros((0, 4)))
    acc = clearmot.MOTAccumulator()
    gt_fr, pr_fr, clear_fr = [], [], []
    to_ltrb = lambda b: np.column_stack([b[:, 0], b[:, 1], b[:, 0] + b[:, 2], b[:, 1] + b[:, 3]]).reshape(-1, 4)
    for f in range(1, last + 1):
        gi, gb = g.get(f, empty)
        pi, pb = p.get(f, empty)
        if device == "cpu":
            acc.update_boxes(gi, gb, pi, pb, max_iou=max_iou)
        else:
            clear_fr.append((gi, gb, pi, pb))
        gt_fr.append((gi, to_ltrb(gb)))
        pr_fr.append((pi, to_ltrb(pb)))
    if not last:
        return {"hota": np.zeros(len(hota.ALPHAS) * 7 + 2), "clear": acc.counts()}
    if device == "gpu":
        return {"hota": hota.pack(hota.hota_sequence_gpu(gt_fr, pr_fr), frames=float(last)), "clear": clearmot.sequence_counts_gpu(clear_fr, max_iou=max_iou)}
    return {"hota": hota.pack(hota.hota_sequence(*hota.sequence_from_rows(gt_fr, pr_fr)), frames=float(last)), "clear": acc.counts()}


def combine(per_sequence: dict) -> dict:
    """{name: evaluate_sequence(...)} -> {'sequences': {name: metrics}, 'combined': metrics} (float summaries only)."""
    def metrics(h, c):
        m = dict(hota.finalize(h)["summary"])
        cm = clearmot.finalize(c)
        m.update({k.upper() if k in ("mota", "motp", "idf1", "idp", "idr") else k: cm[k] for k in
                  ("mota", "motp", "idf1", "idp", "idr", "precision", "recall", "num_switches", "num_false_positives", "num_misses",
                   "num_fragmentations", "mostly_tracked", "mostly_lost", "num_objects", "num_frames")})
        return {k: float(v) for k, v in m.items()}
    out = {"sequences": {n: metrics(r["hota"], r["clear"]) for n, r in per_sequence.items()}}
    if per_sequence:
        hsum = np.sum([r["hota"] for r in per_sequence.values()], axis=0)
        csum = clearmot.unpack(np.sum([clearmot.pack(r["clear"]) for r in per_sequence.values()], axis=0))
        out["combined"] = metrics(hsum, csum)
    return out


def evaluate_folders(gt_dir: str, pred_dir: str, device: str = "cpu") -> dict:
    res = {}
    for name in sorted(os.listdir(gt_dir)):
        if name.endswith(".txt") and os.path.exists(os.path.join(pred_dir, name)):
            res[name[:-4]] = evaluate_sequence(mot_io.load_mot(os.path.join(gt_dir, name)), mot_io.load_mot(os.path.join(pred_dir, name)), device=device)
    return combine(res)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--gpu"]
    if len(args) != 2:
        sys.exit(__doc__)
    print(json.dumps(evaluate_folders(args[0], args[1], device="gpu" if "--gpu" in sys.argv[1:] else "cpu"), indent=1))
