"""Evaluator plugin: the place of tracklab.wrappers.TrackEvalEvaluator (wrappers/eval/trackeval_evaluator.py:14-110) in a TrackLab run
-- same constructor arguments, `run(tracker_state)` -- with the three metric families of configs/eval/trackeval.yaml:11-14 computed by tracklab_amd:
  * ``results["trackeval"]``: TrackEval's RESULT LAYOUT ({sequence: {"pedestrian": {"HOTA" | "CLEAR" | "Identity": fields}}, "COMBINED_SEQ", "SUMMARIES"})
    with TrackEval's DEFINITIONS -- HOTA (tracklab_amd.hota, on the MI355X with `cfg.device: gpu`), CLEAR and Identity
    (tracklab_amd.trackeval_metrics: TrackEval's per-frame 1000x-continuity Hungarian, id switches against the last id ever matched, MT / ML at
    > 0.8 / < 0.2; host numpy) -- this is what goes to ``tracking_dataset.process_trackeval_results`` as in the reference;
  * ``results["sequences"] / results["combined"]``: flat float summaries with HOTA and the py-motmetrics CLEAR-MOT / ID measures
    (tracklab_amd.clearmot; both on the device with `cfg.device: gpu` -> tlk_hota_sequence_f64 / tlk_clear_sequence_f64).  MOTA / IDSW / MT / ML of
    this block follow py-motmetrics (what the reference's PoseTrack21 MOT evaluator uses) and can differ from TrackEval's for the same tracks.

What it takes from the tracker state is exactly what the reference's evaluator writes to its MOTChallenge files before TrackEval reads them back
(TrackingDataset.save_for_eval -> _mot_encoding, datastruct/tracking_dataset.py:161-236): per video, the rows with a track id, a box in
`cfg.bbox_column_for_eval` and a frame, as (frame + 1, track_id, ltwh); the ground truth the same way from `detections_gt` with `bbox_ltwh`.
`cfg.save_files: true` also writes those files (tracklab_amd.mot_io.save_for_eval: byte-identical to the reference's). Not restated: the class /
distractor preprocessing of TrackEval's MotChallenge2DBox dataset class (a ground truth that holds pedestrians only evaluates identically)."""
from __future__ import annotations

import logging
import os

import numpy as np
import pandas as pd

from ..pipeline_api import cfg_get

try:  # pragma: no cover - exercised only where TrackLab is installed
    from tracklab.pipeline import Evaluator as _EvaluatorBase  # type: ignore
except Exception:
    class _EvaluatorBase:                                      # tracklab/pipeline/evaluator.py:7-26: __init__(cfg) + run(tracker_state)
        pass

log = logging.getLogger(__name__)


def _mot_rows(detections: pd.DataFrame, image_metadatas: pd.DataFrame, bbox_column: str):
    """-> {video_id: dict(frame 1-based, track_id, ltwh)} with the row selection and order of mot_io.save_for_eval."""
    out = {}
    if detections is None or len(detections) == 0 or "track_id" not in detections.columns or bbox_column not in detections.columns:
        return out
    det = detections[detections["track_id"].notna() & detections[bbox_column].notna()]
    det = det[det["image_id"].isin(image_metadatas.index)]
    if len(det) == 0:
        return out
    img_pos = pd.Series(np.arange(len(image_metadatas)), index=image_metadatas.index)
    det = det.iloc[np.argsort(img_pos.loc[det["image_id"]].to_numpy(), kind="stable")]
    frames = image_metadatas.loc[det["image_id"], "frame"].to_numpy()
    vids = image_metadatas.loc[det["image_id"], "video_id"].to_numpy()
    ok = ~pd.isna(frames)
    boxes = np.stack(det[bbox_column].to_list()).reshape(-1, 4)
    if boxes.dtype == np.float32:
        # the reference's evaluator reads these numbers back from TEXT: a float32 box is printed with its shortest round-trip digits and parsed as
        # float64 (0.1f -> "0.1" -> 0.1, not 0.100000001490116...); evaluating the in-memory float32 values instead moves LocA by ~1e-8
        boxes = boxes.astype(str).astype(np.float64)
    boxes = boxes.astype(np.float64)
    tid = det["track_id"].to_numpy()
    for vid in pd.unique(vids[ok]):
        sel = ok & (vids == vid)
        out[vid] = {"frame": frames[sel].astype(np.int64) + 1, "track_id": tid[sel].astype(np.int64), "ltwh": boxes[sel]}
    return out


class HipTrackEvalEvaluator(_EvaluatorBase):
    def __init__(self, cfg, eval_set=None, show_progressbar=False, dataset_path=None, tracking_dataset=None, *args, **kwargs):
        self.cfg = cfg
        self.eval_set = eval_set
        self.show_progressbar = show_progressbar
        self.dataset_path = dataset_path
        self.tracking_dataset = tracking_dataset
        self.device = str(cfg_get(cfg, "device", "gpu"))
        if self.device not in ("gpu", "cpu"):
            raise ValueError(f"HipTrackEvalEvaluator: cfg.device must be 'gpu' or 'cpu', not {self.device!r}")
        self.results = None

    def run(self, tracker_state):
        from .. import evaluate, mot_io
        from .. import trackeval_metrics as te_metrics
        bbox_col = str(cfg_get(self.cfg, "bbox_column_for_eval", "bbox_ltwh"))
        gt_all = getattr(tracker_state, "detections_gt", None)
        if gt_all is None or len(gt_all) == 0:                  # trackeval_evaluator.py:46-49
            log.warning("Stopping evaluation because the current split (%s) has no ground truth detections.", self.eval_set)
            return None
        vm, im = tracker_state.video_metadatas, tracker_state.image_metadatas
        folder = cfg_get(self.cfg, "save_folder", None)
        if cfg_get(self.cfg, "save_files", False) and folder:
            mot_io.save_for_eval(tracker_state.detections_pred, im, vm, os.path.join(str(folder), "pred"), bbox_col, False)
            mot_io.save_for_eval(gt_all, im, vm, os.path.join(str(folder), "gt"), "bbox_ltwh", True)
        # the reference writes BOTH sides with cfg.bbox_column_for_eval (trackeval_evaluator.py:36-63); a ground truth without that column (e.g.
        # `track_bbox_kf_ltwh`, which only predictions have) falls back to bbox_ltwh with a warning instead of evaluating nothing
        gt_col = bbox_col if bbox_col in gt_all.columns else "bbox_ltwh"
        if gt_col != bbox_col:
            log.warning("ground truth has no column %r: evaluating it with bbox_ltwh", bbox_col)
        pred, gt = _mot_rows(tracker_state.detections_pred, im, bbox_col), _mot_rows(gt_all, im, gt_col)
        empty = {"frame": np.zeros(0, np.int64), "track_id": np.zeros(0, np.int64), "ltwh": np.zeros((0, 4))}
        metrics = tuple(cfg_get(self.cfg, "metrics", ("CLEAR", "HOTA", "Identity")))
        thr = 1.0 - float(cfg_get(self.cfg, "max_iou", 0.5))             # TrackEval's THRESHOLD on the IoU = 1 - motmetrics' max distance
        per_seq, te_seq = {}, {}
        for vid, video in vm.iterrows():
            n_frames = int(video["nframes"]) if "nframes" in video.index and not pd.isna(video["nframes"]) else None
            g_, p_ = gt.get(vid, empty), pred.get(vid, empty)
            r = evaluate.evaluate_sequence(g_, p_, n_frames=n_frames, max_iou=float(cfg_get(self.cfg, "max_iou", 0.5)), device=self.device)
            per_seq[str(video["name"])] = r
            # TrackEval's own CLEAR / Identity (host: one Hungarian per frame + one global one; the device evaluators above implement HOTA and the
            # py-motmetrics definitions)
            gb, pb = evaluate._by_frame(g_), evaluate._by_frame(p_)
            last = n_frames if n_frames is not None else max([0] + list(gb) + list(pb))
            none = (np.zeros(0, np.int64), np.zeros((0, 4)))
            to_ltrb = lambda b: np.column_stack([b[:, 0], b[:, 1], b[:, 0] + b[:, 2], b[:, 1] + b[:, 3]]).reshape(-1, 4)      # noqa: E731
            gfr = [(gb.get(f, none)[0], to_ltrb(gb.get(f, none)[1])) for f in range(1, last + 1)]
            pfr = [(pb.get(f, none)[0], to_ltrb(pb.get(f, none)[1])) for f in range(1, last + 1)]
            te = te_metrics.evaluate_sequence_frames(gfr, pfr, thr) if last else \
                {"CLEAR": dict({k: 0 for k in te_metrics.CLEAR_SUMMED}, MOTP_sum=0.0), "Identity": {k: 0 for k in te_metrics.ID_INT}}
            te_seq[str(video["name"])] = dict(te, hota=r["hota"])
        self.results = evaluate.combine(per_seq)
        # TrackEval's result layout with TrackEval's definitions -- what the reference hands to tracking_dataset.process_trackeval_results
        self.results["trackeval"] = te_metrics.trackeval_layout(te_seq, str(cfg_get(self.cfg, "class_name", "pedestrian")), metrics)
        summ = self.results["trackeval"].get("SUMMARIES", {})
        for cls_, fams in summ.items():
            try:
                from tabulate import tabulate
                for fam, fields in fams.items():
                    log.info("tracklab_amd evaluation (%s) %s / %s, COMBINED_SEQ\n%s", self.device, cls_, fam,
                             tabulate([list(fields.values())], headers=list(fields.keys()), tablefmt="plain"))
            except Exception:                                   # noqa: BLE001
                log.info("tracklab_amd evaluation (%s): %s", self.device, fams)
        if hasattr(self.tracking_dataset, "process_trackeval_results") and cfg_get(self.cfg, "forward_to_dataset", True):
            self.tracking_dataset.process_trackeval_results(self.results["trackeval"], dict(cfg_get(self.cfg, "dataset", {}) or {}),
                                                            dict(cfg_get(self.cfg, "eval", {}) or {}))
        return self.results
