"""MOTChallenge text files and TrackLab ``.pklz`` tracker states from in-memory tables.

* ``save_for_eval`` writes what TrackingDataset.save_for_eval writes (tracklab/datastruct/tracking_dataset.py:161-236): one
  ``<video name>.txt`` per video with ``frame+1, track_id, left, top, width, height, conf, -1|category, -1, -1``, rows without a
  track id / box / frame dropped, sorted by frame (stable), an empty file for a video without rows. Floats are printed by pandas so
  that a float32 box prints as the reference prints it.
* ``table_to_mot`` does the same straight from the columnar per-video table of ``tracklab_amd.engine`` (no merge, no DataFrame of
  Python objects).
* ``load_mot`` reads such a file back (the evaluator side: trackeval / motmetrics loaders take the same ten columns).
* ``save_tracker_state`` / ``load_tracker_state``: the zip-of-pickles layout of TrackerState.save / load
  (tracklab/datastruct/tracker_state.py:284-349): ``summary.json`` with the column lists, ``<video_id>.pkl`` detections,
  ``<video_id>_image.pkl`` image metadata -- so that detections cached by the reference can feed this tracker and vice versa.
Host-side I/O; nothing here is on the timed path.
"""
from __future__ import annotations

import json
import os
import pickle
import zipfile

import numpy as np
import pandas as pd

MOT_COLUMNS = ["frame", "track_id", "bb_left", "bb_top", "bb_width", "bb_height", "bbox_conf", "class", "y", "z"]


def _write(path, frame, track_id, ltwh, conf, cls):
    """Rows of one video -> file. ``frame`` already 1-based; a stable sort by frame like DataFrame.sort_values(kind default quicksort
    on an already grouped key is not guaranteed stable, so ties keep the reference's order only if the input is image-major, which
    both callers provide)."""
    if len(frame) == 0:
        open(path, "w").close()
        return
    order = np.argsort(np.asarray(frame), kind="stable")
    ltwh = np.asarray(ltwh)
    df = pd.DataFrame({"frame": np.asarray(frame)[order], "track_id": np.asarray(track_id)[order].astype(int),
                       "bb_left": ltwh[order, 0], "bb_top": ltwh[order, 1], "bb_width": ltwh[order, 2], "bb_height": ltwh[order, 3],
                       "bbox_conf": np.asarray(conf)[order], "class": np.asarray(cls)[order], "y": -1, "z": -1})
    df.to_csv(path, header=False, index=False)


def save_for_eval(detections: pd.DataFrame, image_metadatas: pd.DataFrame, video_metadatas: pd.DataFrame, save_folder: str,
                  bbox_column_for_eval: str = "bbox_ltwh", save_classes: bool = False):
    """Same arguments as the reference's method of the same name (without the unused is_ground_truth / save_zip)."""
    os.makedirs(save_folder, exist_ok=True)
    det = detections[detections["track_id"].notna() & detections[bbox_column_for_eval].notna()]
    det = det[det["image_id"].isin(image_metadatas.index)]
    # the reference merges image rows (in image order) with their detections: image-major order of the rows
    img_pos = pd.Series(np.arange(len(image_metadatas)), index=image_metadatas.index)
    det = det.iloc[np.argsort(img_pos.loc[det["image_id"]].to_numpy(), kind="stable")]
    frames = image_metadatas.loc[det["image_id"], "frame"].to_numpy()
    vids = image_metadatas.loc[det["image_id"], "video_id"].to_numpy()
    ok = ~pd.isna(frames)
    boxes = np.stack(det[bbox_column_for_eval].to_list()) if len(det) else np.zeros((0, 4), np.float32)
    for vid, video in video_metadatas.iterrows():
        sel = ok & (vids == vid)
        cls = det["category_id"].to_numpy()[sel] if save_classes else np.full(int(sel.sum()), -1)
        _write(os.path.join(save_folder, f"{video['name']}.txt"), frames[sel] + 1, det["track_id"].to_numpy()[sel], boxes[sel],
               det["bbox_conf"].to_numpy()[sel], cls)


def table_to_mot(table, frame_of_image, path: str, bbox_column: str = "track_bbox_ltwh", conf_column: str = "track_bbox_conf",
                 save_classes: bool = False):
    """One video straight from a ``tracklab_amd.engine.DetectionTable``: rows with a track id only. ``frame_of_image`` maps the
    table's image ids to 0-based frame numbers (an array indexed by image id, or a callable)."""
    n, c = table.n, table.cols
    keep = ~np.isnan(c["track_id"][:n])
    img = c["image_id"][:n][keep]
    frame = frame_of_image(img) if callable(frame_of_image) else np.asarray(frame_of_image)[img]
    cls = c["category_id"][:n][keep] if save_classes else np.full(int(keep.sum()), -1)
    _write(path, np.asarray(frame) + 1, c["track_id"][:n][keep], c[bbox_column][:n][keep], c[conf_column][:n][keep], cls)


def load_mot(path: str) -> dict:
    """-> dict of arrays: frame (1-based int), track_id (int), ltwh (n, 4) float64, conf, cls, plus the two trailing columns."""
    if os.path.getsize(path) == 0:
        z = np.zeros(0)
        return {"frame": z.astype(np.int64), "track_id": z.astype(np.int64), "ltwh": np.zeros((0, 4)), "conf": z, "cls": z, "y": z, "z": z}
    a = np.loadtxt(path, delimiter=",", ndmin=2)
    return {"frame": a[:, 0].astype(np.int64), "track_id": a[:, 1].astype(np.int64), "ltwh": a[:, 2:6], "conf": a[:, 6], "cls": a[:, 7],
            "y": a[:, 8], "z": a[:, 9]}


def save_tracker_state(path: str, videos: dict):
    """videos: {video_id: (detections DataFrame, image DataFrame)} -> one ``.pklz``. A video already in the archive is left alone
    (tracker_state.py:297, :323)."""
    with zipfile.ZipFile(path, mode="a", compression=zipfile.ZIP_STORED, allowZip64=True) as zf:
        names = set(zf.namelist())
        for vid, (det, img) in videos.items():
            if f"{vid}.pkl" in names:
                continue
            if "summary.json" not in names:
                with zf.open("summary.json", "w", force_zip64=True) as fp:
                    fp.write(json.dumps({"columns": {"detection": list(det.columns), "image": list(img.columns)}}, ensure_ascii=False,
                                        indent=4).encode("utf-8"))
                names.add("summary.json")
            if not det.empty:
                with zf.open(f"{vid}.pkl", "w", force_zip64=True) as fp:
                    pickle.dump(det[det.video_id == vid], fp, protocol=pickle.DEFAULT_PROTOCOL)
            if not img.empty:
                with zf.open(f"{vid}_image.pkl", "w", force_zip64=True) as fp:
                    pickle.dump(img[img.video_id == vid], fp, protocol=pickle.DEFAULT_PROTOCOL)
            names.update({f"{vid}.pkl", f"{vid}_image.pkl"})


def load_tracker_state(path: str, video_id, columns=None):
    """-> (detections, image metadata or None) of one video, restricted to ``columns["detection"]`` / ``columns["image"]`` when
    given, else to the archive's own summary (tracker_state.py:332-345). Empty detections frame when the video is absent."""
    with zipfile.ZipFile(path, mode="r") as zf:
        names = zf.namelist()
        if columns is None and "summary.json" in names:
            columns = json.loads(zf.read("summary.json").decode("utf-8"))["columns"]
        det_cols = None if columns is None else columns["detection"]
        if f"{video_id}.pkl" in names:
            with zf.open(f"{video_id}.pkl", "r") as fp:
                det = pd.read_pickle(fp)
            det = det if det_cols is None else det[det_cols]
        else:
            det = pd.DataFrame(columns=det_cols)
        img = None
        if f"{video_id}_image.pkl" in names:
            with zf.open(f"{video_id}_image.pkl", "r") as fp:
                img = pd.read_pickle(fp)
            if columns is not None:
                img = img[[c for c in columns["image"] if c in img.columns]]
    return det, img
