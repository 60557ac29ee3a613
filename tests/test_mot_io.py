"""CPU: tracklab_amd.mot_io against files written by the reference's TrackingDataset.save_for_eval (tests/golden/make_golden.py
gen_mot_io), the engine-table export, the reader, and the .pklz layout of TrackerState.save / load."""
import json
import os
import pickle
import zipfile

import numpy as np
import pandas as pd

from conftest import GOLDEN


def _tables(g):
    boxes = [np.asarray(b, np.float32) if f32 else np.asarray(b, np.float64) for b, f32 in zip(g["det_ltwh"], g["det_is_f32"])]
    det = pd.DataFrame({"image_id": g["det_image_id"], "video_id": g["det_video_id"], "bbox_ltwh": boxes, "bbox_conf": g["det_conf"],
                        "track_id": g["det_track_id"], "category_id": g["det_category"]}, index=pd.Index(g["det_index"], name="id"))
    imgs = pd.DataFrame({"frame": g["img_frame"], "video_id": g["img_video_id"]}, index=pd.Index(g["img_index"], name="id"))
    video = pd.DataFrame({"name": [str(n) for n in g["video_name"]]}, index=pd.Index(g["video_index"], name="id"))
    return det, imgs, video


def test_save_for_eval_writes_the_reference_bytes(tmp_path):
    from tracklab_amd import mot_io
    g = np.load(os.path.join(GOLDEN, "mot_io.npz"))
    det, imgs, video = _tables(g)
    for tag, kw in (("plain", {}), ("classes", dict(save_classes=True))):
        out = tmp_path / tag
        mot_io.save_for_eval(det, imgs, video, str(out), **kw)
        for name in video["name"]:
            assert open(out / f"{name}.txt").read() == str(g[f"{tag}_{name}"]), (tag, name)
    rows = mot_io.load_mot(str(tmp_path / "plain" / "seq-A.txt"))
    assert rows["frame"].min() == 1 and (np.diff(rows["frame"]) >= 0).all() and rows["ltwh"].shape[1] == 4 and (rows["cls"] == -1).all()
    assert len(mot_io.load_mot(str(tmp_path / "plain" / "empty.txt"))["frame"]) == 0


def test_engine_table_export_equals_dataframe_export(tmp_path):
    from tracklab_amd import mot_io
    from tracklab_amd.engine import DetectionTable
    rng = np.random.default_rng(3)
    t = DetectionTable(capacity=4)
    maxd = 8
    for step in range(10):                                     # 10 steps of 3 frames: image ids 500..529
        dcnt = rng.integers(0, 6, 3)
        ltwh = rng.uniform(0, 800, (3, maxd, 4)).astype(np.float32)
        id_base = 1000 * step
        tf, tdet = np.nonzero((np.arange(maxd)[None, :] < dcnt[:, None]) & (rng.random((3, maxd)) > 0.3))
        trk = (tf, id_base + tf * maxd + tdet, rng.integers(1, 9, len(tf)).astype(np.float64), rng.uniform(0, 800, (len(tf), 4)), np.ones(len(tf)))
        t.append_step(500 + 3 * step, 3, id_base, maxd, ltwh, dcnt, trk)
    df = t.to_dataframe(video_id=4)
    imgs = pd.DataFrame({"frame": np.arange(30), "video_id": 4}, index=pd.Index(500 + np.arange(30), name="id"))
    video = pd.DataFrame({"name": ["v"]}, index=pd.Index([4], name="id"))
    mot_io.save_for_eval(df, imgs, video, str(tmp_path), bbox_column_for_eval="track_bbox_ltwh")
    mot_io.table_to_mot(t, lambda image_ids: image_ids - 500, str(tmp_path / "direct.txt"))
    a, b = mot_io.load_mot(str(tmp_path / "v.txt")), mot_io.load_mot(str(tmp_path / "direct.txt"))
    assert len(a["frame"]) == int(df.track_id.notna().sum()) > 20
    for k in ("frame", "track_id", "ltwh"):
        np.testing.assert_array_equal(a[k], b[k])


def test_tracker_state_archive_layout_round_trip(tmp_path):
    from tracklab_amd import mot_io
    det = pd.DataFrame({"image_id": [1, 1, 2, 9], "video_id": [5, 5, 5, 6], "bbox_ltwh": [np.arange(4.0)] * 4, "bbox_conf": [0.5, 0.6, 0.7, 0.8],
                        "track_id": [1.0, 2.0, 1.0, 3.0]}, index=pd.Index([10, 11, 12, 13], name="id"))
    img = pd.DataFrame({"frame": [0, 1, 0], "video_id": [5, 5, 6]}, index=pd.Index([1, 2, 9], name="id"))
    path = str(tmp_path / "state.pklz")
    mot_io.save_tracker_state(path, {5: (det, img), 6: (det, img)})
    mot_io.save_tracker_state(path, {5: (det.iloc[:0], img)})                       # already stored: untouched
    with zipfile.ZipFile(path) as zf:                                                # read it the way TrackerState.load does
        assert sorted(zf.namelist()) == ["5.pkl", "5_image.pkl", "6.pkl", "6_image.pkl", "summary.json"]
        cols = json.loads(zf.read("summary.json"))["columns"]
        assert cols == {"detection": list(det.columns), "image": list(img.columns)}
        with zf.open("5.pkl") as fp:
            d5 = pickle.load(fp)
        with zf.open("6_image.pkl") as fp:
            i6 = pickle.load(fp)
    pd.testing.assert_frame_equal(d5, det[det.video_id == 5])
    pd.testing.assert_frame_equal(i6, img[img.video_id == 6])
    d6, im6 = mot_io.load_tracker_state(path, 6, columns={"detection": ["image_id", "track_id"], "image": ["frame"]})
    assert list(d6.columns) == ["image_id", "track_id"] and d6.index.tolist() == [13] and list(im6.columns) == ["frame"]
    d7, im7 = mot_io.load_tracker_state(path, 7)
    assert d7.empty and im7 is None and list(d7.columns) == list(det.columns)
