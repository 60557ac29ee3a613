"""HipTrackEvalEvaluator (tracklab_amd/wrappers/eval.py): the plugin that stands where tracklab.wrappers.TrackEvalEvaluator stands in a TrackLab run.
CPU here (cfg.device: cpu -- the numpy restatements); the device path it calls by default is tests/test_gpu_eval.py's. Checked: the rows it takes
from a tracker state are the rows the reference's evaluator writes to its MOTChallenge files (mot_io.save_for_eval is byte-identical to
TrackingDataset.save_for_eval, tests/test_mot_io.py), per-video and combined metrics equal evaluate_folders on those files, the early return
without ground truth, the yaml instantiates."""
import os
from types import SimpleNamespace as NS

import numpy as np
import pandas as pd
import pytest
import yaml

from tracklab_amd import evaluate, mot_io
from tracklab_amd.wrappers import HipTrackEvalEvaluator

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _state(seed=3):
    from tracklab_amd.synth import SyntheticStream
    rng = np.random.default_rng(seed)
    vm = pd.DataFrame({"name": ["seqA", "seqB", "empty"], "nframes": [12, 9, 4]}, index=pd.Index([7, 3, 11], name="id"))
    im_rows, gt_rows, pr_rows = [], [], []
    img_id = 100
    for vid, nobj, nfr in ((7, 9, 12), (3, 6, 9), (11, 0, 4)):
        stream = SyntheticStream(seed + vid, max(nobj, 1), nfr, miss_prob=0.2, churn_period=5)
        for f in range(nfr):
            fr = stream.step()
            im_rows.append({"id": img_id, "video_id": vid, "frame": f})
            if nobj:
                b = np.asarray(fr["gt_boxes"], dtype=np.float64)
                for i, gid in enumerate(fr["gt_all_ids"]):
                    gt_rows.append({"image_id": img_id, "video_id": vid, "track_id": float(gid), "bbox_conf": 1.0, "category_id": 1,
                                    "bbox_ltwh": np.array([b[i, 0], b[i, 1], b[i, 2] - b[i, 0], b[i, 3] - b[i, 1]])})
                d = fr["dets"]
                for i in range(len(d)):
                    # ids unique inside a frame (an evaluator input must be), drifting every five frames so that switches occur; some detections stay without a track
                    tid = float((i + f // 5) % len(d) + 1) if rng.random() < 0.85 else np.nan
                    box = np.array([d[i, 0], d[i, 1], d[i, 2] - d[i, 0], d[i, 3] - d[i, 1]], dtype=np.float32) + rng.normal(0, 2, 4).astype(np.float32)
                    pr_rows.append({"image_id": img_id, "video_id": vid, "track_id": tid, "bbox_conf": float(d[i, 4]), "category_id": 1, "bbox_ltwh": box,
                                    "track_bbox_kf_ltwh": box + 1})
            img_id += 1
    im = pd.DataFrame(im_rows).set_index("id")
    gt = pd.DataFrame(gt_rows); pr = pd.DataFrame(pr_rows).sample(frac=1.0, random_state=1)          # predictions arrive shuffled
    return NS(video_metadatas=vm, image_metadatas=im, detections_gt=gt, detections_pred=pr)


@pytest.mark.parametrize("col", ["bbox_ltwh", "track_bbox_kf_ltwh"])
def test_plugin_equals_the_file_based_evaluation_of_the_same_state(tmp_path, col):
    st = _state()
    ev = HipTrackEvalEvaluator(NS(device="cpu", bbox_column_for_eval=col, save_files=True, save_folder=str(tmp_path)), eval_set="val", show_progressbar=False,
                               dataset_path="unused", tracking_dataset=None)
    res = ev.run(st)
    assert set(res["sequences"]) == {"seqA", "seqB", "empty"}
    # the files the plugin wrote are the reference evaluator's files (byte-identical writer); evaluating THEM must give the same numbers
    mot_io.save_for_eval(st.detections_gt, st.image_metadatas, st.video_metadatas, str(tmp_path / "gt2"), "bbox_ltwh", True)
    for n in ("seqA", "seqB", "empty"):
        assert open(tmp_path / "gt" / f"{n}.txt").read() == open(tmp_path / "gt2" / f"{n}.txt").read()
    ref = evaluate.evaluate_folders(str(tmp_path / "gt"), str(tmp_path / "pred"), device="cpu")
    for n in ("seqA", "seqB"):
        for k, v in ref["sequences"][n].items():
            if k == "num_frames":
                continue                                  # the plugin knows nframes from the video metadata, a file only its last annotated frame
            assert res["sequences"][n][k] == pytest.approx(v, rel=1e-12, abs=1e-12), (n, k)
    assert res["combined"]["HOTA"] == pytest.approx(ref["combined"]["HOTA"], rel=1e-12)
    assert 0.0 < res["combined"]["HOTA"] < 1.0 and res["combined"]["num_switches"] > 0
    assert res["sequences"]["empty"]["num_objects"] == 0


def test_plugin_returns_early_without_ground_truth_and_rejects_a_bad_device():
    st = _state()
    st.detections_gt = pd.DataFrame(columns=["image_id"])
    assert HipTrackEvalEvaluator(NS(device="cpu"), "val", False, "x", None).run(st) is None
    with pytest.raises(ValueError):
        HipTrackEvalEvaluator(NS(device="tpu"), "val", False, "x", None)


def test_the_eval_yaml_instantiates_with_the_arguments_main_passes():
    node = yaml.safe_load(open(os.path.join(REPO, "tracklab_amd", "configs", "eval", "hip_trackeval.yaml")))
    assert node.pop("_target_") == "tracklab_amd.wrappers.HipTrackEvalEvaluator"
    ev = HipTrackEvalEvaluator(**node, tracking_dataset=NS())           # main.py: instantiate(cfg.eval, tracking_dataset=tracking_dataset)
    assert ev.device == "gpu" and ev.cfg["bbox_column_for_eval"] == "bbox_ltwh"


def test_plugin_hands_the_trackeval_layout_to_the_dataset_hook():
    """the reference always calls tracking_dataset.process_trackeval_results(results, dataset_config, eval_config) with TrackEval's
    output_res[dataset][tracker] (trackeval_evaluator.py:106-110); a hook written like wrappers/dataset/mot_like/common.py:242-258 must work"""
    seen = {}

    class Dataset:
        def process_trackeval_results(self, results, dataset_config, eval_config):
            assert "SUMMARIES" in results and "pedestrian" in results["SUMMARIES"]
            seen["flat"] = {k: float(v) if "." in v else int(v) for _, metrics in results["SUMMARIES"]["pedestrian"].items() for k, v in metrics.items()}
            seen["by_video"] = {}
            for video_name, video_data in results.items():
                if video_name != "SUMMARIES":
                    for category, metrics in video_data["pedestrian"].items():
                        for metric_name, metric_value in metrics.items():
                            if not isinstance(metric_value, np.ndarray):
                                seen["by_video"][f"{video_name}/{metric_name}"] = metric_value

    ev = HipTrackEvalEvaluator(NS(device="cpu", bbox_column_for_eval="bbox_ltwh"), eval_set="val", show_progressbar=False, dataset_path="unused",
                               tracking_dataset=Dataset())
    res = ev.run(_state())
    te = res["trackeval"]
    assert {"seqA", "seqB", "empty", "COMBINED_SEQ", "SUMMARIES"} == set(te)
    assert {"HOTA", "MOTA", "IDF1", "IDSW", "CLR_TP", "MT", "ML"} <= set(seen["flat"])
    assert seen["by_video"]["seqA/MOTA"] == te["seqA"]["pedestrian"]["CLEAR"]["MOTA"]
    comb = te["COMBINED_SEQ"]["pedestrian"]
    assert float(np.mean(comb["HOTA"]["HOTA"])) == pytest.approx(res["combined"]["HOTA"], rel=1e-12)       # same HOTA in both blocks
    assert comb["CLEAR"]["CLR_TP"] + comb["CLEAR"]["CLR_FN"] == res["combined"]["num_objects"]
