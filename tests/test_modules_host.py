"""Host logic of the TrackLab modules (DataFrame <-> arrays, index handling, plugin contract), on CPU with the
oracle injected as the backend (tests may use the oracle; the product backend is libtlk and needs a GPU)."""
import numpy as np
import pandas as pd
import pytest
from torch.utils.data.dataloader import default_collate

from tracklab_amd.pipeline_api import ImageLevelModule
from tracklab_amd.synth import SyntheticStream, ltrb_to_ltwh_rows
from tracklab_amd.wrappers import HipBPBReIDStrongSORT, HipOCSORT


class NS(dict):
    __getattr__ = dict.__getitem__


HYPER = dict(asso_func="giou", delta_t=1, det_thresh=0, inertia=0.3941737016672115,
             iou_threshold=0.22136877277096445, max_age=50, min_hits=1, use_byte=False)


def _frame_df(fr, dtype=np.float64, id0=0):
    d = fr["dets"]
    ltwh = ltrb_to_ltwh_rows(d[:, :4]).astype(dtype)
    return pd.DataFrame({"bbox_ltwh": list(ltwh), "bbox_conf": d[:, 4], "category_id": np.ones(len(d), dtype=int),
                         "image_id": fr["frame"], "video_id": 0}, index=(d[:, 6].astype(int) + id0))


def test_plugin_contract():
    m = HipOCSORT(NS(min_confidence=0.4, hyperparams=HYPER), "cuda:0", tracking_dataset=None)
    assert isinstance(m, ImageLevelModule) and m.level == "image" and m.name == "HipOCSORT" and m.batch_size == 1
    assert m.input_columns == ["bbox_ltwh", "bbox_conf", "category_id"]
    assert m.output_columns == ["track_id", "track_bbox_ltwh", "track_bbox_conf"]
    b = HipBPBReIDStrongSORT(NS(ecc=False), "cuda:0", batch_size=8)
    assert b.level == "image" and b.batch_size == 1 and len(b.output_columns) == 9
    HipBPBReIDStrongSORT(NS(ecc=True), "cuda:0")     # inert in the reference too: prepare_next_frame (its only ECC call site) is never called
    assert m.preprocess(None, pd.DataFrame(), pd.Series(dtype=float)) == {"input": []}
    assert m.process({"input": []}, pd.DataFrame(), pd.DataFrame()) == []


def test_ocsort_module_matches_oracle_wrapper(orc):
    class Backend:                                   # same surface as tracklab_amd._lib.OCSortBank
        def __init__(self):
            self.t = orc.OCSort(**HYPER)

        def update(self, dets, stream):
            return orc.ocsort_wrapper_step(self.t, dets, 0.4)

        def reset(self, stream):
            self.t = orc.OCSort(**HYPER)

    m = HipOCSORT(NS(min_confidence=0.4, hyperparams=HYPER), "cuda:0")
    m._make_backend = lambda: Backend()
    ref = orc.OCSort(**HYPER)
    m.reset()
    for dtype in (np.float64, np.float32):
        m.reset()
        ref = orc.OCSort(**HYPER)
        for fr in SyntheticStream(3, 25, 40, miss_prob=0.1):
            df = _frame_df(fr, dtype, id0=1000)
            sample = m.preprocess(None, df, pd.Series({"frame": fr["frame"]}))
            # reference semantics of OCSORT.preprocess: ltwh -> ltrb in the column's dtype, then a float64 row
            ltwh = np.stack(df.bbox_ltwh.to_list())
            exp_in = np.column_stack([ltwh[:, 0], ltwh[:, 1], ltwh[:, 0] + ltwh[:, 2], ltwh[:, 1] + ltwh[:, 3],
                                      df.bbox_conf, df.category_id, df.index]).astype(np.float64)
            np.testing.assert_array_equal(sample["input"], exp_in)
            batch = default_collate([sample])               # what the engine's DataLoader does (batch_size 1)
            out = m.process(batch, df, None)
            exp = orc.ocsort_wrapper_step(ref, exp_in, 0.4)
            if len(exp) == 0:
                assert isinstance(out, list) and not out
                continue
            assert list(out.index) == list(exp[:, 7].astype(int)) and set(out.index) <= set(df.index)
            np.testing.assert_array_equal(out.track_id.to_numpy(), exp[:, 4])
            np.testing.assert_array_equal(np.stack(out.track_bbox_ltwh.to_list()),
                                          np.column_stack([exp[:, 0], exp[:, 1], exp[:, 2] - exp[:, 0], exp[:, 3] - exp[:, 1]]))
            np.testing.assert_array_equal(out.track_bbox_conf.to_numpy(), exp[:, 6])


def test_bpbss_module_matches_oracle(orc):
    K, D = 6, 16
    cfg = NS(ecc=False, ema_alpha=0.9, mc_lambda=0.995, max_dist=0.5, motion_criterium="iou", max_iou_distance=0.8,
             max_oks_distance=0.7, max_age=300, n_init=0, nn_budget=100, min_bbox_confidence=0.0,
             only_position_for_kf_gating=False, max_kalman_prediction_without_update=7,
             matching_strategy="strong_sort_matching", gating_thres_factor=1, w_kfgd=1, w_reid=1, w_st=1)
    kw = {k: v for k, v in cfg.items() if k != "ecc"}

    class Backend:
        def __init__(self, parts, dim):
            self.t = orc.StrongSORT(parts, dim, **kw)

        def update(self, ids, ltwh, emb, vis, conf, stream, keypoints=None):
            return self.t.update(ids, ltwh, emb, vis, conf, keypoints=keypoints)

    m = HipBPBReIDStrongSORT(cfg, "cuda:0")
    m._make_backend = lambda parts, dim: Backend(parts, dim)
    ref = orc.StrongSORT(K, D, **kw)
    for fr in SyntheticStream(4, 15, 30, parts=K, dim=D, with_embeddings=True, miss_prob=0.1):
        d = fr["dets"]
        df = _frame_df(fr)
        df["embeddings"] = list(fr["embeddings"])
        df["visibility_scores"] = list(fr["visibility"])
        sample = m.preprocess(None, df, pd.Series({"frame": fr["frame"]}))
        out = m.process(default_collate([sample]), df, None)
        exp = ref.update(d[:, 6].astype(np.int64), ltrb_to_ltwh_rows(d[:, :4]), fr["embeddings"], fr["visibility"], d[:, 4])
        assert list(out.columns) == m.output_columns
        assert list(out.index) == list(exp["det_id"])
        np.testing.assert_array_equal(out.track_id.to_numpy(), exp["track_id"])
        assert list(out.state) == [{0: "t", 1: "c", 2: "d"}[s] for s in exp["state"]]
        for (mw, name, dist) in zip(out.matched_with, exp["matched_name"], exp["matched_dist"]):
            assert (mw is None) == (name == 0)
            if mw is not None:
                assert mw[0] == {1: "R", 2: "S"}[name] and mw[1] == dist
        np.testing.assert_array_equal(np.stack(out.track_bbox_kf_ltwh.to_list()), exp["kf_ltwh"])
        assert all((p is None) == (v == 0) for p, v in zip(out.track_bbox_pred_kf_ltwh, exp["pred_valid"]))


def test_strongsort_module_host_logic_with_oracle_backend(orc):
    """HipStrongSORT's DataFrame plumbing with the oracle standing in for the bank and synthetic features for the ReID forward:
    rows, index (= tracklab ids, including the stale id of a coasting track) and ltwh conversion as strong_sort_api.py:73-101."""
    from tracklab_amd._lib import SSORT_ROW
    from tracklab_amd.wrappers import HipStrongSORT
    hyper = dict(ema_alpha=0.9, max_age=10, max_dist=0.2, max_iou_dist=0.7, max_unmatched_preds=7, mc_lambda=0.995, n_init=2, nn_budget=10)
    D = 32

    class Backend:                                   # same surface as tracklab_amd._lib.SsortBank
        def __init__(self):
            self.t = orc.PlainStrongSORT(D, **hyper, img_w=1920, img_h=1080)

        def update(self, dets, feat, stream):
            keep = dets[:, 4] > 0.4                  # the bank applies min_confidence itself
            r = self.t.update(dets[keep], feat[keep])
            out = np.zeros(len(r), dtype=SSORT_ROW)
            out["ltrb"], out["track_id"], out["class_id"], out["conf"], out["det_id"] = r[:, :4], r[:, 4], r[:, 5], r[:, 6], r[:, 7]
            return out

        def reset(self, stream):
            self.t = orc.PlainStrongSORT(D, **hyper, img_w=1920, img_h=1080)

    m = HipStrongSORT(NS(min_confidence=0.4, ecc=False, hyperparams=hyper), "cuda:0", tracking_dataset=None)
    assert m.level == "image" and m.batch_size == 1 and m.output_columns == ["track_id", "track_bbox_ltwh", "track_bbox_conf"]
    m._make_backend = lambda dim, h, w: Backend()
    import torch
    m._frame_on_device = lambda image: torch.from_numpy(np.ascontiguousarray(np.asarray(image)).reshape(1080, 1920, 3))     # no GPU here: the "device" frame stays on the host
    ref = orc.PlainStrongSORT(D, **hyper, img_w=1920, img_h=1080)
    frame = np.zeros((1080, 1920, 3), np.uint8)
    seen = 0
    for fr in SyntheticStream(5, 20, 40, parts=1, dim=D, with_embeddings=True, miss_prob=0.1, low_conf_frac=0.2):
        df = _frame_df(fr, np.float64, id0=500)
        emb = fr["embeddings"][:, 0, :].astype(np.float32)
        m._features = lambda image, dets, emb=emb: emb
        sample = m.preprocess(frame, df, pd.Series({"frame": fr["frame"]}))
        batch = default_collate([sample])
        out = m.process(batch, df, pd.DataFrame({"file_path": ["unused"]}))
        d = fr["dets"].copy()
        d[:, 5] = 1.0; d[:, 6] += 500
        keep = d[:, 4] > 0.4
        exp = ref.update(d[keep], emb[keep])
        if len(exp) == 0:
            assert len(out) == 0
            continue
        seen += len(exp)
        np.testing.assert_array_equal(out.index.to_numpy(), exp[:, 7].astype(int))
        np.testing.assert_array_equal(out.track_id.to_numpy(), exp[:, 4])
        np.testing.assert_array_equal(np.stack(out.track_bbox_ltwh.to_list()),
                                      np.stack([exp[:, 0], exp[:, 1], exp[:, 2] - exp[:, 0], exp[:, 3] - exp[:, 1]], axis=1))
        np.testing.assert_array_equal(out.track_bbox_conf.to_numpy(), exp[:, 6])
    assert seen > 300


def test_strongsort_module_ecc_flow_with_a_stand_in_estimator(orc, monkeypatch):
    """`ecc: true`: the module asks the estimator (tlk_ecc_*, on the GPU) for one warp per frame, also on frames without detections, and
    hands it to the bank's camera_update before the update; no warp on the first frame of a video or where the estimator gives up --
    checked with a stand-in estimator (no GPU here) and the oracle as the bank, against the oracle driven directly in the reference's
    order (strong_sort_api.py:60-72)."""
    from tracklab_amd import _lib
    from tracklab_amd._lib import SSORT_ROW
    from tracklab_amd.wrappers import HipStrongSORT
    hyper = dict(ema_alpha=0.9, max_age=10, max_dist=0.2, max_iou_dist=0.7, max_unmatched_preds=7, mc_lambda=0.995, n_init=2, nn_budget=10)
    D = 16
    rng = np.random.default_rng(3)
    calls = []

    class Estimator:                                 # same surface as tracklab_amd._lib.EccEstimator
        made = 0

        def __init__(self, h, w, device=0):
            assert (h, w) == (1080, 1920)
            self.h, self.w, self.first = h, w, True
            Estimator.made += 1

        def apply_dev(self, frame, stream_ptr=None):             # (warp (6,) float64, status) tensors, as tlk_ecc_apply_dev leaves them in HBM
            import torch
            assert tuple(frame.shape) == (1080, 1920, 3) and frame.dtype == torch.uint8
            if self.first:
                self.first = False
                return torch.zeros(6, dtype=torch.float64), torch.tensor([0], dtype=torch.int32)
            if len(calls) == 4:
                calls.append(None)                   # cv2.error in the reference: no camera update for this frame
                return torch.zeros(6, dtype=torch.float64), torch.tensor([-1], dtype=torch.int32)
            w = np.array([[1, -0.002, rng.normal(0, 3)], [0.002, 1, rng.normal(0, 2)]], dtype=np.float32)
            calls.append(w.copy())
            return torch.from_numpy(w.astype(np.float64).reshape(6)), torch.tensor([7], dtype=torch.int32)

        def reset(self):
            self.first = True
    monkeypatch.setattr(_lib, "EccEstimator", Estimator)

    class Backend:
        def __init__(self):
            self.t = orc.PlainStrongSORT(D, **hyper, img_w=1920, img_h=1080)

        def camera_update(self, warp, stream):
            self.t.camera_update(warp)

        def update(self, dets, feat, stream):
            keep = dets[:, 4] > 0.4
            r = self.t.update(dets[keep], feat[keep])
            out = np.zeros(len(r), dtype=SSORT_ROW)
            out["ltrb"], out["track_id"], out["class_id"], out["conf"], out["det_id"] = r[:, :4], r[:, 4], r[:, 5], r[:, 6], r[:, 7]
            return out

        def reset(self, stream):
            self.t = orc.PlainStrongSORT(D, **hyper, img_w=1920, img_h=1080)

    m = HipStrongSORT(NS(min_confidence=0.4, ecc=True, hyperparams=hyper), "cuda:0", tracking_dataset=None)
    m._make_backend = lambda dim, h, w: Backend()
    import torch
    m._frame_on_device = lambda image: torch.from_numpy(np.ascontiguousarray(np.asarray(image)).reshape(1080, 1920, 3))     # no GPU here: the "device" frame stays on the host
    ref = orc.PlainStrongSORT(D, **hyper, img_w=1920, img_h=1080)
    frame = np.zeros((1080, 1920, 3), np.uint8)
    seen, k = 0, 0
    for fr in SyntheticStream(8, 12, 12, parts=1, dim=D, with_embeddings=True, miss_prob=0.1):
        df = _frame_df(fr, np.float64, id0=100)
        emb = fr["embeddings"][:, 0, :].astype(np.float32)
        m._features = lambda image, dets, emb=emb: emb
        sample = m.preprocess(frame, df, pd.Series({"frame": fr["frame"]}))
        out = m.process(default_collate([sample]), df, pd.DataFrame({"file_path": ["unused"]}))
        if fr["frame"] >= 1:                                   # strong_sort_api.py:62-65: from the second frame on
            w = calls[k]; k += 1
            if w is not None:
                ref.camera_update(w)
        keep = sample["input"][:, 4] > 0.4
        exp = ref.update(sample["input"][keep], emb[keep])
        assert len(out) == len(exp)
        if len(exp):
            np.testing.assert_array_equal(out.track_id.to_numpy(), exp[:, 4])
            np.testing.assert_array_equal(np.stack(out.track_bbox_ltwh.to_list())[:, :2], exp[:, :2])
            seen += len(exp)
    assert seen > 60 and k == len(calls) == 11 and calls[4] is None and Estimator.made == 1
    m.reset()
    assert m._ecc_est.first                                   # new video: the previous frame is forgotten


def test_botsort_module_host_logic_with_oracle_backend(orc):
    """HipBoTSORT's DataFrame plumbing with the oracle standing in for the bank: the ReID forward only sees the detections above
    track_high_thresh (bot_sort.py:293-314), rows / index / ltwh conversion as bot_sort_api.py:63-86."""
    from tracklab_amd._lib import BOTSORT_ROW
    from tracklab_amd.wrappers import HipBoTSORT
    hyper = dict(track_high_thresh=0.5, new_track_thresh=0.6, track_buffer=10, match_thresh=0.8, proximity_thresh=0.5, appearance_thresh=0.25,
                 cmc_method="none", frame_rate=30, lambda_=0.985)
    core = {k: v for k, v in hyper.items() if k != "cmc_method"}
    D = 32

    class Backend:                                   # same surface as tracklab_amd._lib.BoTSORTBank
        def __init__(self):
            self.t = orc.BoTSORT(D, **core)

        def update(self, dets, feat, stream, warp=None):
            keep = dets[:, 4] > 0.4                  # the bank applies min_confidence itself
            r = self.t.update(dets[keep], feat[keep], warp=warp)
            out = np.zeros(len(r), dtype=BOTSORT_ROW)
            out["ltrb"], out["track_id"], out["cls"], out["score"], out["det_id"] = r[:, :4], r[:, 4], r[:, 5], r[:, 6], r[:, 7]
            return out

        def reset(self, stream):
            self.t = orc.BoTSORT(D, **core)

    m = HipBoTSORT(NS(min_confidence=0.4, feature_dim=D, hyperparams=hyper), "cuda:0", tracking_dataset=None)
    assert m.level == "image" and m.batch_size == 1 and m.output_columns == ["track_id", "track_bbox_ltwh", "track_bbox_conf"]
    with pytest.raises(NotImplementedError):
        HipBoTSORT(NS(hyperparams=dict(hyper, cmc_method="orb")), "cuda:0")         # cv2 feature matching: not part of the HIP path
    HipBoTSORT(NS(hyperparams=dict(hyper, cmc_method="sparseOptFlow")), "cuda:0")   # the reference's default: accepted (estimator on the device)
    m._make_backend = lambda dim, h, w: Backend()
    ref = orc.BoTSORT(D, **core)
    frame = np.zeros((1080, 1920, 3), np.uint8)
    seen = 0
    for fr in SyntheticStream(5, 20, 40, parts=1, dim=D, with_embeddings=True, miss_prob=0.1, low_conf_frac=0.3):
        df = _frame_df(fr, np.float64, id0=500)
        emb = fr["embeddings"][:, 0, :].astype(np.float32)
        d = fr["dets"].copy()
        d[:, 5] = 1.0; d[:, 6] += 500
        hi = (d[:, 4] > 0.4) & (d[:, 4] > 0.5)

        def features(image, dets, emb=emb, d=d, hi=hi):
            np.testing.assert_allclose(dets, d[hi], rtol=1e-14)        # only the high-score rows reach the network
            return emb[hi]
        m._features = features
        sample = m.preprocess(frame, df, pd.Series({"frame": fr["frame"]}))
        out = m.process(default_collate([sample]), df, pd.DataFrame({"file_path": ["unused"]}))
        keep = d[:, 4] > 0.4
        exp = ref.update(sample["input"][keep], emb[keep])
        if len(exp) == 0:
            assert len(out) == 0
            continue
        seen += len(exp)
        np.testing.assert_array_equal(out.index.to_numpy(), exp[:, 7].astype(int))
        np.testing.assert_array_equal(out.track_id.to_numpy(), exp[:, 4])
        np.testing.assert_array_equal(np.stack(out.track_bbox_ltwh.to_list()),
                                      np.stack([exp[:, 0], exp[:, 1], exp[:, 2] - exp[:, 0], exp[:, 3] - exp[:, 1]], axis=1))
        np.testing.assert_array_equal(out.track_bbox_conf.to_numpy(), exp[:, 6])
    assert seen > 300


def test_deepocsort_module_host_logic_with_oracle_backend(orc):
    """HipDeepOCSORT's DataFrame plumbing with the oracle standing in for the bank (rows / index / ltwh as deep_oc_sort_api.py:63-82)."""
    from tracklab_amd.wrappers import HipDeepOCSORT
    hyper = dict(det_thresh=0.45, max_age=10, min_hits=1, iou_threshold=0.25, delta_t=2, asso_func="giou", inertia=0.3, w_association_emb=0.75,
                 alpha_fixed_emb=0.95, aw_param=0.5, embedding_off=False, cmc_off=True, aw_off=False, new_kf_off=False)
    D = 32

    class Backend:                                   # same surface as tracklab_amd._lib.DeepOCSortBank
        def __init__(self):
            self.t = orc.DeepOCSort(D, **hyper)

        def update(self, dets, emb, stream):
            keep = dets[:, 4] > 0.4
            return self.t.update(dets[keep], emb[keep])

        def reset(self, stream):
            self.t = orc.DeepOCSort(D, **hyper)

    m = HipDeepOCSORT(NS(min_confidence=0.4, feature_dim=D, hyperparams=hyper), "cuda:0", tracking_dataset=None)
    assert m.level == "image" and m.batch_size == 1 and m.output_columns == ["track_id", "track_bbox_ltwh", "track_bbox_conf"]
    for bad in (dict(embedding_off=True), dict(new_kf_off=True)):
        with pytest.raises(NotImplementedError):
            HipDeepOCSORT(NS(hyperparams=dict(hyper, **bad)), "cuda:0")
    HipDeepOCSORT(NS(hyperparams=dict(hyper, cmc_off=False)), "cuda:0")     # the reference's default: inert there (tag-keyed cache), inert here
    m._make_backend = lambda dim, h, w: Backend()
    ref = orc.DeepOCSort(D, **hyper)
    frame = np.zeros((1080, 1920, 3), np.uint8)
    seen = 0
    for fr in SyntheticStream(5, 20, 40, parts=1, dim=D, with_embeddings=True, miss_prob=0.1, low_conf_frac=0.3):
        df = _frame_df(fr, np.float64, id0=500)
        emb = fr["embeddings"][:, 0, :].astype(np.float32)
        conf = fr["dets"][:, 4]
        use = (conf > 0.4) & (conf > 0.45)
        m._features = lambda image, dets, emb=emb, use=use: emb[use]          # only rows above both thresholds reach the network
        sample = m.preprocess(frame, df, pd.Series({"frame": fr["frame"]}))
        out = m.process(default_collate([sample]), df, pd.DataFrame({"file_path": ["unused"]}))
        keep = conf > 0.4
        exp = ref.update(sample["input"][keep], emb[keep])
        if len(exp) == 0:
            assert len(out) == 0
            continue
        seen += len(exp)
        np.testing.assert_array_equal(out.index.to_numpy(), exp[:, 7].astype(int))
        np.testing.assert_array_equal(out.track_id.to_numpy(), exp[:, 4])
        np.testing.assert_array_equal(np.stack(out.track_bbox_ltwh.to_list()),
                                      np.stack([exp[:, 0], exp[:, 1], exp[:, 2] - exp[:, 0], exp[:, 3] - exp[:, 1]], axis=1))
        np.testing.assert_array_equal(out.track_bbox_conf.to_numpy(), exp[:, 6])
    assert seen > 300


def test_rtmpose_module_contract_and_preprocess():
    from tracklab_amd.wrappers import HipRTMPose
    m = HipRTMPose("cuda:0", cfg=NS(arch="m", model_input_size=[192, 256], max_dets=16), tracking_dataset=None)
    assert m.level == "image" and m.batch_size == 1 and m.input_columns == [] and m.output_columns == ["keypoints_xyc", "keypoints_conf"]
    img = np.arange(4 * 6 * 3, dtype=np.uint8).reshape(4, 6, 3)
    df = pd.DataFrame({"bbox_ltwh": [np.array([1.5, 2.0, 3.0, 4.0], np.float32), np.array([0, 0, 2, 2], np.float32)]}, index=[7, 9])
    s = m.preprocess(img, df, pd.Series(dtype=float))
    assert (s["image"] == img).all() and s["count"] == 2           # stays RGB: the warp kernel reads it as BGR (TLK_SWAP_RB), like cv2.imread would give
    np.testing.assert_array_equal(s["boxes"][:2], [[1.5, 2.0, 4.5, 6.0], [0, 0, 2, 2]])
    assert not s["boxes"][2:].any() and s["boxes"].shape == (16, 4)
    empty = pd.DataFrame()
    assert m.process({}, empty, pd.DataFrame()) is empty


def test_bytetrack_module_host_logic_with_oracle_backend(orc):
    from tracklab_amd._lib import BYTETRACK_ROW
    from tracklab_amd.wrappers import HipByteTrack
    hyper = dict(track_thresh=0.6, track_buffer=30, match_thresh=0.8, frame_rate=30)

    class Backend:                                   # same surface as tracklab_amd._lib.ByteTrackBank
        def __init__(self):
            self.t = orc.ByteTrack(**hyper)

        def update(self, dets, stream):
            r = self.t.update(dets[dets[:, 4] > 0.4])
            out = np.zeros(len(r), dtype=BYTETRACK_ROW)
            out["ltrb"], out["track_id"], out["cls"], out["score"], out["det_id"] = r[:, :4], r[:, 4], r[:, 5], r[:, 6], r[:, 7]
            return out

        def reset(self, stream):
            self.t = orc.ByteTrack(**hyper)

    m = HipByteTrack(NS(min_confidence=0.4, hyperparams=hyper), "cuda:0", tracking_dataset=None)
    assert m.level == "image" and m.batch_size == 1 and m.input_columns == ["bbox_ltwh", "bbox_conf", "category_id"]
    m._make_backend = lambda: Backend()
    m.reset()
    ref = orc.ByteTrack(**hyper)
    n_rows = 0
    for fr in SyntheticStream(7, 20, 40, miss_prob=0.1, low_conf_frac=0.3):
        df = _frame_df(fr, np.float64, id0=300)
        out = m.process(default_collate([m.preprocess(None, df, pd.Series({"frame": fr["frame"]}))]), df, None)
        d = fr["dets"].copy()
        d[:, 5] = 1.0; d[:, 6] += 300
        exp = ref.update(d[d[:, 4] > 0.4])
        if len(exp) == 0:
            assert len(out) == 0
            continue
        n_rows += len(exp)
        np.testing.assert_array_equal(out.index.to_numpy(), exp[:, 7].astype(int))
        np.testing.assert_array_equal(out.track_id.to_numpy(), exp[:, 4])
        np.testing.assert_array_equal(out.track_bbox_conf.to_numpy(), exp[:, 6])
        np.testing.assert_array_equal(np.stack(out.track_bbox_ltwh.to_list()),
                                      np.stack([exp[:, 0], exp[:, 1], exp[:, 2] - exp[:, 0], exp[:, 3] - exp[:, 1]], axis=1))
    assert n_rows > 300


def test_unpinned_camera_motion_warns_once_per_estimator(caplog):
    """the shipped yamls carry the reference's camera-motion defaults while the OpenCV restatement is unpinned in this image: the
    wrapper says so once per process and estimator, and stays silent where the cv2 fixture exists."""
    import logging
    import os
    from tracklab_amd.wrappers import track
    fixture = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cmc_opencv.npz")
    track._CAMERA_MOTION_WARNED.clear()
    with caplog.at_level(logging.WARNING, logger=track.__name__):
        track._warn_unpinned_camera_motion("HipBoTSORT")
        track._warn_unpinned_camera_motion("HipBoTSORT")
        track._warn_unpinned_camera_motion("HipStrongSORT")
    msgs = [r.getMessage() for r in caplog.records]
    if os.path.exists(fixture):
        assert msgs == []
    else:
        assert len(msgs) == 2 and "UNPINNED" in msgs[0] and "cmc_method: none" in msgs[0] and "ecc: false" in msgs[1]


def test_camera_motion_fixture_is_found_through_the_environment(caplog, tmp_path, monkeypatch):
    """an installed package has no tests/ directory beside it; the fixture is then named by TLK_CMC_FIXTURE (or shipped as package
    data) and the warning stays silent."""
    import logging
    from tracklab_amd.wrappers import track
    f = tmp_path / "cmc_opencv.npz"
    f.write_bytes(b"x")
    monkeypatch.setenv("TLK_CMC_FIXTURE", str(f))
    track._CAMERA_MOTION_WARNED.clear()
    with caplog.at_level(logging.WARNING, logger=track.__name__):
        track._warn_unpinned_camera_motion("HipBoTSORT")
    assert [r.getMessage() for r in caplog.records] == []
    track._CAMERA_MOTION_WARNED.clear()
