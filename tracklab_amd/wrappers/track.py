"""Tracker modules: same plugin surface as ``tracklab.wrappers.OCSORT`` / ``BPBReIDStrongSORT``
(tracklab/wrappers/track/oc_sort_api.py:14-76, bpbreid_strong_sort_api.py:14-118), association on the GPU.

``preprocess`` is numpy/pandas only (it may run in a DataLoader worker process, datapipe.py:37-46);
all HIP state is created lazily in the main process on the first ``process`` / ``reset``.
"""
from __future__ import annotations

import numpy as np
import pandas as pd

from ..pipeline_api import ImageLevelModule, cfg_get, to_numpy

_CAMERA_MOTION_WARNED = set()


def _warn_unpinned_camera_motion(kind: str) -> None:
    """One WARNING per process and estimator kind when the device camera-motion path is active while its OpenCV fixture has never been
    checked on this installation: the yaml defaults follow the reference (ecc: true / cmc_method: sparseOptFlow), the estimators
    restate OpenCV (cv2.findTransformECC; goodFeaturesToTrack + calcOpticalFlowPyrLK + estimateAffinePartial2D with the legacy cv::LMSolver
    refinement of OpenCV <= 4.6 -- 4.7+ routes it through cv::LevMarq) and no cv2 exists in the build image, so track ids under camera motion can
    drift from the reference's.  tests/golden/make_cmc_golden.py writes the fixture wherever cv2 is installed; tests/test_gpu_cmc.py then pins it."""
    if kind in _CAMERA_MOTION_WARNED:
        return
    _CAMERA_MOTION_WARNED.add(kind)
    import logging
    import os
    # where the fixture is looked for: TLK_CMC_FIXTURE (an installed package has no tests/ directory beside it -- ADVICE r05), package data
    # (tracklab_amd/data/cmc_opencv.npz), then the source tree's tests/golden/
    pkg = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    candidates = [os.environ.get("TLK_CMC_FIXTURE"), os.path.join(pkg, "data", "cmc_opencv.npz"),
                  os.path.join(os.path.dirname(pkg), "tests", "golden", "cmc_opencv.npz")]
    if any(c and os.path.exists(c) for c in candidates):
        return
    logging.getLogger(__name__).warning(
        "%s: camera-motion compensation runs on the device restatement of OpenCV (targets the 4.5-4.6 algorithms); parity with cv2 is UNPINNED "
        "on this installation (no cmc_opencv.npz under TLK_CMC_FIXTURE, tracklab_amd/data/ or tests/golden/: tests/golden/make_cmc_golden.py writes it where cv2 is installed). Track ids under "
        "camera motion may differ from the reference's; set %s to switch it off.", kind,
        "ecc: false" if kind == "HipStrongSORT" else "cmc_method: none")

STATE_CHARS = {0: "t", 1: "c", 2: "d"}       # TrackState, bpbreid_strong_sort/sort/track.py:16-18
MATCH_NAMES = {1: "R", 2: "S"}


class HipOCSORT(ImageLevelModule):
    input_columns = ["bbox_ltwh", "bbox_conf", "category_id"]
    output_columns = ["track_id", "track_bbox_ltwh", "track_bbox_conf"]

    def __init__(self, cfg, device, **kwargs):          # kwargs swallows tracking_dataset (main.py:36-39)
        super().__init__(batch_size=1)                  # trackers are strictly sequential (oc_sort_api.py:23)
        self.cfg = cfg
        self.device = device
        self._bank = None

    # -- backend -----------------------------------------------------------------------------------------
    def _make_backend(self):
        from .._lib import OCSortBank
        hyper = dict(cfg_get(self.cfg, "hyperparams"))
        dev = _device_index(self.device)
        return OCSortBank(**hyper, min_confidence=float(cfg_get(self.cfg, "min_confidence", 0.0)), wrapper_mode=True,
                          device=dev, max_tracks=int(cfg_get(self.cfg, "max_tracks", 256)),
                          max_dets=int(cfg_get(self.cfg, "max_dets", 128)))

    def reset(self):
        """New video: tracker state dropped, ids restart at 1 (oc_sort_api.py:28-30 re-creates the tracker)."""
        if self._bank is None:
            self._bank = self._make_backend()
        else:
            self._bank.reset(-1)

    # -- plugin API --------------------------------------------------------------------------------------
    def preprocess(self, image, detections: pd.DataFrame, metadata: pd.Series):
        if len(detections) == 0:
            return {"input": []}
        ltwh = np.stack(detections.bbox_ltwh.to_list())                 # keeps the detector's dtype (float32 for rtmlib)
        ltrb = np.stack([ltwh[:, 0], ltwh[:, 1], ltwh[:, 0] + ltwh[:, 2], ltwh[:, 1] + ltwh[:, 3]], axis=1)   # ltwh_to_ltrb
        conf = detections.bbox_conf.to_numpy(dtype=np.float64) if "bbox_conf" in detections else np.ones(len(detections))
        out = np.empty((len(detections), 7), dtype=np.float64)
        out[:, :4] = ltrb
        out[:, 4] = conf
        out[:, 5] = detections.category_id.to_numpy(dtype=np.float64)
        out[:, 6] = detections.index.to_numpy().astype(np.int64)
        return {"input": out}

    def process(self, batch, detections: pd.DataFrame, metadatas: pd.DataFrame):
        if len(detections) == 0:
            return []
        if self._bank is None:
            self.reset()
        inputs = to_numpy(batch["input"])
        inputs = inputs[0] if inputs.ndim == 3 else inputs              # (1, N, 7) after default_collate
        results = self._bank.update(np.ascontiguousarray(inputs, dtype=np.float64).reshape(-1, 7), 0)
        if not results.size:
            return []
        idxs = results[:, 7].astype(int)
        assert set(idxs).issubset(detections.index), \
            "Mismatch of indexes during the tracking. The results should match the detections."
        ltwh = np.stack([results[:, 0], results[:, 1], results[:, 2] - results[:, 0], results[:, 3] - results[:, 1]], axis=1)
        return pd.DataFrame({"track_bbox_ltwh": list(ltwh), "track_bbox_conf": list(results[:, 6]),
                             "track_id": list(results[:, 4])}, index=pd.Index(idxs, name="idxs"))


class HipBPBReIDStrongSORT(ImageLevelModule):
    input_columns = ["bbox_ltwh", "embeddings", "visibility_scores"]
    output_columns = ["track_id", "track_bbox_kf_ltwh", "track_bbox_pred_kf_ltwh", "matched_with", "costs",
                      "hits", "age", "time_since_update", "state"]

    def __init__(self, cfg, device, batch_size=None, **kwargs):
        super().__init__(batch_size=1)
        self.cfg = cfg
        self.device = device
        self._bank = None
        self._shape = None
        if cfg_get(cfg, "ecc", False):
            # The reference's only ECC call site for this tracker is BPBReIDStrongSORT.prepare_next_frame
            # (bpbreid_strong_sort_api.py:62-70), which nothing calls: StrongSORT.update predicts itself (strong_sort.py:86).
            # `ecc: true` therefore changes nothing there, and nothing here.
            import logging
            logging.getLogger(__name__).info("ecc: true has no effect for BPBReID-StrongSORT (the reference never calls prepare_next_frame)")

    def _make_backend(self, parts, dim):
        from .._lib import BpbssBank
        c = self.cfg
        names = ("ema_alpha", "mc_lambda", "max_dist", "motion_criterium", "max_iou_distance", "max_oks_distance", "max_age",
                 "n_init", "nn_budget", "min_bbox_confidence", "only_position_for_kf_gating",
                 "max_kalman_prediction_without_update", "matching_strategy", "gating_thres_factor", "w_kfgd", "w_reid", "w_st")
        kw = {n: cfg_get(c, n) for n in names if cfg_get(c, n) is not None}
        return BpbssBank(parts, dim, **kw, wrapper_mode=True, device=_device_index(self.device),
                         max_tracks=int(cfg_get(c, "max_tracks", 4096)), max_dets=int(cfg_get(c, "max_dets", 128)))

    def reset(self):
        if self._bank is not None:
            self._bank.reset(-1)

    def preprocess(self, image, detections: pd.DataFrame, metadata: pd.Series):
        if len(detections) == 0:
            return {"id": [], "bbox_ltwh": [], "reid_features": [], "visibility_scores": [], "scores": [], "classes": [], "frame": []}
        score = detections.bbox_conf if "bbox_conf" in detections else detections.keypoints_conf
        return {"id": detections.index.to_numpy(),
                "bbox_ltwh": np.stack(detections.bbox_ltwh.to_list()),
                "reid_features": np.stack(detections.embeddings.to_list()),
                "visibility_scores": np.stack(detections.visibility_scores.to_list()),
                "scores": np.asarray(score, dtype=np.float64),
                "classes": np.zeros(len(detections.index)),
                "frame": np.ones(len(detections.index)) * metadata.frame,
                **({"keypoints": np.stack(detections.keypoints_xyc.to_list())} if "keypoints_xyc" in detections else {})}

    def process(self, batch, detections: pd.DataFrame, metadatas: pd.DataFrame):
        if len(detections) == 0:
            return []
        ids = _strip(to_numpy(batch["id"]), 1)                      # default_collate adds a leading batch dim of 1
        ltwh = _strip(to_numpy(batch["bbox_ltwh"]), 2)
        emb = _strip(to_numpy(batch["reid_features"]), 3)
        vis = _strip(to_numpy(batch["visibility_scores"]), 2)
        conf = _strip(to_numpy(batch["scores"]), 1)
        if self._bank is None:
            self._bank = self._make_backend(emb.shape[1], emb.shape[2])
        kps = _strip(to_numpy(batch["keypoints"]), 3).astype(np.float64) if "keypoints" in batch else None
        rows = self._bank.update(ids.astype(np.int64), ltwh.astype(np.float64), emb.astype(np.float32),
                                 vis.astype(bool).astype(np.uint8), conf.astype(np.float64), 0, keypoints=kps)
        assert set(rows["det_id"]).issubset(detections.index), \
            "Mismatch of indexes during the tracking. The results should match the detections."
        out = pd.DataFrame({
            "track_id": rows["track_id"].astype(int),
            "track_bbox_kf_ltwh": list(rows["kf_ltwh"]),
            "track_bbox_pred_kf_ltwh": [p if v else None for p, v in zip(rows["pred_ltwh"], rows["pred_valid"])],
            "matched_with": [(MATCH_NAMES[m], d) if m else None for m, d in zip(rows["matched_name"], rows["matched_dist"])],
            "costs": [{} for _ in range(len(rows))],      # debug-only dictionaries of the reference are not produced
            "hits": rows["hits"].astype(int), "age": rows["age"].astype(int),
            "time_since_update": rows["tsu"].astype(int), "state": [STATE_CHARS[s] for s in rows["state"]],
        }, index=np.asarray(rows["det_id"]), columns=self.output_columns)
        return out


class HipByteTrack(ImageLevelModule):
    """ByteTrack (tracklab/wrappers/track/byte_track_api.py:14-84) with the tracker step on the GPU (tlk_bytetrack_update)."""
    input_columns = ["bbox_ltwh", "bbox_conf", "category_id"]
    output_columns = ["track_id", "track_bbox_ltwh", "track_bbox_conf"]

    def __init__(self, cfg, device, **kwargs):
        super().__init__(batch_size=1)
        self.cfg = cfg
        self.device = device
        self._bank = None

    def _make_backend(self):
        from .._lib import ByteTrackBank
        hyper = dict(cfg_get(self.cfg, "hyperparams"))
        return ByteTrackBank(**hyper, min_confidence=float(cfg_get(self.cfg, "min_confidence", 0.0)), wrapper_mode=True,
                             device=_device_index(self.device), max_tracks=int(cfg_get(self.cfg, "max_tracks", 256)),
                             max_dets=int(cfg_get(self.cfg, "max_dets", 128)))

    def reset(self):
        """New video (byte_track_api.py:29-31 rebuilds the tracker). The reference's id counter is class-level (BaseTrack._count,
        byte_track/basetrack.py:13,35-37) and keeps counting across videos: so does this one (``reset_ids_per_video: true`` in the yaml
        restarts at 1 instead)."""
        if self._bank is None:
            self._bank = self._make_backend()
        else:
            self._bank.reset(-1, keep_ids=not bool(cfg_get(self.cfg, "reset_ids_per_video", False)))

    preprocess = HipOCSORT.preprocess          # same (N, 7) [*ltrb, conf, cls, tracklab_id] rows (byte_track_api.py:33-50)

    def process(self, batch, detections: pd.DataFrame, metadatas: pd.DataFrame):
        if len(detections) == 0:
            return []
        if self._bank is None:
            self.reset()
        inputs = to_numpy(batch["input"])
        inputs = inputs[0] if inputs.ndim == 3 else inputs
        rows = self._bank.update(np.ascontiguousarray(inputs, dtype=np.float64).reshape(-1, 7), 0)
        if not len(rows):
            return []
        idxs = rows["det_id"].astype(int)
        assert set(idxs).issubset(detections.index), \
            "Mismatch of indexes during the tracking. The results should match the detections."
        ltrb = rows["ltrb"]
        ltwh = np.stack([ltrb[:, 0], ltrb[:, 1], ltrb[:, 2] - ltrb[:, 0], ltrb[:, 3] - ltrb[:, 1]], axis=1)
        return pd.DataFrame({"track_bbox_ltwh": list(ltwh), "track_bbox_conf": list(rows["score"]),
                             "track_id": list(rows["track_id"].astype(float))}, index=pd.Index(idxs, name="idxs"))


class _ReidTrackerBase:
    """What the two trackers that own a ReID network share (strong_sort_api.py / bot_sort_api.py are the same wrapper around a
    different tracker): DataFrame -> (n, 7) rows, GPU crop-out + backbone forward, bank update, rows -> DataFrame. A mixin, not a
    base module: TrackLab derives a module's level from its FIRST base class (pipeline/module.py:33-37)."""
    _conf_field = "conf"

    def reset(self):
        """New video (strong_sort_api.py:31-40 rebuilds model + tracker): state dropped, ids restart at 1."""
        if self._bank is not None:
            self._bank.reset(-1)

    def preprocess(self, image, detections: pd.DataFrame, metadata: pd.Series):
        if len(detections) == 0:
            return {"input": []}
        ltwh = np.stack(detections.bbox_ltwh.to_list())
        out = np.empty((len(detections), 7), dtype=np.float64)             # [*ltrb, conf, cls, tracklab_id] (strong_sort_api.py:47-55)
        out[:, 0], out[:, 1], out[:, 2], out[:, 3] = ltwh[:, 0], ltwh[:, 1], ltwh[:, 0] + ltwh[:, 2], ltwh[:, 1] + ltwh[:, 3]
        out[:, 4] = detections.bbox_conf.to_numpy(dtype=np.float64)
        out[:, 5] = detections.category_id.to_numpy(dtype=np.float64)
        out[:, 6] = detections.index.to_numpy().astype(np.int64)
        return {"input": out, "image": np.ascontiguousarray(image)}         # the frame travels with the batch instead of a second disk read

    def _frame_on_device(self, image):
        """The frame of the current `process` call as a (H, W, 3) uint8 device tensor -- uploaded ONCE and shared by the ReID crop-out and the
        camera-motion estimator (the estimator used to copy the host frame a second time, synchronously)."""
        import torch
        if hasattr(image, "detach") and image.is_cuda:
            return image[0] if image.dim() == 4 else image
        arr = np.asarray(to_numpy(image) if hasattr(image, "detach") else image)
        if arr.ndim == 4:
            arr = arr[0]
        key = (arr.__array_interface__["data"][0], arr.shape)
        cached = getattr(self, "_dev_frame", None)
        if cached is None or cached[0] != key:
            self._dev_frame = (key, torch.from_numpy(np.ascontiguousarray(arr)).to(self.device))
        return self._dev_frame[1]

    def _features(self, image, dets):
        """StrongSORT._get_features (strong_sort.py:135-145) for all rows of `dets` (n, 7) on the GPU -> (n, D) float32 numpy."""
        import torch
        from .. import _lib
        if self._model is None:
            from ..backbones.reid import part_based_reid
            self._dim = int(cfg_get(self.cfg, "feature_dim", 512))
            self._model = part_based_reid(1, self._dim, device=self.device, dtype=torch.float16, channels_last=True)
            ckpt = cfg_get(self.cfg, "model_weights", None)
            if ckpt:
                import os
                if not os.path.exists(str(ckpt)):       # never fall back to random weights silently (the reference's OSNet / BPBReID
                    raise FileNotFoundError(             # checkpoints are not loadable either: INTEGRATION.md "checkpoints")
                        f"model_weights {ckpt!r} does not exist; set model_weights: null to run with random-init weights (throughput only)")
                from ..weights import load_checkpoint
                self.checkpoint_report = load_checkpoint(self._model, ckpt, (torch.zeros(1, 3, 256, 128),))
                import logging
                logging.getLogger(__name__).info("%s: checkpoint %s -> %s", type(self).__name__, ckpt, self.checkpoint_report)
        frames = self._frame_on_device(image)[None]
        n = len(dets)
        boxes = torch.from_numpy(np.ascontiguousarray(dets, dtype=np.float64)[None]).to(self.device)
        counts = torch.tensor([n], dtype=torch.int32, device=self.device)
        with torch.no_grad():
            crops = _lib.roi_crop_pil_resize_norm(frames.contiguous(), boxes, counts, 256, 128, "nhwc", torch.float16)
            emb, _ = self._model(crops)
        return to_numpy(emb[:, 0, :].float())

    def _frame_features(self, image, inputs):
        return self._features(image, inputs)

    def _camera_step(self, image, metadatas):
        """Camera-motion compensation ahead of the tracker step; nothing unless a subclass estimates a warp."""

    def _update(self, inputs, feats, image):
        return self._bank.update(inputs, feats, 0)

    def process(self, batch, detections: pd.DataFrame, metadatas: pd.DataFrame):
        # the uploaded frame is shared by the stages of THIS call only: a host buffer's address says nothing about its content in the next call
        # (an allocator hands the address of the previous frame's freed buffer to the next frame -- a stale hit fed the estimator the old frame)
        self._dev_frame = None
        if len(detections) == 0:
            self._camera_step(None, metadatas)               # the reference compensates before its empty-frame return too
            return []
        inputs = to_numpy(batch["input"])
        inputs = np.ascontiguousarray(inputs[0] if inputs.ndim == 3 else inputs, dtype=np.float64).reshape(-1, 7)
        image = batch.get("image") if hasattr(batch, "get") else None
        if image is None:                                                    # reference: cv2_load_image(metadatas['file_path'])
            from PIL import Image
            image = np.asarray(Image.open(metadatas["file_path"].values[0]).convert("RGB"))
        image = to_numpy(image) if not hasattr(image, "detach") else image
        if getattr(image, "ndim", 3) == 4:
            image = image[0]
        h, w = int(image.shape[0]), int(image.shape[1])
        self._camera_step(image, metadatas)
        feats = self._frame_features(image, inputs)
        if self._bank is None or self._img_hw != (h, w):
            self._bank = self._make_backend(feats.shape[1], h, w)
            self._img_hw = (h, w)
        rows = self._update(inputs, feats, image)
        if not len(rows):
            return []
        ltrb = rows["ltrb"]
        ltwh = np.stack([ltrb[:, 0], ltrb[:, 1], ltrb[:, 2] - ltrb[:, 0], ltrb[:, 3] - ltrb[:, 1]], axis=1)      # ltrb_to_ltwh
        # no subset assert: like the reference (strong_sort_api.py:85-89) a coasting track reports the id of its previous detection
        return pd.DataFrame({"track_bbox_ltwh": list(ltwh), "track_bbox_conf": list(rows[self._conf_field]),
                             "track_id": list(rows["track_id"].astype(float))}, index=pd.Index(rows["det_id"].astype(int), name="idxs"))


class HipStrongSORT(ImageLevelModule, _ReidTrackerBase):
    """Plain StrongSORT (tracklab/wrappers/track/strong_sort_api.py:17-105): the tracker owns its ReID network. Crops are cut
    and resized on the GPU with the reference's own arithmetic (int-truncated box, Pillow bilinear, ImageNet normalisation:
    tlk_roi_crop_pil_resize_norm), one backbone forward gives the 512-d features, tlk_ssort_update does the rest."""
    input_columns = ["bbox_ltwh", "bbox_conf", "category_id"]
    output_columns = ["track_id", "track_bbox_ltwh", "track_bbox_conf"]
    preprocess, process = _ReidTrackerBase.preprocess, _ReidTrackerBase.process    # ahead of the abstract ones in the MRO

    def __init__(self, cfg, device, **kwargs):
        super().__init__(batch_size=1)
        self.cfg = cfg
        self.device = device
        self._bank = None
        self._model = None
        self._img_hw = None
        self._ecc = bool(cfg_get(cfg, "ecc", True))        # the reference's default (configs/modules/track/strong_sort.yaml:13)
        self._ecc_est = None        # _lib.EccEstimator, created with the first frame's size

    def reset(self):
        if self._ecc_est is not None:
            self._ecc_est.reset()
        if self._bank is not None:
            self._bank.reset(-1)

    def _camera_step(self, image, metadatas):
        """strong_sort_api.py:61-65: tracker.camera_update(previous frame, current frame) ahead of the update. Track.ECC (sort/track.py:129-211:
        grey, 0.1-scaled frames, Euclidean model, translation rescaled) runs on the GPU (tlk_ecc_*: cv2.findTransformECC restated, parity
        unpinned -- DESIGN.md); one estimate per frame (the reference recomputes the same one for every track)."""
        if not self._ecc:
            return
        _warn_unpinned_camera_motion("HipStrongSORT")
        if image is None:                                                    # strong_sort_api.py:61: the frame is read before the empty check
            from PIL import Image
            image = np.asarray(Image.open(metadatas["file_path"].values[0]).convert("RGB"))
        dev = self._frame_on_device(image)
        if self._ecc_est is None or (self._ecc_est.h, self._ecc_est.w) != tuple(dev.shape[:2]):
            from .. import _lib
            if self._ecc_est is not None:
                import logging
                logging.getLogger(__name__).warning("HipStrongSORT: frame size changed inside a video (%s -> %s): the ECC estimator restarts without a "
                                                    "previous frame", (self._ecc_est.h, self._ecc_est.w), tuple(dev.shape[:2]))
            self._ecc_est = _lib.EccEstimator(int(dev.shape[0]), int(dev.shape[1]), device=_device_index(self.device))
        warp_dev, status_dev = self._ecc_est.apply_dev(dev.contiguous())     # on the current stream, from the frame that is already in HBM
        status = int(status_dev.item())                                      # one 4-byte read-back: >= 1 iterations run, -1 where cv2 raises, 0 first frame
        warp = warp_dev.cpu().numpy().reshape(2, 3).astype(np.float32) if status >= 1 else None
        if warp is not None and self._bank is not None:
            self._bank.camera_update(warp, 0)

    def _make_backend(self, dim, img_h, img_w):
        from .._lib import SsortBank
        hyper = dict(cfg_get(self.cfg, "hyperparams"))
        return SsortBank(dim, **hyper, min_confidence=float(cfg_get(self.cfg, "min_confidence", 0.0)), wrapper_mode=True,
                         img_w=int(img_w), img_h=int(img_h), device=_device_index(self.device),
                         max_tracks=int(cfg_get(self.cfg, "max_tracks", 256)), max_dets=int(cfg_get(self.cfg, "max_dets", 128)),
                         gallery_rows=int(cfg_get(self.cfg, "gallery_rows", 4096)))     # rows per track when hyperparams.nn_budget is null


class HipBoTSORT(ImageLevelModule, _ReidTrackerBase):
    """BoT-SORT (tracklab/wrappers/track/bot_sort_api.py:16-88). Same crop-out as plain StrongSORT (both use
    ReIDDetectMultiBackend's ToPILImage / Resize / Normalize on the int-truncated box: bot_sort.py:487-505,
    deep_oc_sort/reid_multibackend.py:44-53), features only for the detections above track_high_thresh (bot_sort.py:293-314),
    tlk_botsort_update for the rest. cmc_method "sparseOptFlow" (the reference's default) runs on the device: tlk_cmc_* estimates the
    frame's warp (grey + 2:1 resize, Shi-Tomasi corners, pyramidal Lucas-Kanade, RANSAC similarity; OpenCV restated, parity
    unpinned), tlk_botsort_update_gmc applies it between predict and association. orb / sift / ecc / file stay cv2-only."""
    input_columns = ["bbox_ltwh", "bbox_conf", "category_id"]
    output_columns = ["track_id", "track_bbox_ltwh", "track_bbox_conf"]
    _conf_field = "score"
    preprocess, process = _ReidTrackerBase.preprocess, _ReidTrackerBase.process    # ahead of the abstract ones in the MRO

    def reset(self):
        """New video: state dropped; the id counter keeps counting like the reference's class-level BaseTrack._count
        (bot_sort/basetrack.py) unless ``reset_ids_per_video: true``. The camera-motion estimator forgets its previous frame."""
        if self._bank is not None:
            self._bank.reset(-1, keep_ids=not bool(cfg_get(self.cfg, "reset_ids_per_video", False)))
        if self._cmc is not None:
            self._cmc.reset()

    def __init__(self, cfg, device, **kwargs):
        super().__init__(batch_size=1)
        self.cfg = cfg
        self.device = device
        self._bank = None
        self._model = None
        self._img_hw = None
        self._cmc = None
        self._cmc_method = dict(cfg_get(cfg, "hyperparams")).get("cmc_method", "sparseOptFlow")
        if self._cmc_method in ("None",):
            self._cmc_method = "none"
        if self._cmc_method not in ("none", None, "sparseOptFlow"):
            raise NotImplementedError(f"cmc_method {self._cmc_method!r} (gmc.py: cv2 ORB / SIFT / ECC estimators, GMC files) is not part of the HIP path; "
                                      "use sparseOptFlow (the reference's default, on the device) or none.  (In the reference itself 'file' / 'files' cannot "
                                      "be constructed through BoTSORT -- bot_sort.py:273 passes verbose=[None, False] and gmc.py:38-45 then tests "
                                      "'-FRCNN' in None -- and 'ecc' aligns every frame with the FIRST one: gmc.py:82-110 never refreshes prevFrame.)")

    def _update(self, inputs, feats, image):
        warp = None
        if self._cmc_method == "sparseOptFlow":              # bot_sort.py:341: warp = self.gmc.apply(img, dets)
            from .._lib import CmcEstimator
            _warn_unpinned_camera_motion("HipBoTSORT")
            dev = self._frame_on_device(image)
            if self._cmc is None or (self._cmc.h, self._cmc.w) != tuple(dev.shape[:2]):
                if self._cmc is not None:
                    import logging
                    logging.getLogger(__name__).warning("HipBoTSORT: frame size changed inside a video (%s -> %s): the camera-motion estimator restarts "
                                                        "without a previous frame", (self._cmc.h, self._cmc.w), tuple(dev.shape[:2]))
                self._cmc = CmcEstimator(int(dev.shape[0]), int(dev.shape[1]), downscale=2, device=_device_index(self.device))
            warp = self._cmc.apply_dev(dev.contiguous()).cpu().numpy().reshape(2, 3)      # estimated from the frame already in HBM; 48 bytes back
        return self._bank.update(inputs, feats, 0, warp=warp)

    def _make_backend(self, dim, img_h, img_w):
        from .._lib import BoTSORTBank
        hyper = dict(cfg_get(self.cfg, "hyperparams"))
        return BoTSORTBank(dim, **hyper, min_confidence=float(cfg_get(self.cfg, "min_confidence", 0.0)), wrapper_mode=True,
                           device=_device_index(self.device), max_tracks=int(cfg_get(self.cfg, "max_tracks", 256)),
                           max_dets=int(cfg_get(self.cfg, "max_dets", 128)))

    def _frame_features(self, image, inputs):
        hyper = dict(cfg_get(self.cfg, "hyperparams"))
        hi = (inputs[:, 4] > float(cfg_get(self.cfg, "min_confidence", 0.0))) & (inputs[:, 4] > float(hyper.get("track_high_thresh", 0.45)))
        dim = int(cfg_get(self.cfg, "feature_dim", 512))
        feats = np.zeros((len(inputs), dim), dtype=np.float32)
        if hi.any():
            feats[hi] = self._features(image, inputs[hi])
        return feats


class HipDeepOCSORT(ImageLevelModule, _ReidTrackerBase):
    """Deep-OC-SORT (tracklab/wrappers/track/deep_oc_sort_api.py:16-88). The crop box is the detection's corners truncated to int
    with no clamping (ocsort.py:543-547: a negative corner wraps around like numpy indexing, which the HIP crop does not imitate:
    boxes are clamped to the frame), resized / normalised like the other two ReID trackers; the embeddings go to
    tlk_deepocsort_update as delivered (the reference does not normalise them). cmc_off: false is accepted and inert, as it is in the
    reference's wiring (see __init__)."""
    input_columns = ["bbox_ltwh", "bbox_conf", "category_id"]
    output_columns = ["track_id", "track_bbox_ltwh", "track_bbox_conf"]
    preprocess, reset = _ReidTrackerBase.preprocess, _ReidTrackerBase.reset

    def __init__(self, cfg, device, **kwargs):
        super().__init__(batch_size=1)
        self.cfg = cfg
        self.device = device
        self._bank = None
        self._model = None
        self._img_hw = None
        hyper = dict(cfg_get(cfg, "hyperparams"))
        if not hyper.get("cmc_off", False):
            # OCSort.update is always called with its default tag 'blub' (deep_oc_sort_api.py:64, ocsort.py:392) and CMCComputer caches its
            # result by tag (cmc.py:68-72): the first frame's estimate -- the identity, there is no previous frame yet (cmc.py:144-147) --
            # is returned for every later frame. apply_affine_correction with the identity changes nothing, so cmc_off: false is inert
            # in TrackLab's wiring (short of a stale ./cache/affine_ocsort.pkl), and it is inert here.
            import logging
            logging.getLogger(__name__).info("cmc_off: false is inert for Deep-OC-SORT as TrackLab wires it (CMCComputer caches the first frame's "
                                             "identity under the constant tag); running without camera-motion compensation")
        if hyper.get("embedding_off", False) or hyper.get("new_kf_off", False):
            raise NotImplementedError("embedding_off / new_kf_off are not part of the HIP path (embedding_off fails in the reference itself)")

    def _make_backend(self, dim, img_h, img_w):
        from .._lib import DeepOCSortBank
        hyper = dict(cfg_get(self.cfg, "hyperparams"), cmc_off=True)       # see __init__: false is the identity in the reference's wiring
        return DeepOCSortBank(dim, **hyper, min_confidence=float(cfg_get(self.cfg, "min_confidence", 0.0)), wrapper_mode=True,
                              device=_device_index(self.device), max_tracks=int(cfg_get(self.cfg, "max_tracks", 256)),
                              max_dets=int(cfg_get(self.cfg, "max_dets", 128)))

    def _frame_features(self, image, inputs):
        hyper = dict(cfg_get(self.cfg, "hyperparams"))
        use = (inputs[:, 4] > float(cfg_get(self.cfg, "min_confidence", 0.0))) & (inputs[:, 4] > float(hyper.get("det_thresh", 0.0)))
        feats = np.zeros((len(inputs), int(cfg_get(self.cfg, "feature_dim", 512))), dtype=np.float32)
        if use.any():
            feats[use] = self._features(image, inputs[use])
        return feats

    def process(self, batch, detections: pd.DataFrame, metadatas: pd.DataFrame):
        self._dev_frame = None                               # (the uploaded frame is shared within one call only: _ReidTrackerBase.process)
        if len(detections) == 0:
            return []
        inputs = to_numpy(batch["input"])
        inputs = np.ascontiguousarray(inputs[0] if inputs.ndim == 3 else inputs, dtype=np.float64).reshape(-1, 7)
        image = batch.get("image") if hasattr(batch, "get") else None
        if image is None:
            from PIL import Image
            image = np.asarray(Image.open(metadatas["file_path"].values[0]).convert("RGB"))
        image = to_numpy(image) if not hasattr(image, "detach") else image
        if getattr(image, "ndim", 3) == 4:
            image = image[0]
        feats = self._frame_features(image, inputs)
        if self._bank is None:
            self._bank = self._make_backend(feats.shape[1], int(image.shape[0]), int(image.shape[1]))
        rows = self._bank.update(inputs, feats, 0)
        if not len(rows):
            return []
        ltwh = np.stack([rows[:, 0], rows[:, 1], rows[:, 2] - rows[:, 0], rows[:, 3] - rows[:, 1]], axis=1)      # ltrb_to_ltwh
        out = pd.DataFrame({"track_bbox_ltwh": list(ltwh), "track_bbox_conf": list(rows[:, 6]), "track_id": list(rows[:, 4])},
                           index=pd.Index(rows[:, 7].astype(int), name="idxs"))
        return out[~out.index.duplicated(keep="first")]                              # deep_oc_sort_api.py:82


def _strip(a, ndim):
    return a[0] if a.ndim == ndim + 1 else a


def _device_index(device) -> int:
    s = str(device)
    if ":" in s:
        return int(s.split(":")[1])
    return 0
