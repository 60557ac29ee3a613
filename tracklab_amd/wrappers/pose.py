"""Pose-estimator module: the plugin surface of ``RTMPose`` (tracklab/wrappers/pose_estimator/rtmlib_api.py:14-33) with the
whole chain on the GPU: tlk_pose_crop_warp_norm (rtmlib's top-down affine crop of every box in ONE launch, straight from the
frame in HBM) -> RTMPose-shaped network (PyTorch-ROCm) -> tlk_simcc_decode. Adds ``keypoints_xyc`` (17, 3) float64 and
``keypoints_conf`` (mean keypoint score) to the detections, like the reference.
"""
from __future__ import annotations

import numpy as np
import pandas as pd

from ..pipeline_api import ImageLevelModule, cfg_get, to_numpy


class HipRTMPose(ImageLevelModule):
    input_columns = []
    output_columns = ["keypoints_xyc", "keypoints_conf"]

    def __init__(self, device, cfg=None, model=None, batch_size=1, **kwargs):
        super().__init__(batch_size=1)
        self.device = device
        self.cfg = cfg
        size = cfg_get(cfg, "model_input_size", [192, 256]) or [192, 256]          # (w, h), rtmpose_rtmlib.yaml:7
        self.in_w, self.in_h = int(size[0]), int(size[1])
        self.arch = str(cfg_get(cfg, "arch", "m"))
        self.max_dets = int(cfg_get(cfg, "max_dets", 128))
        self.checkpoint = cfg_get(cfg, "checkpoint", None)
        self._model = None

    def _ensure_model(self):
        if self._model is None:
            import torch
            from ..backbones.rtmpose import rtmpose
            self._torch = torch
            self._model = rtmpose(self.arch, device=self.device, dtype=torch.float16, channels_last=True)
            if self.checkpoint:          # state_dict, the reference's own ONNX artefact, or a BatchNorm ResNet-50 checkpoint (tracklab_amd/weights.py)
                from ..weights import load_checkpoint
                self.checkpoint_report = load_checkpoint(self._model, self.checkpoint, (torch.zeros(1, 3, self.in_h, self.in_w),))
                import logging
                logging.getLogger(__name__).info("%s: checkpoint %s -> %s", type(self).__name__, self.checkpoint, self.checkpoint_report)

    def preprocess(self, image, detections: pd.DataFrame, metadata: pd.Series):
        n = len(detections)
        if n > self.max_dets:
            raise RuntimeError("HipRTMPose: more detections in a frame than max_dets")
        boxes = np.zeros((self.max_dets, 4), dtype=np.float64)
        if n:
            ltwh = np.stack(detections.bbox_ltwh.to_list())                          # detections.bbox.ltrb(): l, t, l + w, t + h
            boxes[:n] = np.stack([ltwh[:, 0], ltwh[:, 1], ltwh[:, 0] + ltwh[:, 2], ltwh[:, 1] + ltwh[:, 3]], axis=1)
        # TrackLab hands RGB; the reference re-reads the file with cv2.imread: BGR (rtmlib_api.py:28) -> the warp kernel reads the RGB
        # frame as BGR (TLK_SWAP_RB), no flip pass on the host
        return {"image": np.ascontiguousarray(np.asarray(image)), "boxes": boxes, "count": np.int32(n)}

    def process(self, batch, detections: pd.DataFrame, metadatas: pd.DataFrame):
        if len(detections) == 0:
            return detections
        from .. import _lib
        self._ensure_model()
        torch = self._torch
        frames = batch["image"]
        frames = (frames if hasattr(frames, "detach") else torch.from_numpy(np.asarray(frames))).to(self.device).contiguous()
        boxes = torch.as_tensor(to_numpy(batch["boxes"]), dtype=torch.float64, device=self.device).contiguous()
        counts = torch.as_tensor(to_numpy(batch["count"]).reshape(-1), dtype=torch.int32, device=self.device)
        if frames.dim() == 3:
            frames, boxes = frames[None], boxes[None]
        n = len(detections)
        with torch.no_grad():
            crops, meta = _lib.pose_crop_warp_norm(frames, boxes, counts, self.in_w, self.in_h, "nhwc", torch.float16, swap_rb=True)
            sx, sy = self._model(crops[:n])
            out = _lib.simcc_decode(sx, sy, meta[:n], self.in_w, self.in_h, 2.0)
        detections = detections.copy()
        detections["keypoints_xyc"] = list(to_numpy(out["kps_xyc"]))
        detections["keypoints_conf"] = list(to_numpy(out["conf"]))
        return detections
