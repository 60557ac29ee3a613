"""Detector module: the plugin surface of ``RTMLibDetector`` (tracklab/wrappers/bbox_detector/rtmlib_api.py:14-46)
with the whole chain on the GPU: H2D of the frame -> tlk_letterbox_u8 -> YOLOX forward (PyTorch-ROCm) ->
tlk_yolox_decode_nms. Emits one Series per box with ``bbox_conf = 1.0`` and ``category_id = 1`` exactly like
the reference adapter (it discards the detector scores), ids from a running counter.
"""
from __future__ import annotations

import numpy as np
import pandas as pd

from ..pipeline_api import ImageLevelModule, cfg_get, to_numpy


class HipYOLOX(ImageLevelModule):
    input_columns = []
    output_columns = ["image_id", "video_id", "category_id", "bbox_ltwh", "bbox_conf"]

    def __init__(self, device, cfg=None, model=None, batch_size=1, **kwargs):
        super().__init__(batch_size=int(cfg_get(cfg, "batch_size", batch_size)))
        self.device = device
        self.cfg = cfg
        self.size = int((cfg_get(cfg, "model_input_size", [640, 640]) or [640, 640])[0])
        self.arch = str(cfg_get(cfg, "arch", "s"))
        self.max_dets = int(cfg_get(cfg, "max_dets", 128))
        self.nms_thr = float(cfg_get(cfg, "nms_thr", 0.45))          # rtmlib YOLOX defaults
        self.score_thr = float(cfg_get(cfg, "score_thr", 0.7))
        self.checkpoint = cfg_get(cfg, "checkpoint", None)
        self.id = 0
        self._model = None

    def _ensure_model(self):
        if self._model is None:
            import torch
            from ..backbones.yolox import yolox
            self._torch = torch
            self._model = yolox(self.arch, 1, device=self.device, dtype=torch.float16, channels_last=True)
            if self.checkpoint:          # state_dict, the reference's own ONNX artefact, or a BatchNorm ResNet-50 checkpoint (tracklab_amd/weights.py)
                from ..weights import load_checkpoint
                self.checkpoint_report = load_checkpoint(self._model, self.checkpoint, (torch.zeros(1, 3, self.size, self.size),))
                import logging
                logging.getLogger(__name__).info("%s: checkpoint %s -> %s", type(self).__name__, self.checkpoint, self.checkpoint_report)

    def preprocess(self, image, detections: pd.DataFrame, metadata: pd.Series):
        # TrackLab hands RGB (cv2_load_image); the reference detector re-reads the file as BGR (rtmlib_api.py:28): the frame stays RGB
        # here and the letterbox kernel reads it as BGR (TLK_SWAP_RB) -- no flip pass on the host
        return {"image": np.ascontiguousarray(np.asarray(image))}

    def process(self, batch, detections: pd.DataFrame, metadatas: pd.DataFrame):
        from .. import _lib
        self._ensure_model()
        torch = self._torch
        frames = batch["image"]
        frames = frames if hasattr(frames, "detach") else torch.from_numpy(np.asarray(frames))
        frames = frames.to(self.device, non_blocking=True).contiguous()
        B, H, W, _ = frames.shape
        with torch.no_grad():
            x, ratio = _lib.letterbox(frames, self.size, "focus_nhwc", torch.float16, swap_rb=True)
            pred = self._model(x, focused=True)
            out = _lib.yolox_decode_nms(pred, self.size, float(np.float32(ratio)), W, H, self.max_dets, self.nms_thr, self.score_thr)
        counts = to_numpy(out["counts"])
        ltwh = to_numpy(out["ltwh"])
        if (counts < 0).any():
            raise RuntimeError("HipYOLOX: more detections than max_dets / NMS candidate capacity")
        series = []
        for b in range(B):
            meta = metadatas.iloc[b]
            for i in range(int(counts[b])):
                series.append(pd.Series(dict(image_id=meta["id"] if "id" in meta else meta.name, bbox_ltwh=ltwh[b, i].copy(),
                                             bbox_conf=1.0, video_id=meta["video_id"], category_id=1), name=self.id))
                self.id += 1
        return series
