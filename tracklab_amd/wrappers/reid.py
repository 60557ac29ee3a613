"""ReID module: crop-out + part-based embedding, image-level so that all patches of a frame are cut by ONE
tlk_roi_crop_resize_norm launch straight from the frame in HBM (the reference crops per detection on the CPU,
tracklab/wrappers/reid/kpreid_api.py:115-144, then runs the torchreid network :147-182).
Outputs the same columns: ``embeddings`` (K, D) float32 and ``visibility_scores`` (K,) bool per detection.
"""
from __future__ import annotations

import numpy as np
import pandas as pd

from ..pipeline_api import ImageLevelModule, cfg_get, to_numpy


class HipPartReID(ImageLevelModule):
    input_columns = {"detection": ["bbox_ltwh"], "image": []}
    output_columns = {"detection": ["embeddings", "visibility_scores"], "image": []}

    def __init__(self, device, cfg=None, batch_size=1, **kwargs):
        super().__init__(batch_size=1)
        self.device = device
        self.cfg = cfg
        self.height = int(cfg_get(cfg, "height", 384))       # configs/modules/reid/bpbreid.yaml:25-26
        self.width = int(cfg_get(cfg, "width", 128))
        self.parts = int(cfg_get(cfg, "parts", 6))
        self.dim = int(cfg_get(cfg, "dim", 256))
        self.max_dets = int(cfg_get(cfg, "max_dets", 128))
        self.checkpoint = cfg_get(cfg, "checkpoint", None)
        self.backbone = str(cfg_get(cfg, "backbone", "resnet50"))       # or "hrnet32" (bpbreid.yaml:53)
        self._model = None

    def _ensure_model(self):
        if self._model is None:
            import torch
            from ..backbones.reid import part_based_reid
            self._torch = torch
            self._model = part_based_reid(self.parts, self.dim, device=self.device, dtype=torch.float16, channels_last=True, arch=self.backbone)
            if self.checkpoint:          # state_dict, the reference's own ONNX artefact, or a BatchNorm ResNet-50 checkpoint (tracklab_amd/weights.py)
                from ..weights import load_checkpoint
                self.checkpoint_report = load_checkpoint(self._model, self.checkpoint, (torch.zeros(1, 3, self.height, self.width),))
                import logging
                logging.getLogger(__name__).info("%s: checkpoint %s -> %s", type(self).__name__, self.checkpoint, self.checkpoint_report)

    def preprocess(self, image, detections: pd.DataFrame, metadata: pd.Series):
        n = len(detections)
        boxes = np.zeros((self.max_dets, 4), dtype=np.float32)
        if n > self.max_dets:
            raise RuntimeError("HipPartReID: more detections in a frame than max_dets")
        if n:
            boxes[:n] = np.stack(detections.bbox_ltwh.to_list()).astype(np.float32)
        return {"image": np.ascontiguousarray(image), "boxes": boxes, "count": np.int32(n)}

    def process(self, batch, detections: pd.DataFrame, metadatas: pd.DataFrame):
        if len(detections) == 0:
            return []
        from .. import _lib
        self._ensure_model()
        torch = self._torch
        frames = batch["image"]
        frames = (frames if hasattr(frames, "detach") else torch.from_numpy(np.asarray(frames))).to(self.device).contiguous()
        boxes = torch.as_tensor(to_numpy(batch["boxes"]), dtype=torch.float32, device=self.device).contiguous()
        counts = torch.as_tensor(to_numpy(batch["count"]).reshape(-1), dtype=torch.int32, device=self.device)
        if frames.dim() == 3:
            frames, boxes = frames[None], boxes[None]
        n = int(counts.sum().item())
        with torch.no_grad():
            crops = _lib.roi_crop_resize_norm(frames, boxes, counts, self.height, self.width, "nhwc", torch.float16)
            emb, vis = self._model(crops[:n] if frames.shape[0] == 1 else crops)
        if frames.shape[0] != 1:           # batched frames: keep only the valid slots, frame-major
            keep = torch.cat([torch.arange(int(c)) + b * self.max_dets for b, c in enumerate(counts.tolist())]).to(self.device)
            emb, vis = emb[keep], vis[keep]
        emb, vis = to_numpy(emb.float()), to_numpy(vis)
        return pd.DataFrame({"embeddings": list(emb), "visibility_scores": list(vis)}, index=detections.index)
