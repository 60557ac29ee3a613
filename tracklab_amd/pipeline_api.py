"""The TrackLab plugin ABCs our modules subclass.

When TrackLab is installed we subclass ITS classes (``tracklab.pipeline``), so the engine's isinstance /
``level`` / ``name`` logic sees ordinary modules. When it is not importable (this build container, the GPU
box) we fall back to mirrors with the same names and contracts:

* ``level`` = first-base class name lower-cased up to the first "_" (tracklab/pipeline/module.py:33-37) --
  hence the mirrors MUST be called ``ImageLevelModule`` / ``DetectionLevelModule``;
* ``name`` = class name (module.py:28-31); ``input_columns`` / ``output_columns`` list-or-dict contract
  (module.py:51-61); ``batch_size`` + ``collate_fn`` class attribute (imagelevel_module.py:34-47,92-100).
"""
from __future__ import annotations

import re
from abc import ABCMeta, abstractmethod

try:  # pragma: no cover - exercised only where TrackLab is installed
    from tracklab.pipeline import DetectionLevelModule, ImageLevelModule, Module  # type: ignore
    HAVE_TRACKLAB = True
except Exception:  # ImportError, or one of its heavy dependencies missing
    HAVE_TRACKLAB = False

    def _default_collate(batch):
        from torch.utils.data.dataloader import default_collate
        return default_collate(batch)

    class Module(metaclass=ABCMeta):
        input_columns = None
        output_columns = None
        training_enabled = False
        forget_columns = []

        @property
        def name(self):
            return self.__class__.__name__

        @property
        def level(self):
            name = self.__class__.__bases__[0].__name__
            name = re.sub("([a-z0-9])([A-Z])", r"\1_\2", name).lower()
            return name.split("_")[0]

        def get_input_columns(self, level):
            if isinstance(self.input_columns, list):
                return self.input_columns if level == "detection" else []
            elif isinstance(self.input_columns, dict):
                return self.input_columns.get(level, [])

        def get_output_columns(self, level):
            if isinstance(self.output_columns, list):
                return self.output_columns if level == "detection" else []
            elif isinstance(self.output_columns, dict):
                return self.output_columns.get(level, [])

    class ImageLevelModule(Module):
        collate_fn = staticmethod(_default_collate)

        @abstractmethod
        def __init__(self, batch_size: int):
            self.batch_size = batch_size
            self._datapipe = None

        @abstractmethod
        def preprocess(self, image, detections, metadata):
            ...

        @abstractmethod
        def process(self, batch, detections, metadatas):
            ...

    class DetectionLevelModule(Module):
        collate_fn = staticmethod(_default_collate)

        @abstractmethod
        def __init__(self, batch_size: int):
            self.batch_size = batch_size
            self._datapipe = None

        @abstractmethod
        def preprocess(self, image, detection, metadata):
            ...

        @abstractmethod
        def process(self, batch, detections, metadatas):
            ...


def cfg_get(cfg, key, default=None):
    """Read ``key`` from an OmegaConf DictConfig, a dict or an attribute namespace."""
    if cfg is None:
        return default
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    try:
        v = getattr(cfg, key)
    except Exception:
        try:
            v = cfg[key]
        except Exception:
            return default
    return default if v is None else v


def to_numpy(x):
    """Collated batches arrive as CPU torch tensors with a leading batch dim (engine.py:152-153)."""
    if hasattr(x, "detach"):
        return x.detach().cpu().numpy()
    import numpy as np
    return np.asarray(x)
