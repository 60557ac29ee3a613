"""The TrackLab plugin ABCs our modules subclass.

When TrackLab is installed we subclass ITS classes (``tracklab.pipeline``), so the engine's isinstance /
``level`` / ``name`` logic sees ordinary modules. When it is not importable (this build container, the GPU
box) we fall back to mirrors with the same names and contracts:

* ``level`` comes from the FIRST base class (tracklab/pipeline/module.py:33-37) -- hence the mirrors are called
  ``ImageLevelModule`` / ``DetectionLevelModule`` and the Hip* modules list them first;
* ``name`` = class name (module.py:28-31); ``input_columns`` / ``output_columns`` class attributes; ``batch_size`` +
  ``collate_fn`` class attribute (imagelevel_module.py:34-47,92-100). Nothing else of TrackLab's Module is mirrored.
"""
from __future__ import annotations

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
        """What the Hip* modules themselves rely on when TrackLab is absent (tests, the GPU box): a module's display name and its level
        (TrackLab derives the level from the FIRST base class, pipeline/module.py:33-37). The column bookkeeping of TrackLab's Module
        (get_input_columns / get_output_columns, training hooks) belongs to TrackLab's Pipeline and comes with the real base class."""
        input_columns = None
        output_columns = None

        @property
        def name(self):
            return type(self).__name__

        @property
        def level(self):
            return "image" if "Image" in type(self).__bases__[0].__name__ else "detection"

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
