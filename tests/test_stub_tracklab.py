"""The `HAVE_TRACKLAB` branch and the yaml `_target_`s, executed against a STUB TrackLab (the real one -- lightning, hydra, mmdet, ... -- is not
installable here). The stub holds only what our code touches: `tracklab.pipeline.{Module, ImageLevelModule, DetectionLevelModule}` with
the attributes pipeline/module.py:28-61 derives (`name`, `level`, column getters) and `tracklab.engine.TrackingEngine` with the
constructor of engine/engine.py:76-103. Hydra is absent too: `_instantiate` below is its `_target_` contract (resolve the dotted path,
remaining keys = keyword arguments, nested `_target_`s first, call-site kwargs override)."""
import importlib
import inspect
import os
import subprocess
import sys
import textwrap

import pytest
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = os.path.join(REPO, "tracklab_amd", "configs")

STUB = {
    "tracklab/__init__.py": "",
    "tracklab/pipeline/__init__.py": """
        import re
        class Module:
            input_columns = None; output_columns = None; training_enabled = False; forget_columns = []
            STUB = True
            @property
            def name(self): return self.__class__.__name__
            @property
            def level(self):
                n = re.sub("([a-z0-9])([A-Z])", r"\\1_\\2", self.__class__.__bases__[0].__name__).lower()
                return n.split("_")[0]
            def get_input_columns(self, level):
                c = self.input_columns
                return (c if level == "detection" else []) if isinstance(c, list) else c.get(level, [])
            def get_output_columns(self, level):
                c = self.output_columns
                return (c if level == "detection" else []) if isinstance(c, list) else c.get(level, [])
        class ImageLevelModule(Module):
            def __init__(self, batch_size): self.batch_size = batch_size; self._datapipe = None
        class DetectionLevelModule(Module):
            def __init__(self, batch_size): self.batch_size = batch_size; self._datapipe = None
        """,
    "tracklab/engine/__init__.py": """
        class TrackingEngine:
            STUB = True
            def __init__(self, modules, tracker_state, num_workers, callbacks=None):
                self.module_names = [m.name for m in modules]; self.tracker_state = tracker_state; self.num_workers = num_workers
        """,
}


def _write_stub(root):
    for rel, src in STUB.items():
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fp:
            fp.write(textwrap.dedent(src))


def _yamls():
    out = []
    for d, _, files in os.walk(CONFIGS):
        out += [os.path.join(d, f) for f in files if f.endswith(".yaml")]
    return sorted(out)


def _resolve(target):
    mod, _, attr = target.rpartition(".")
    return getattr(importlib.import_module(mod), attr)


def _instantiate(node, **overrides):
    """hydra.utils.instantiate for the subset our yamls use (no interpolation: `${...}` values are passed through as strings)."""
    kw = {k: (_instantiate(v) if isinstance(v, dict) and "_target_" in v else v) for k, v in node.items() if k != "_target_"}
    kw.update(overrides)
    return _resolve(node["_target_"])(**kw)


def test_every_yaml_target_resolves_and_accepts_its_keys():
    ys = _yamls()
    assert len(ys) >= 13 and any(y.endswith("engine/hip_fused.yaml") for y in ys)
    for y in ys:
        node = yaml.safe_load(open(y))
        todo = [node]
        while todo:
            n = todo.pop()
            todo += [v for v in n.values() if isinstance(v, dict) and "_target_" in v]
            if not n["_target_"].startswith("tracklab_amd."):
                continue                                           # tracklab.callbacks.*: the reference's own classes
            cls = _resolve(n["_target_"])
            kw = {k: v for k, v in n.items() if k != "_target_"}
            if "wrappers" in n["_target_"] and "Evaluator" not in n["_target_"]:
                kw.setdefault("device", "cuda"); kw.setdefault("batch_size", 1)     # what main.py's instantiate adds (main.py:36-39)
            if n["_target_"].endswith("Evaluator"):
                kw.update(tracking_dataset=None)                                    # main.py: instantiate(cfg.eval, tracking_dataset=...)
            if n["_target_"].endswith("HipTrackingEngine"):
                kw.update(modules=[], tracker_state=None)                          # main.py:55-59
            inspect.signature(cls.__init__).bind(None, **kw)                        # TypeError = a key the constructor does not take


def test_with_tracklab_importable_our_classes_subclass_its_abcs(tmp_path):
    _write_stub(str(tmp_path))
    code = """
        import tracklab.pipeline as tp, tracklab.engine as te
        from tracklab_amd import pipeline_api, engine, wrappers
        assert pipeline_api.HAVE_TRACKLAB and engine.HAVE_TRACKLAB_ENGINE
        assert pipeline_api.ImageLevelModule is tp.ImageLevelModule and getattr(tp.Module, "STUB")
        mods = [wrappers.HipYOLOX, wrappers.HipRTMPose, wrappers.HipPartReID, wrappers.HipOCSORT, wrappers.HipByteTrack,
                wrappers.HipBPBReIDStrongSORT, wrappers.HipStrongSORT, wrappers.HipBoTSORT, wrappers.HipDeepOCSORT]
        for m in mods:
            assert issubclass(m, tp.ImageLevelModule), m
            assert m.__bases__[0] is tp.ImageLevelModule, m        # `level` is read off the FIRST base (pipeline/module.py:33-37)
            assert isinstance(m.input_columns, (list, dict)) and isinstance(m.output_columns, (list, dict)), m
        assert issubclass(engine.HipTrackingEngine, te.TrackingEngine)
        print("stub ok", len(mods))
    """
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path), REPO, os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "stub ok 9" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_every_module_yaml_instantiates_on_the_device_under_the_stub(tmp_path):
    """The Hydra path of main.py:36-39 for every module yaml, with the stub TrackLab importable: constructors run (device banks, networks),
    `name` / `level` come from TrackLab's Module, and one module of each kind processes a frame."""
    _write_stub(str(tmp_path))
    code = f"""
        import sys, yaml, numpy as np, pandas as pd
        sys.path.insert(0, {os.path.join(REPO, "tests")!r})
        from test_stub_tracklab import _instantiate, _yamls
        import tracklab.pipeline as tp
        made = 0
        for y in _yamls():
            node = yaml.safe_load(open(y))
            if y.split("/")[-2] in ("engine", "eval"):
                continue
            m = _instantiate(node, device="cuda", batch_size=1)
            assert isinstance(m, tp.ImageLevelModule) and m.level == "image" and m.name == node["_target_"].rsplit(".", 1)[1], y
            made += 1
            if m.name == "HipOCSORT":
                img = np.zeros((1080, 1920, 3), np.uint8)
                det = pd.DataFrame({{"bbox_ltwh": [np.array([100., 100., 50., 120.]), np.array([600., 300., 60., 150.])], "bbox_conf": [0.9, 0.8],
                                    "category_id": [1, 1], "image_id": [0, 0], "video_id": [0, 0]}}, index=pd.Index([0, 1], name="id"))
                meta = pd.Series({{"id": 0, "video_id": 0, "frame": 0}}, name=0)
                for _ in range(3):
                    out = m.process(m.preprocess(img, det, meta), det, pd.DataFrame([meta]))
                assert len(out) == 2 and out.track_id.notna().all() and sorted(out.index) == [0, 1], out
        print("instantiated", made)
    """
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path), REPO, os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "instantiated 12" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
