"""Per-video online loop over the fused GPU pipeline with a columnar, HBM-resident detection table.

The reference's engines walk ``pipeline=[bbox_detector, reid, track]`` module by module and glue the results with
``merge_dataframes`` once per module and batch (tracklab/engine/engine.py:18-41, 148-185; the true per-frame loop is
tracklab/engine/video.py:67-117): every step re-slices and re-merges a growing pandas frame. Here one video is

    frames (decoded once) -> H2D of ``frames_per_step`` frames (pinned, double-buffered, on a copy stream) -> ``pipeline.step``
    (letterbox, detector, decode+NMS, [pose,] [crop, ReID,] association: all on the GPU) -> device-to-device append of the step's
    rows to the per-video table in HBM

with NO host synchronisation inside the loop (``resident`` mode, the default): the table crosses PCIe once, at the end of the
video, and only then becomes a ``pd.DataFrame`` with the columns the reference's module chain would have produced
(``image_id, video_id, category_id, bbox_ltwh, bbox_conf`` from the detector, ``track_id`` + the tracker's box columns) indexed by
detection id. ``online`` mode copies every step's rows to pinned host memory instead and drains step k while step k+1 runs, so
per-image consumers (``on_image_loop_end`` callbacks: visualisation, live export) see detections one step late at most.

``HipTrackingEngine`` wraps the loop in TrackLab's engine contract (``TrackingEngine.__init__`` / ``track_dataset`` / callback
hooks, engine/engine.py:76-126) so that ``engine=hip_fused`` drops in next to ``offline`` / ``video``.
"""
from __future__ import annotations

import numpy as np
import pandas as pd
import torch

from . import _lib  # noqa: F401  (fails loudly when libtlk is missing: no CPU fallback)


class DetectionTable:
    """Columnar, append-only per-video detection table on the host (amortised O(1) appends of whole steps; no per-row Python objects)."""

    def __init__(self, capacity: int = 4096):
        self.n = 0
        self.cols = {"image_id": np.empty(capacity, np.int64), "bbox_ltwh": np.empty((capacity, 4), np.float32),
                     "bbox_conf": np.empty(capacity, np.float64), "category_id": np.empty(capacity, np.int64),
                     "track_id": np.full(capacity, np.nan), "track_bbox_ltwh": np.full((capacity, 4), np.nan),
                     "track_bbox_conf": np.full(capacity, np.nan)}
        self.index = np.empty(capacity, np.int64)

    def _grow(self, need):
        cap = len(self.index)
        if self.n + need <= cap:
            return
        new = max(2 * cap, self.n + need)
        self.index = np.resize(self.index, new)
        for k, v in self.cols.items():
            w = np.empty((new,) + v.shape[1:], v.dtype)
            if v.dtype.kind == "f" and k.startswith("track"):
                w[...] = np.nan
            w[:cap] = v
            self.cols[k] = w

    def append_step(self, first_frame, n_frames, id_base, maxd, ltwh, dcnt, trk):
        """One pipeline step, vectorised. ltwh (F, maxd, 4), dcnt (F,) detector output; detection ids are
        id_base + frame * maxd + i (what tlk_yolox_decode_nms wrote into the tracker rows); trk = (frame, det_id, track_id, ltwh,
        conf) flat arrays of the tracker's rows (``pipeline.track_columns``). Returns the slice of new table rows."""
        dcnt = np.asarray(dcnt[:n_frames], dtype=np.int64)
        if (dcnt < 0).any():
            raise RuntimeError("HipVideoEngine: detection capacity exceeded (more boxes than max_dets / NMS candidates)")
        f, i = np.nonzero(np.arange(maxd)[None, :] < dcnt[:, None])          # row-major: by frame, then by detection
        m = len(f)
        self._grow(m)
        s = slice(self.n, self.n + m)
        c = self.cols
        self.index[s] = id_base + f * maxd + i
        c["image_id"][s] = first_frame + f
        c["bbox_ltwh"][s] = ltwh[f, i]
        c["bbox_conf"][s] = 1.0                   # RTMLibDetector: bbox_conf 1.0, category 1 (rtmlib_api.py:36-45)
        c["category_id"][s] = 1
        c["track_id"][s] = np.nan; c["track_bbox_ltwh"][s] = np.nan; c["track_bbox_conf"][s] = np.nan
        tf, tdet, tid, tl, tconf = trk
        keep = tf < n_frames
        if keep.any():
            tf, tdet, tid, tl, tconf = tf[keep], tdet[keep], tid[keep], tl[keep], tconf[keep]
            off = np.concatenate([[0], np.cumsum(dcnt)[:-1]])
            rel = tdet - id_base
            rf, ri = rel // maxd, rel % maxd
            ok = (rel >= 0) & (rf < n_frames) & (ri < dcnt[np.clip(rf, 0, n_frames - 1)])
            rows = self.n + off[rf[ok]] + ri[ok]
            c["track_id"][rows] = tid[ok]
            c["track_bbox_ltwh"][rows] = tl[ok]
            c["track_bbox_conf"][rows] = tconf[ok]
        self.n += m
        return s

    def to_dataframe(self, video_id=0, rows: slice | None = None) -> pd.DataFrame:
        c = self.cols
        s = rows if rows is not None else slice(0, self.n)
        return pd.DataFrame({"image_id": c["image_id"][s], "video_id": video_id, "category_id": c["category_id"][s],
                             "bbox_ltwh": list(c["bbox_ltwh"][s]), "bbox_conf": c["bbox_conf"][s], "track_id": c["track_id"][s],
                             "track_bbox_ltwh": list(c["track_bbox_ltwh"][s]), "track_bbox_conf": c["track_bbox_conf"][s]},
                            index=pd.Index(self.index[s], name="id"))


class DeviceStepLog:
    """The per-video table while it lives in HBM: every step's result tensors appended by device-to-device copies on the
    association stream (chunks of ``chunk`` steps, no reallocation), fetched with ONE synchronisation at the end of the video."""

    def __init__(self, chunk: int = 64):
        self.chunk = chunk
        self.chunks = []            # dict name -> (chunk, ...) device tensor
        self.meta = []              # (first frame, frames in the step, id_base) per step
        self.n = 0

    def sink(self, first_frame, n_frames, id_base):
        k = self.n
        self.n += 1
        self.meta.append((first_frame, n_frames, id_base))

        def write(res):
            ci, si = divmod(k, self.chunk)
            if ci == len(self.chunks):
                self.chunks.append({name: torch.empty((self.chunk,) + tuple(t.shape), dtype=t.dtype, device=t.device) for name, t in res.items()})
            for name, t in res.items():
                self.chunks[ci][name][si].copy_(t, non_blocking=True)
        return write

    def fetch(self):
        """-> list of (first frame, n frames, id_base, dict of numpy arrays) per step; ONE device synchronisation."""
        host = [{name: t[:min(self.chunk, self.n - ci * self.chunk)].to("cpu", non_blocking=False) for name, t in ch.items()}
                for ci, ch in enumerate(self.chunks)]
        out = []
        for k, (first, n, idb) in enumerate(self.meta):
            ci, si = divmod(k, self.chunk)
            out.append((first, n, idb, {name: t[si].numpy() for name, t in host[ci].items()}))
        return out


class HipVideoEngine:
    """One video through a ``gpu_pipeline.DetTrackPipeline`` (detector + OC-SORT / ByteTrack) or ``DetReidTrackPipeline`` (detector +
    [pose +] ReID + BPBReID-StrongSORT / plain StrongSORT / BoT-SORT / Deep-OC-SORT) with ``n_streams == 1``. ``video_loop`` mirrors
    ``VideoOnlineTrackingEngine.video_loop`` (engine/video.py:67-117): modules are reset, frames go through in order, one
    detections frame comes back."""

    def __init__(self, pipeline):
        assert pipeline.S == 1, "one video = one stream"
        self.pipe = pipeline
        self.F, self.maxd = pipeline.F, pipeline.maxd
        H, W = pipeline.H, pipeline.W
        self._pinned = [torch.empty((self.F, H, W, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
        self._dev = [torch.empty((self.F, H, W, 3), dtype=torch.uint8, device=pipeline.dev) for _ in range(2)]
        self._h2d_done = [torch.cuda.Event(), torch.cuda.Event()]
        self._frames_free = [None, None]
        # r06: a copy stream MEASURED to run beside the compute stream and the pipeline's own side streams (gpu_pipeline.pick_stream: about one
        # pool stream in four shares the compute stream's hardware queue, and an upload behind it does not overlap anything)
        from .gpu_pipeline import pick_stream
        others = [torch.cuda.current_stream(pipeline.dev)] + [getattr(pipeline, n_, None) for n_ in ("trk_stream", "det_stream", "reid_stream")]
        self._copy_stream = pick_stream(pipeline.dev, others)
        self.h2d_bytes = 0

    # ---- frame batching -------------------------------------------------------------------------------------------------
    def _batches(self, frames):
        """-> (cpu uint8 tensor (n, H, W, 3) with n <= F, is_pinned) per step. Pinned (n, H, W, 3) tensors pass through untouched
        (a decoder writing into pinned memory); single frames / numpy arrays are packed into the engine's own pinned staging buffers."""
        F = self.F
        k = 0
        if isinstance(frames, np.ndarray) and frames.ndim == 4:
            frames = iter(frames)
        it = iter(frames)
        pending = None
        while True:
            first = pending if pending is not None else next(it, None)
            pending = None
            if first is None:
                return
            if torch.is_tensor(first) and first.dim() == 4:
                assert first.shape[0] <= F and first.dtype == torch.uint8
                yield first, first.is_pinned()
                continue
            slot = k % 2
            self._h2d_done[slot].synchronize()           # the H2D that last read this staging buffer (two steps ago) has finished
            buf = self._pinned[slot]
            n = 0
            fr = first
            while fr is not None:
                buf[n].copy_(torch.from_numpy(np.ascontiguousarray(fr)) if not torch.is_tensor(fr) else fr)
                n += 1
                if n == F:
                    break
                fr = next(it, None)
                if fr is not None and torch.is_tensor(fr) and fr.dim() == 4:
                    pending, fr = fr, None
            k += 1
            yield buf[:n], True

    @torch.no_grad()
    def video_loop(self, frames, video_id=0, synth_heads=None, online=False, on_step=None, keep_ids=False, fetch_table=True) -> pd.DataFrame:
        """frames: (T, H, W, 3) uint8 RGB array, an iterable of (H, W, 3) frames, or an iterable of pinned (n <= F, H, W, 3) uint8
        tensors (uploaded without a staging copy). RGB like TrackLab's cv2_load_image (channel contract: gpu_pipeline docstring).
        synth_heads: optional callable(first_frame, n) -> (n, A, 6) float32 detector-head activations (numpy, or a cuda tensor of
        a full step) replacing the network's (random-init detectors produce no boxes; benchmarks and tests feed a head that encodes
        known boxes). online: drain every step to the host while the next one runs and call on_step(detections_of_the_step: DataFrame)
        -- the per-image hook of the reference's online engine; otherwise the table stays in HBM until the video ends.
        keep_ids: the tracker is reset for the video but ByteTrack's / BoT-SORT's id counter keeps counting, like the reference's class-level
        BaseTrack._count (the other trackers restart at 1 per video in the reference too: their counters are re-created with the tracker)."""
        pipe, F, maxd = self.pipe, self.F, self.maxd
        pipe.reset(keep_ids=keep_ids)
        table = DetectionTable()
        log = None if online else DeviceStepLog()
        main = torch.cuda.current_stream(pipe.dev)
        t0, k, pending = 0, 0, None
        for src, pinned in self._batches(frames):
            n = src.shape[0]
            slot = k % 2
            dev = self._dev[slot]
            with torch.cuda.stream(self._copy_stream):       # H2D of step k overlaps the kernels of step k-1
                if self._frames_free[slot] is not None:
                    self._copy_stream.wait_event(self._frames_free[slot])      # step k-2 no longer reads this device buffer
                dev[:n].copy_(src, non_blocking=pinned)
                self._h2d_done[slot].record(self._copy_stream)
            self.h2d_bytes += src.numel()
            main.wait_event(self._h2d_done[slot])
            head = None
            if synth_heads is not None:
                h = synth_heads(t0, n)
                if torch.is_tensor(h) and h.is_cuda and h.shape[0] == F:
                    head = h
                else:
                    hn = np.zeros((F,) + tuple(h.shape[1:]), np.float32)
                    hn[..., 4] = -1.0                         # padded frames of the last step: no detections
                    hn[:n] = h.cpu().numpy() if torch.is_tensor(h) else h
                    head = torch.from_numpy(hn).to(pipe.dev)
            elif n < F:
                dev[n:].zero_()
            id_base = pipe.frames_done * maxd
            if online:
                pipe.step(dev, head, fetch=True)
                step = (t0, n, id_base, pipe.last)
            else:
                pipe.step(dev, head, fetch=False, sink=log.sink(t0, n, id_base))
                step = None
            self._frames_free[slot] = pipe.frames_free
            t0 += n
            k += 1
            if pending is not None:
                self._drain(table, pending, video_id, on_step)
            pending = step
        if pending is not None:
            self._drain(table, pending, video_id, on_step)
        self.last_log = log                     # the video's table where it was written: HBM (tracklab_amd.evaluate.evaluate_device_log reads it there)
        if log is not None:
            pipe.synchronize()                  # the appends ran on the association stream: they must be complete before the ONE fetch
            if not fetch_table:
                return None
            for first, n, idb, res in log.fetch():
                table.append_step(first, n, idb, maxd, res["ltwh"], res["dcnt"], self._trk_columns(res))
        return table.to_dataframe(video_id)

    def _trk_columns(self, res):
        if (np.asarray(res["ocnt"]) < 0).any():
            raise RuntimeError("HipVideoEngine: tracker capacity exceeded")
        return self.pipe.track_columns(res["rows"], np.asarray(res["ocnt"], dtype=np.int64))

    def _drain(self, table, step, video_id, on_step):
        first, n, id_base, buf = step
        buf[self.pipe.done_key].synchronize()     # this step's association + pinned result copies; the next step keeps running
        res = self.pipe.host_results(buf)
        rows = table.append_step(first, n, id_base, self.maxd, res["ltwh"], res["dcnt"], self._trk_columns(res))
        if on_step is not None:
            on_step(table.to_dataframe(video_id, rows))


class _CallbackList:
    """``fabric.call(name, **kwargs)`` of the reference's engines (lightning.Fabric is used only as a callback dispatcher,
    engine/engine.py:92-93): call ``name`` on every callback that defines it."""

    def __init__(self, callbacks):
        self.callbacks = list(callbacks)

    def call(self, name, *args, **kwargs):
        for cb in self.callbacks:
            fn = getattr(cb, name, None)
            if callable(fn):
                fn(*args, **kwargs)


try:  # pragma: no cover - only where TrackLab (and lightning) are installed
    from tracklab.engine import TrackingEngine as _EngineBase  # type: ignore
    HAVE_TRACKLAB_ENGINE = True
except Exception:
    _EngineBase = object
    HAVE_TRACKLAB_ENGINE = False


class HipTrackingEngine(_EngineBase):
    """TrackLab engine whose ``video_loop`` is the fused GPU pipeline (one ``HipVideoEngine`` per video) instead of the
    module-by-module dataloader walk. Same constructor and hooks as ``TrackingEngine`` (engine/engine.py:76-126):
    ``on_dataset_track_start/end``, ``on_video_loop_start/end`` around every video, ``on_module_step_start/end`` around every fused
    step (task = "hip_fused_pipeline"), and -- when a callback defines ``on_image_loop_end`` -- the per-image hooks of the online
    engine (engine/video.py:93-117), fed from the ``online`` drain. ``modules`` may be empty: the pipeline replaces them.

    image_loader(file_path) -> (H, W, 3) uint8 RGB array (default: Pillow, like cv2_load_image's RGB output).
    The fused pipeline is announced to the callbacks like a module of the reference (``on_module_start`` with a sized stand-in for the
    dataloader, one ``on_module_step_end`` per drained step, ``on_module_end``), so ``tracklab.callbacks.Progressbar`` works unchanged.
    ``reset_ids_per_video`` (default false): detection ids keep counting across the videos of a dataset like the reference detector's
    counter, and ByteTrack / BoT-SORT track ids like ``BaseTrack._count``; true restarts both per video."""

    TASK = "hip_fused_pipeline"

    def __init__(self, modules=None, tracker_state=None, num_workers: int = 0, callbacks=None, pipeline=None, image_loader=None,
                 synth_heads=None, reset_ids_per_video: bool = False, **_unused):
        self.module_names = [m.name for m in (modules or [])]
        self.callbacks = dict(callbacks or {})
        cbs = list(self.callbacks.values())
        before = [c for c in cbs if not getattr(c, "after_saved_state", False)]
        after = [c for c in cbs if getattr(c, "after_saved_state", False)]
        self.fabric = _CallbackList(before + ([tracker_state] if tracker_state is not None else []) + after)
        self.callback = lambda name, **kw: self.fabric.call(name, engine=self, **kw)
        self.num_workers = int(num_workers) if isinstance(num_workers, (int, float)) or str(num_workers).lstrip("-").isdigit() else 0
        self.tracker_state = tracker_state
        self.img_metadatas = getattr(tracker_state, "image_metadatas", None)
        self.video_metadatas = getattr(tracker_state, "video_metadatas", None)
        self.models = {m.name: m for m in (modules or [])}
        self.reset_ids_per_video = bool(reset_ids_per_video)
        self._det_id_offset = 0          # detection ids are dataset-global like the reference detector's running counter (rtmlib_api.py self.id)
        if pipeline is None:
            raise ValueError("HipTrackingEngine needs a gpu_pipeline.DetTrackPipeline / DetReidTrackPipeline (engine.pipeline in the yaml)")
        self.pipeline = pipeline
        self.models[self.TASK] = pipeline         # callbacks look the task up here (callbacks/progress.py:59-75 reads engine.models[task])
        self.video_engine = HipVideoEngine(pipeline)
        self.image_loader = image_loader or self._load_rgb
        self.synth_heads = synth_heads
        self._per_image = any(callable(getattr(c, "on_image_loop_end", None)) for c in cbs)

    @staticmethod
    def _load_rgb(path):
        from PIL import Image
        return np.asarray(Image.open(path).convert("RGB"))

    def _decode_ahead(self, paths):
        """The frames of a video in order, decoded by ``num_workers`` threads up to two steps ahead of the GPU (the role of the reference
        engines' DataLoader workers, engine/engine.py:128-146: image decoding releases the GIL, so threads suffice and no frame is pickled
        between processes). num_workers <= 0: decode in line."""
        if self.num_workers <= 0 or len(paths) == 0:
            for p in paths:
                yield self.image_loader(p)
            return
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        depth = max(2 * self.video_engine.F, self.num_workers)
        with ThreadPoolExecutor(max_workers=self.num_workers) as pool:
            pending = deque()
            it = iter(paths)
            for p in it:
                pending.append(pool.submit(self.image_loader, p))
                if len(pending) >= depth:
                    break
            while pending:
                img = pending.popleft().result()
                nxt = next(it, None)
                if nxt is not None:
                    pending.append(pool.submit(self.image_loader, nxt))
                yield img

    def track_dataset(self):
        """engine/engine.py:105-126."""
        self.callback("on_dataset_track_start")
        for i, (video_idx, video_metadata) in enumerate(self.video_metadatas.iterrows()):
            ctx = self.tracker_state(video_idx) if callable(self.tracker_state) else _Null(self.tracker_state)
            with ctx as tracker_state:
                self.callback("on_video_loop_start", video_metadata=video_metadata, video_idx=video_idx, index=i)
                detections, image_pred = self.video_loop(tracker_state, video_metadata, video_idx)
                self.callback("on_video_loop_end", video_metadata=video_metadata, video_idx=video_idx, detections=detections,
                              image_pred=image_pred)
        self.callback("on_dataset_track_end")

    def video_loop(self, tracker_state, video_metadata, video_id):
        imgs = self.img_metadatas[self.img_metadatas.video_id == video_id]
        image_ids = imgs.index.to_numpy()
        paths = imgs.file_path.to_list()
        F = self.video_engine.F

        task = self.TASK
        n_steps = (len(paths) + F - 1) // F
        # one tick per drained step in the online form, one per video otherwise (the table crosses PCIe once)
        self.callback("on_module_start", task=task, dataloader=range(n_steps if self._per_image else 1))

        def frames():
            for j, img in enumerate(self._decode_ahead(paths)):
                if j % F == 0:
                    self.callback("on_module_step_start", task=task, batch=(image_ids[j:j + F], None))
                yield img

        off = self._det_id_offset

        def on_step(df):
            df = df.assign(image_id=image_ids[df.image_id.to_numpy()])
            df.index = df.index + off
            self.callback("on_module_step_end", task=task, batch=None, detections=df)
            for img_id, sub in df.groupby("image_id"):
                self.callback("on_image_loop_end", image_metadata=imgs.loc[img_id], image=None, image_idx=img_id, detections=sub)

        heads = None
        if self.synth_heads is not None:
            heads = lambda t0, n: self.synth_heads(video_id, t0, n)      # noqa: E731
        det = self.video_engine.video_loop(frames(), video_id=video_id, synth_heads=heads, online=self._per_image,
                                           on_step=on_step if self._per_image else None, keep_ids=not self.reset_ids_per_video)
        det["image_id"] = image_ids[det.image_id.to_numpy()] if len(det) else det.image_id
        det.index = det.index + off
        if not self.reset_ids_per_video:
            self._det_id_offset += n_steps * F * self.video_engine.maxd      # the id space one video occupies (id = frame * max_dets + i)
        if not self._per_image:
            self.callback("on_module_step_end", task=task, batch=None, detections=det)
        self.callback("on_module_end", task=task, detections=det)
        return det, imgs


class _Null:
    def __init__(self, v):
        self.v = v

    def __enter__(self):
        return self.v

    def __exit__(self, *a):
        return False
