"""Per-video online loop over the fused GPU pipeline with a columnar detection table.

The reference's engines walk ``pipeline=[bbox_detector, reid, track]`` module by module and glue the results with
``merge_dataframes`` once per module and batch (tracklab/engine/engine.py:18-41, 148-185; the true per-frame loop is
tracklab/engine/video.py:67-117): every step re-slices and re-merges a growing pandas frame. Here one video is

    decode once -> H2D of ``frames_per_step`` frames (pinned, double-buffered) -> ``pipeline.step`` (letterbox, detector,
    decode+NMS, [pose,] [crop, ReID,] association: all on the GPU, no host round trip) -> D2H of the small result block

and the per-video table is columnar numpy that only becomes a ``pd.DataFrame`` once, at the end, with the columns the
reference's module chain would have produced (``image_id, video_id, category_id, bbox_ltwh, bbox_conf`` from the detector,
``track_id`` + the tracker's box columns) indexed by detection id.
"""
from __future__ import annotations

import numpy as np
import pandas as pd
import torch

from . import _lib


class DetectionTable:
    """Columnar, append-only per-video detection table (amortised O(1) appends of whole frames; no per-row Python objects)."""

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

    def append_frame(self, image_id, det_ids, ltwh, conf, category):
        m = len(det_ids)
        self._grow(m)
        s = slice(self.n, self.n + m)
        self.index[s] = det_ids
        c = self.cols
        c["image_id"][s] = image_id; c["bbox_ltwh"][s] = ltwh; c["bbox_conf"][s] = conf; c["category_id"][s] = category
        c["track_id"][s] = np.nan; c["track_bbox_ltwh"][s] = np.nan; c["track_bbox_conf"][s] = np.nan
        self.n += m
        return s.start

    def set_tracks(self, base, det_ids_frame, row_det_ids, track_ids, track_ltwh, track_conf):
        """Tracker rows of one frame keyed by detection id -> table rows [base, base + len(det_ids_frame)) (ids ascending)."""
        if len(row_det_ids) == 0 or len(det_ids_frame) == 0:
            return
        pos = np.searchsorted(det_ids_frame, row_det_ids)
        ok = (pos < len(det_ids_frame)) & (det_ids_frame[np.minimum(pos, len(det_ids_frame) - 1)] == row_det_ids)
        rows = base + pos[ok]
        self.cols["track_id"][rows] = track_ids[ok]
        self.cols["track_bbox_ltwh"][rows] = track_ltwh[ok]
        self.cols["track_bbox_conf"][rows] = track_conf[ok]

    def to_dataframe(self, video_id=0) -> pd.DataFrame:
        n, c = self.n, self.cols
        return pd.DataFrame({"image_id": c["image_id"][:n], "video_id": video_id, "category_id": c["category_id"][:n],
                             "bbox_ltwh": list(c["bbox_ltwh"][:n]), "bbox_conf": c["bbox_conf"][:n], "track_id": c["track_id"][:n],
                             "track_bbox_ltwh": list(c["track_bbox_ltwh"][:n]), "track_bbox_conf": c["track_bbox_conf"][:n]},
                            index=pd.Index(self.index[:n], name="id"))


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
        self._copy_stream = torch.cuda.Stream(device=pipeline.dev)
        self._is_reid = hasattr(pipeline, "reid")

    @torch.no_grad()
    def video_loop(self, frames, video_id=0, rgb=True, synth_heads=None) -> pd.DataFrame:
        """frames: (T, H, W, 3) uint8 array or an iterable of (H, W, 3) frames (RGB like cv2_load_image unless rgb=False).
        synth_heads: optional callable(first_frame, n) -> (n, A, 6) float32 detector-head activations replacing the network's
        (random-init detectors produce no boxes; the benchmarks and tests feed a head that encodes known boxes)."""
        pipe, F, maxd = self.pipe, self.F, self.maxd
        pipe.reset()
        table = DetectionTable()
        it = iter(frames)
        t0, done, k = 0, False, 0
        pending = None                      # (first frame index, frames in the step, result handles)
        main = torch.cuda.current_stream(pipe.dev)
        while not done or pending is not None:
            step = None
            if not done:
                buf = self._pinned[k % 2]
                n = 0
                for fr in it:
                    buf[n].copy_(torch.from_numpy(np.ascontiguousarray(fr)))
                    n += 1
                    if n == F:
                        break
                if n < F:
                    done = True
                if n > 0:
                    if n < F:
                        buf[n:].zero_()
                    dev = self._dev[k % 2]
                    with torch.cuda.stream(self._copy_stream):       # H2D of step k overlaps the kernels of step k-1
                        self._copy_stream.wait_stream(main)
                        dev.copy_(buf, non_blocking=True)
                        if rgb:
                            dev.copy_(dev.flip(-1))                   # the reference's detector re-reads the file as BGR
                    main.wait_stream(self._copy_stream)
                    head = None
                    if synth_heads is not None:
                        h = np.zeros((F,) + synth_heads(t0, 1).shape[1:], np.float32)
                        h[..., 4] = -1.0                              # padded frames of the last step: no detections
                        h[:n] = synth_heads(t0, n)
                        head = torch.from_numpy(h).to(pipe.dev)
                    h_rows, h_cnt = pipe.step(dev, head)
                    det = pipe.det if self._is_reid else pipe.last["det"]
                    h_ltwh = det["ltwh"].to("cpu", non_blocking=True)
                    h_dcnt = det["counts"].to("cpu", non_blocking=True)
                    ev = pipe.last["done" if self._is_reid else "trk_done"]
                    step = (t0, n, h_rows, h_cnt, h_ltwh, h_dcnt, (pipe.frames_done - F) * maxd, ev)
                    t0 += n
                    k += 1
            if pending is not None:
                self._drain(table, pending)
            pending = step
            if pending is not None and done:
                self._drain(table, pending)
                pending = None
        return table.to_dataframe(video_id)

    def _drain(self, table, step):
        first, n, h_rows, h_cnt, h_ltwh, h_dcnt, id_base, ev = step
        pipe, maxd = self.pipe, self.maxd
        ev.synchronize()                      # this step's association + result copies; the next step keeps running
        ltwh, dcnt = h_ltwh.numpy(), h_dcnt.numpy()
        if self._is_reid:
            rows_sf, cnt = pipe.rows_numpy(h_rows, h_cnt)
        else:
            rows_arr, cnt = h_rows.numpy(), h_cnt.numpy()
        for f in range(n):
            m = int(dcnt[f])
            if m < 0 or int(cnt[0][f] if self._is_reid else cnt[0, f]) < 0:
                raise RuntimeError("HipVideoEngine: detection / tracker capacity exceeded")
            det_ids = id_base + f * maxd + np.arange(m, dtype=np.int64)
            base = table.append_frame(first + f, det_ids, ltwh[f, :m], 1.0, 1)         # RTMLibDetector: bbox_conf 1.0, category 1
            if self._is_reid:
                r = rows_sf[0][f]
                names = r.dtype.names
                if "kf_ltwh" in names:               # BPBReID-StrongSORT rows
                    tl, conf = r["kf_ltwh"], np.ones(len(r))
                else:                                # plain StrongSORT / BoT-SORT / Deep-OC-SORT rows: ltrb + the tracker's confidence column
                    b = r["ltrb"]
                    tl = np.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], axis=1) if len(r) else np.zeros((0, 4))
                    conf = r["conf"] if "conf" in names else r["score"]
                table.set_tracks(base, det_ids, r["det_id"].astype(np.int64), r["track_id"].astype(np.float64), tl, conf)
            else:
                r = rows_arr[0, f, :int(cnt[0, f])]
                tl = np.stack([r[:, 0], r[:, 1], r[:, 2] - r[:, 0], r[:, 3] - r[:, 1]], axis=1) if len(r) else np.zeros((0, 4))
                table.set_tracks(base, det_ids, r[:, 7].astype(np.int64), r[:, 4], tl, r[:, 6])
