"""Seeded synthetic 1080p multi-object streams.

One generator feeds the golden-vector script, the parity tests and ``bench.py`` so
that every consumer sees bit-identical inputs for a given ``(seed, n_objects)``.

Boxes follow the TrackLab conventions of the tracker wrappers
(``tracklab/wrappers/track/oc_sort_api.py:33-47``): a frame's detections are an
``(n, 7)`` float64 array ``[l, t, r, b, conf, cls, tracklab_id]``; the
BPBReID-StrongSORT wrapper (``bpbreid_strong_sort_api.py:73-99``) consumes
``bbox_ltwh (n,4) f64``, ``embeddings (n,K,D) f32``, ``visibility (n,K) bool``.

numpy only; no torch, no HIP (safe inside DataLoader worker processes).
"""
from __future__ import annotations

import numpy as np

WIDTH, HEIGHT = 1920, 1080


class SyntheticStream:
    """Deterministic stream of jittered boxes with births, deaths and misses.

    Parameters mirror SURVEY.md §8(d): centres U([100,1820]x[100,980]),
    w~U(40,120), h=w*U(1.8,2.6), velocity N(0,3)xN(0,2) px/frame reflected at the
    borders, per-frame box jitter N(0,1) px on all four coordinates (keeps
    association costs tie-free), conf~U(0.5,1), 2 % per-frame miss probability,
    a birth+death roughly every ``churn_period`` frames.
    """

    def __init__(self, seed: int, n_objects: int = 100, n_frames: int = 600,
                 parts: int = 6, dim: int = 256, with_embeddings: bool = False,
                 miss_prob: float = 0.02, churn_period: int = 100,
                 cls: float = 1.0, low_conf_frac: float = 0.0):
        self.seed = int(seed)
        self.n_objects = int(n_objects)
        self.n_frames = int(n_frames)
        self.parts = int(parts)
        self.dim = int(dim)
        self.with_embeddings = bool(with_embeddings)
        self.miss_prob = float(miss_prob)
        self.churn_period = int(churn_period)
        self.cls = float(cls)
        self.low_conf_frac = float(low_conf_frac)
        self._rng = np.random.default_rng(self.seed)
        self._next_gt = 0
        self._next_det = 0
        self._frame = 0
        n = self.n_objects
        self._cx = self._rng.uniform(100, WIDTH - 100, n)
        self._cy = self._rng.uniform(100, HEIGHT - 100, n)
        self._w = self._rng.uniform(40, 120, n)
        self._h = self._w * self._rng.uniform(1.8, 2.6, n)
        self._vx = self._rng.normal(0, 3, n)
        self._vy = self._rng.normal(0, 2, n)
        self._gt = np.arange(n, dtype=np.int64)
        self._next_gt = n
        if self.with_embeddings:
            self._proto = self._rng.normal(0, 1, (n, self.parts, self.dim)).astype(np.float32)
        else:
            self._proto = None

    # ------------------------------------------------------------------
    def _respawn(self, idx: int) -> None:
        r = self._rng
        self._cx[idx] = r.uniform(100, WIDTH - 100)
        self._cy[idx] = r.uniform(100, HEIGHT - 100)
        self._w[idx] = r.uniform(40, 120)
        self._h[idx] = self._w[idx] * r.uniform(1.8, 2.6)
        self._vx[idx] = r.normal(0, 3)
        self._vy[idx] = r.normal(0, 2)
        self._gt[idx] = self._next_gt
        self._next_gt += 1
        if self._proto is not None:
            self._proto[idx] = r.normal(0, 1, (self.parts, self.dim)).astype(np.float32)

    def step(self) -> dict:
        """Advance one frame and return its detections.

        Returns a dict with ``dets`` (n,7) f64, ``gt_ids`` (n,) i64, ``gt_boxes``
        (n_objects,4) ltrb of every live object (for HOTA), ``gt_all_ids`` and,
        if enabled, ``embeddings`` (n,K,D) f32 / ``visibility`` (n,K) bool.
        """
        r = self._rng
        n = self.n_objects
        if self._frame > 0:
            self._cx += self._vx
            self._cy += self._vy
            for c, v, lo, hi in ((self._cx, self._vx, 100.0, WIDTH - 100.0),
                                 (self._cy, self._vy, 100.0, HEIGHT - 100.0)):
                below = c < lo
                above = c > hi
                c[below] = 2 * lo - c[below]
                c[above] = 2 * hi - c[above]
                v[below | above] *= -1
            if self.churn_period > 0 and self._frame % max(1, self.churn_period // max(1, n // 50 + 1)) == 0:
                self._respawn(int(r.integers(0, n)))
        jitter = r.normal(0, 1, (n, 4))
        l = self._cx - self._w / 2 + jitter[:, 0]
        t = self._cy - self._h / 2 + jitter[:, 1]
        rr = self._cx + self._w / 2 + jitter[:, 2]
        b = self._cy + self._h / 2 + jitter[:, 3]
        conf = r.uniform(0.5, 1.0, n)
        if self.low_conf_frac > 0:
            low = r.uniform(0, 1, n) < self.low_conf_frac
            conf = np.where(low, r.uniform(0.05, 0.45, n), conf)
        keep = r.uniform(0, 1, n) >= self.miss_prob
        order = r.permutation(n)            # detectors do not emit boxes in identity order
        order = order[keep[order]]
        k = len(order)
        det_ids = np.arange(self._next_det, self._next_det + k, dtype=np.float64)
        self._next_det += k
        dets = np.empty((k, 7), dtype=np.float64)
        dets[:, 0] = l[order]
        dets[:, 1] = t[order]
        dets[:, 2] = rr[order]
        dets[:, 3] = b[order]
        dets[:, 4] = conf[order]
        dets[:, 5] = self.cls
        dets[:, 6] = det_ids
        gt_boxes = np.stack([self._cx - self._w / 2, self._cy - self._h / 2,
                             self._cx + self._w / 2, self._cy + self._h / 2], axis=1)
        out = {"frame": self._frame, "dets": dets, "gt_ids": self._gt[order].copy(),
               "gt_boxes": gt_boxes, "gt_all_ids": self._gt.copy()}
        if self._proto is not None:
            noise = r.normal(0, 0.1, (n, self.parts, self.dim)).astype(np.float32)
            emb_all = self._proto + noise
            vis_all = r.uniform(0, 1, (n, self.parts)) < 0.9
            vis_all[:, 0] = True
            out["embeddings"] = np.ascontiguousarray(emb_all[order])
            out["visibility"] = np.ascontiguousarray(vis_all[order])
        self._frame += 1
        return out

    def __iter__(self):
        for _ in range(self.n_frames):
            yield self.step()


def ltrb_to_ltwh_rows(ltrb: np.ndarray) -> np.ndarray:
    """(n,4) ltrb -> (n,4) ltwh; same arithmetic as ``coordinates.py:318-328`` per row."""
    out = np.empty_like(ltrb)
    out[:, 0] = ltrb[:, 0]
    out[:, 1] = ltrb[:, 1]
    out[:, 2] = ltrb[:, 2] - ltrb[:, 0]
    out[:, 3] = ltrb[:, 3] - ltrb[:, 1]
    return out


def render_frame(rng: np.random.Generator, gt_boxes: np.ndarray,
                 height: int = HEIGHT, width: int = WIDTH) -> np.ndarray:
    """uint8 HxWx3 frame: background noise + one flat-colour rectangle per object."""
    img = rng.integers(0, 64, (height, width, 3), dtype=np.uint8)
    for i, (l, t, r, b) in enumerate(gt_boxes):
        l, t = max(0, int(l)), max(0, int(t))
        r, b = min(width, int(r)), min(height, int(b))
        if r > l and b > t:
            img[t:b, l:r] = ((37 * i) % 200 + 55, (91 * i) % 200 + 55, (53 * i) % 200 + 55)
    return img


def synth_yolox_head(rng: np.random.Generator, boxes_xyxy: np.ndarray, size: int = 640, ratio: float = 1.0 / 3,
                     num_classes: int = 1, dup: int = 3) -> np.ndarray:
    """Raw YOLOX head tensor (A, 5+C) float32 whose rtmlib-style decode yields `boxes_xyxy` (image scale),
    plus lower-scored jittered duplicates on neighbouring anchors (NMS work) over sub-threshold clutter.
    Stands in for the activations of a trained detector: offline there are no checkpoints, and a
    random-init network detects nothing."""
    strides = [8, 16, 32]
    n = [(size // s) ** 2 for s in strides]
    A = sum(n)
    pred = np.zeros((A, 5 + num_classes), dtype=np.float32)
    pred[:, :4] = rng.normal(0, 0.5, (A, 4))
    pred[:, 4] = rng.uniform(0, 0.6, A)
    pred[:, 5:] = rng.uniform(0, 0.9, (A, num_classes))
    used = set()
    for k, (x1, y1, x2, y2) in enumerate(boxes_xyxy):
        cx, cy = (x1 + x2) / 2 * ratio, (y1 + y2) / 2 * ratio
        w, h = (x2 - x1) * ratio, (y2 - y1) * ratio
        lvl = 0 if max(w, h) < 64 else (1 if max(w, h) < 128 else 2)
        s = strides[lvl]
        ws = size // s
        for d in range(dup):
            gx = int(np.clip(cx // s + (d % 2) * (1 if d else 0), 0, ws - 1))
            gy = int(np.clip(cy // s + (d // 2), 0, ws - 1))
            a = sum(n[:lvl]) + gy * ws + gx
            if a in used:
                continue
            used.add(a)
            jit = rng.normal(0, 0.6, 4) if d else np.zeros(4)
            pred[a, 0] = (cx + jit[0]) / s - gx
            pred[a, 1] = (cy + jit[1]) / s - gy
            pred[a, 2] = np.log(max(w + jit[2], 1.0) / s)
            pred[a, 3] = np.log(max(h + jit[3], 1.0) / s)
            pred[a, 4] = rng.uniform(0.9, 1.0) if d == 0 else rng.uniform(0.85, 0.95)
            pred[a, 5 + (k % num_classes)] = rng.uniform(0.9, 1.0)
    return pred


_KP_REL = np.array([[0.50, 0.07], [0.56, 0.05], [0.44, 0.05], [0.63, 0.08], [0.37, 0.08], [0.75, 0.22], [0.25, 0.22],
                    [0.85, 0.38], [0.15, 0.38], [0.88, 0.52], [0.12, 0.52], [0.65, 0.55], [0.35, 0.55],
                    [0.66, 0.75], [0.34, 0.75], [0.67, 0.95], [0.33, 0.95]])


def synth_keypoints(rng: np.random.Generator, ltrb: np.ndarray, invisible_prob: float = 0.15) -> np.ndarray:
    """COCO-17 keypoints (n,17,3) [x, y, conf] placed at fixed relative positions inside each box + 2 px jitter;
    a random subset gets conf 0 (invisible; at least the shoulders and hips stay visible)."""
    n = len(ltrb)
    w, h = ltrb[:, 2] - ltrb[:, 0], ltrb[:, 3] - ltrb[:, 1]
    kp = np.empty((n, 17, 3))
    kp[:, :, 0] = ltrb[:, None, 0] + _KP_REL[None, :, 0] * w[:, None] + rng.normal(0, 2, (n, 17))
    kp[:, :, 1] = ltrb[:, None, 1] + _KP_REL[None, :, 1] * h[:, None] + rng.normal(0, 2, (n, 17))
    conf = rng.uniform(0.3, 1.0, (n, 17))
    conf[rng.uniform(0, 1, (n, 17)) < invisible_prob] = 0.0
    conf[:, [5, 6, 11, 12]] = np.maximum(conf[:, [5, 6, 11, 12]], 0.3)
    kp[:, :, 2] = conf
    return kp
