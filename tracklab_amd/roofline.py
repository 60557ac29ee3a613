"""Algorithmic byte counts used by bench.py's roofline block (definitions in DESIGN.md §4).

"Algorithmic" = bytes the algorithm must move once with perfect caching, at cache-line granularity of
whole source rows: every source row that cv2's INTER_LINEAR sampling touches is counted in full, every
output element is counted once. (SURVEY.md §8(d) budgets the whole 1080p frame, 6.22 MB, for the letterbox
read; with rtmlib/cv2 semantics at ratio 1/3 only every third row is ever sampled, so the honest figure is
2.07 MB. We report against the smaller number.)
"""
from __future__ import annotations

import math


def _cv_rows_touched(src: int, dst: int) -> int:
    """Number of distinct source rows read when resizing src -> dst rows (OpenCV fixed-point bilinear)."""
    rows = set()
    scale = src / dst
    for d in range(dst):
        f = (d + 0.5) * scale - 0.5
        s = math.floor(f)
        fr = f - s
        w1 = round(fr * 2048)
        rows.add(min(max(s, 0), src - 1))
        if w1 != 0:
            rows.add(min(max(s + 1, 0), src - 1))
    return len(rows)


def letterbox_bytes(h: int, w: int, size: int, rh: int, rw: int, elem_bytes: int = 2) -> int:
    """Per frame: touched source rows x full row bytes + the whole (3, size, size) output."""
    return _cv_rows_touched(h, rh) * w * 3 + 3 * size * size * elem_bytes


def crop_bytes(crop_h: float, crop_w: float, out_h: int, out_w: int, elem_bytes: int = 2) -> float:
    """Per crop: the crop's source pixels once + the (3, out_h, out_w) output."""
    return crop_h * crop_w * 3 + 3 * out_h * out_w * elem_bytes
