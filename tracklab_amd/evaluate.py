"""Evaluation of MOTChallenge-format results from memory or from files: HOTA (tracklab_amd.hota, TrackEval's
definition) and the CLEAR-MOT / ID measures (tracklab_amd.clearmot, py-motmetrics' definition) for a set of sequences, per sequence
and combined the way the two libraries combine them (sums of sufficient statistics) -- the same vectors a multi-GPU run
all-reduces. Mirrors what tracklab/wrappers/eval/trackeval_evaluator.py does with pip `trackeval` on the files
TrackingDataset.save_for_eval wrote: here the rows can also come straight from tables.

    python -m tracklab_amd.evaluate GT_DIR PRED_DIR [--gpu]     # <name>.txt in both; prints one JSON object; --gpu: both evaluators on the device
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

from . import clearmot, hota, mot_io


def _by_frame(rows: dict):
    """MOT rows (frame 1-based, track_id, ltwh) -> {frame: (ids, ltwh)} with frames in ascending order."""
    out = {}
    if len(rows["frame"]):
        order = np.argsort(rows["frame"], kind="stable")
        fr, ids, box = rows["frame"][order], rows["track_id"][order], np.asarray(rows["ltwh"], dtype=np.float64)[order]
        cut = np.flatnonzero(np.diff(fr)) + 1
        for f, i, b in zip(fr[np.r_[0, cut]], np.split(ids, cut), np.split(box, cut)):
            out[int(f)] = (i, b)
    return out


def evaluate_sequence(gt: dict, pred: dict, n_frames: int | None = None, max_iou: float = 0.5, device: str = "cpu") -> dict:
    """gt / pred: dicts with 'frame' (1-based), 'track_id', 'ltwh' arrays (what mot_io.load_mot returns).
    -> {'hota': packed HOTA statistics, 'clear': CLEAR-MOT / ID counts} of this sequence (summable across sequences).
    device "gpu": both evaluators run on the MI355X (tlk_hota_sequence_f64, tlk_clear_sequence_f64; at most 512 boxes per frame and side) --
    same statistics (HOTA's floating-point sums to ~1e-15, every count exactly); there is no silent fall-back to the host path."""
    if device not in ("cpu", "gpu"):
        raise ValueError(f"evaluate_sequence: device must be 'cpu' or 'gpu', not {device!r}")
    g, p = _by_frame(gt), _by_frame(pred)
    last = n_frames if n_frames is not None else max([0] + list(g) + list(p))
    empty = (np.zeros(0, np.int64), np.zeros((0, 4)))
    acc = clearmot.MOTAccumulator()
    gt_fr, pr_fr, clear_fr = [], [], []
    to_ltrb = lambda b: np.column_stack([b[:, 0], b[:, 1], b[:, 0] + b[:, 2], b[:, 1] + b[:, 3]]).reshape(-1, 4)
    for f in range(1, last + 1):
        gi, gb = g.get(f, empty)
        pi, pb = p.get(f, empty)
        if device == "cpu":
            acc.update_boxes(gi, gb, pi, pb, max_iou=max_iou)
        else:
            clear_fr.append((gi, gb, pi, pb))
        gt_fr.append((gi, to_ltrb(gb)))
        pr_fr.append((pi, to_ltrb(pb)))
    if not last:
        return {"hota": np.zeros(len(hota.ALPHAS) * 7 + 2), "clear": acc.counts()}
    if device == "gpu":
        return {"hota": hota.pack(hota.hota_sequence_gpu(gt_fr, pr_fr), frames=float(last)), "clear": clearmot.sequence_counts_gpu(clear_fr, max_iou=max_iou)}
    return {"hota": hota.pack(hota.hota_sequence(*hota.sequence_from_rows(gt_fr, pr_fr)), frames=float(last)), "clear": acc.counts()}


def _device_field(rows_u8, row_dtype, name):
    """Field `name` of a (frames, cap, row bytes) uint8 device block laid out like numpy structured dtype `row_dtype` -> float64 tensor
    (frames, cap[, k]); a strided byte view turned contiguous on the device, no host copy."""
    import torch
    dt, off = row_dtype.fields[name][:2]
    base, shape = (dt.subdtype[0], dt.subdtype[1]) if dt.subdtype else (dt, ())
    n = int(np.prod(shape)) if shape else 1
    tdt = {np.dtype("float64"): torch.float64, np.dtype("float32"): torch.float32, np.dtype("int64"): torch.int64, np.dtype("int32"): torch.int32}[np.dtype(base)]
    v = rows_u8[..., off:off + n * base.itemsize].contiguous().view(tdt).to(torch.float64)
    return v if n > 1 else v[..., 0]


def device_log_tracks(log, pipe):
    """The tracker side of an HBM-resident per-video table (engine.DeviceStepLog) in the evaluators' layout, built ON THE DEVICE with the
    semantics of engine.DetectionTable.append_step (= the reference's merge of a module's rows into the detections by index): a tracker row
    belongs to the DETECTION its det_id names -- (det_id - id_base of the step) // max_dets is the frame inside the step, % max_dets the slot;
    rows that name a detection outside the step or beyond the frame's count are dropped; when several rows name one detection (a coasting
    plain-StrongSORT track reports its previous detection's id) the LAST one in table order wins, like a fancy-index assignment. Output:
    frames in order, a frame's tracked detections in slot order; ids dense 0..n-1 in the sorted order of the track ids (np.unique's inverse);
    boxes as ltwh and ltrb float64; int64 frame offsets. The only host traffic is ONE fetch of three scalars (rows, distinct ids, smallest
    row count) -- never the table. -> dict of device tensors + those scalars."""
    import torch
    if not log.meta:                                                                   # a video without a single step
        z = torch.zeros(1, dtype=torch.int64)
        return {"ids": z.to(torch.int32), "ltrb": torch.zeros((1, 4), dtype=torch.float64), "ltwh": torch.zeros((1, 4), dtype=torch.float64), "off": z, "count": z[:0],
                "n_boxes": 0, "n_ids": 0, "n_frames": 0, "cap": int(pipe.maxd)}
    steps = [(log.chunks[k // log.chunk], k % log.chunk, n) for k, (_, n, _) in enumerate(log.meta)]
    rows = torch.cat([ch["rows"][si][:n] for ch, si, n in steps])                      # (frames, cap, 8 doubles | row bytes)
    ocnt = torch.cat([ch["ocnt"][si][:n] for ch, si, n in steps]).to(torch.int64)
    dcnt = torch.cat([ch["dcnt"][si][:n] for ch, si, n in steps]).to(torch.int64) if "dcnt" in steps[0][0] else None
    dev = rows.device
    T, cap, maxd = rows.shape[0], rows.shape[1], int(pipe.maxd)
    # per frame (host-known step geometry): first frame of its step, frames in its step, id_base of its step
    first = torch.tensor(np.concatenate([np.full(n, f0, np.int64) for f0, n, _ in log.meta]), device=dev)
    nstep = torch.tensor(np.concatenate([np.full(n, n, np.int64) for _, n, _ in log.meta]), device=dev)
    idb = torch.tensor(np.concatenate([np.full(n, ib, np.int64) for _, n, ib in log.meta]), device=dev)
    first = first - int(log.meta[0][0])                                                # frame index inside this table
    rd = pipe.row_dtype
    if rd is None:                                                                     # OC-SORT rows: 8 doubles [x1, y1, x2, y2, track_id, cls, conf, det]
        tid, ltrb, det = rows[..., 4], rows[..., :4], rows[..., 7]
    else:
        u8 = rows.view(torch.uint8).reshape(T, cap, -1)
        tid, det = _device_field(u8, rd, "track_id"), _device_field(u8, rd, "det_id")
        if "kf_ltwh" in rd.names:                                                      # BPBReID-StrongSORT: the Kalman box, ltwh
            b = _device_field(u8, rd, "kf_ltwh")
            ltrb = torch.stack([b[..., 0], b[..., 1], b[..., 0] + b[..., 2], b[..., 1] + b[..., 3]], dim=-1)
        else:
            ltrb = _device_field(u8, rd, "ltrb")
    slot_ok = (torch.arange(cap, device=dev)[None, :] < ocnt[:, None]) & ~torch.isnan(tid)
    rel = torch.nan_to_num(det).to(torch.int64) - idb[:, None]
    rf, ri = torch.div(rel, maxd, rounding_mode="floor"), torch.remainder(rel, maxd)
    tgt_frame = first[:, None] + rf
    ok = slot_ok & (rel >= 0) & (rf < nstep[:, None])
    if dcnt is not None:
        ok = ok & (ri < dcnt[tgt_frame.clamp(0, T - 1)])
    # one winner per detection: the last row (table order) that names it -- scatter-max of the row index, deterministic
    key = torch.where(ok, tgt_frame * maxd + ri, torch.full_like(rel, T * maxd)).reshape(-1)
    src = torch.arange(T * cap, device=dev, dtype=torch.int64)
    winner = torch.full((T * maxd + 1,), -1, dtype=torch.int64, device=dev).scatter_reduce_(0, key, src, reduce="amax", include_self=True)[:T * maxd]
    has = winner >= 0                                                                  # (frames * maxd,) in frame-major, slot-minor order
    cnt = has.reshape(T, maxd).sum(dim=1)
    off = torch.zeros(T + 1, dtype=torch.int64, device=dev)
    torch.cumsum(cnt, 0, out=off[1:])
    w = winner.clamp(min=0)
    ids_all = torch.nan_to_num(tid).reshape(-1).to(torch.int64)[w]
    box_all = ltrb.reshape(-1, 4)[w]
    # stable compaction without a data-dependent shape: destination = rank among the tracked detections, the others go to a dump slot
    dest = torch.where(has, torch.cumsum(has, 0) - 1, torch.full_like(w, T * maxd))
    ids_raw = torch.zeros(T * maxd + 1, dtype=torch.int64, device=dev).scatter_(0, dest, ids_all)
    box = torch.zeros((T * maxd + 1, 4), dtype=torch.float64, device=dev).index_copy_(0, dest, box_all.contiguous())
    # dense ids in sorted order: presence flags over [0, id bound) -> scan.  The ids are taken RELATIVE to the smallest id of this video: with
    # `reset_ids_per_video: false` (ByteTrack / BoT-SORT keep the reference's class-level counter across videos) a later video's ids start far
    # above 1, but their RANGE inside one video is bounded by the tracks one video can create (at most one per detection slot and frame), which
    # is what `bound` holds.  An id beyond the bound is an error, never clamped into another id.
    bound = T * max(cap, maxd) + 2
    big = torch.iinfo(torch.int64).max
    id_min = torch.where(has, ids_all, torch.full_like(ids_all, big)).min() if T else torch.zeros((), dtype=torch.int64, device=dev)
    id_min = torch.where(id_min == big, torch.zeros_like(id_min), id_min)            # no tracked detection at all
    rel_all = ids_all - id_min + 1                                                    # >= 1 where `has`; slot 0 collects the untracked slots
    id_over = torch.where(has, rel_all, torch.zeros_like(rel_all)).max() if T else id_min
    present = torch.zeros(bound, dtype=torch.int64, device=dev)
    present.scatter_(0, torch.where(has, rel_all.clamp(0, bound - 1), torch.zeros_like(w)), has.to(torch.int64))
    rank = torch.cumsum(present.clamp_(max=1), 0) - 1
    rel_raw = torch.where(ids_raw > 0, ids_raw - id_min + 1, torch.zeros_like(ids_raw))
    dense = rank[rel_raw.clamp(0, bound - 1)].to(torch.int32)
    nt, n_ids, worst, over = (int(v) for v in torch.stack([off[-1], present.sum(), ocnt.min() if T else off[-1], id_over]).tolist())   # the ONE host fetch (four scalars)
    if over >= bound:
        raise RuntimeError(f"evaluate_device_log: track ids of this video span {over} values, more than the {bound - 2} tracks one video of {T} frames "
                           "can create -- the table does not hold one video")
    if worst < 0:
        raise RuntimeError("evaluate_device_log: a step of this video overflowed the tracker's capacity (negative row count in the table)")
    ltwh = torch.stack([box[:, 0], box[:, 1], box[:, 2] - box[:, 0], box[:, 3] - box[:, 1]], dim=-1)
    return {"ids": dense[:max(nt, 1)].contiguous(), "ltrb": box[:max(nt, 1)].contiguous(), "ltwh": ltwh[:max(nt, 1)].contiguous(), "off": off, "count": cnt,
            "n_boxes": nt, "n_ids": n_ids, "n_frames": T, "cap": maxd}


def evaluate_device_log(gt: dict, log, pipe, max_iou: float = 0.5) -> dict:
    """evaluate_sequence(gt, <the table>, device="gpu") WITHOUT fetching the table: the tracker side is read where HipVideoEngine left it (the
    HBM-resident DeviceStepLog of the video), re-arranged on the device (device_log_tracks) and handed to
    tlk_hota_sequence_dev_f64 / tlk_clear_sequence_dev_f64; only the ground truth (an input that lives on the host) is uploaded and only the
    two result vectors come back. Frames are the table's frames 0..T-1 = MOT frames 1..T. Same dict as evaluate_sequence."""
    import ctypes as C

    import torch
    from . import _lib
    tr = device_log_tracks(log, pipe)
    T, dev = tr["n_frames"], tr["off"].device
    if T == 0:
        return {"hota": np.zeros(len(hota.ALPHAS) * 7 + 2), "clear": clearmot.MOTAccumulator().counts()}
    g = _by_frame(gt)
    empty = (np.zeros(0, np.int64), np.zeros((0, 4)))
    gi, gb, goff = [], [], [0]
    for f in range(1, T + 1):
        i, b = g.get(f, empty)
        gi.append(np.asarray(i, dtype=np.int64)); gb.append(np.asarray(b, dtype=np.float64).reshape(-1, 4)); goff.append(goff[-1] + len(i))
    gi = np.concatenate(gi) if gi else np.zeros(0, np.int64)
    gb = np.concatenate(gb) if gb else np.zeros((0, 4))
    gu, gd = np.unique(gi, return_inverse=True) if len(gi) else (np.zeros(0, np.int64), np.zeros(0, np.int64))
    ng = int(goff[-1])
    d_gid = torch.from_numpy(np.ascontiguousarray(gd, dtype=np.int32) if ng else np.zeros(1, np.int32)).to(dev)
    d_gwh = torch.from_numpy(np.ascontiguousarray(gb) if ng else np.zeros((1, 4))).to(dev)
    d_gbr = torch.stack([d_gwh[:, 0], d_gwh[:, 1], d_gwh[:, 0] + d_gwh[:, 2], d_gwh[:, 1] + d_gwh[:, 3]], dim=-1).contiguous()
    d_goff = torch.from_numpy(np.asarray(goff, dtype=np.int64)).to(dev)
    n_match = int((np.diff(goff) * tr["cap"]).sum())                                  # upper bound known on the host: gt boxes x table capacity per frame
    L = _lib.lib()
    vp, ci, i64 = C.c_void_p, C.c_int, C.c_int64
    L.tlk_hota_sequence_dev_f64.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, i64, i64, i64, vp, vp, vp]
    L.tlk_clear_sequence_dev_f64.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, C.c_double, vp, vp]
    stats = np.zeros((7, len(hota.ALPHAS)))
    alphas = np.ascontiguousarray(hota.ALPHAS, dtype=np.float64)
    sp = _lib.current_stream_ptr()
    _lib.check(L.tlk_hota_sequence_dev_f64(d_gid.data_ptr(), d_gbr.data_ptr(), d_goff.data_ptr(), tr["ids"].data_ptr(), tr["ltrb"].data_ptr(), tr["off"].data_ptr(),
                                           T, len(gu), tr["n_ids"], ng, tr["n_boxes"], n_match, alphas.ctypes.data, stats.ctypes.data, sp))
    res = {k: stats[i].copy() for i, k in enumerate(("HOTA_TP", "HOTA_FN", "HOTA_FP", "LocA_sum", "AssA", "AssRe", "AssPr"))}
    counts = np.zeros(len(clearmot.SUM_FIELDS))
    _lib.check(L.tlk_clear_sequence_dev_f64(d_gid.data_ptr(), d_gwh.data_ptr(), d_goff.data_ptr(), tr["ids"].data_ptr(), tr["ltwh"].data_ptr(), tr["off"].data_ptr(),
                                            T, len(gu), tr["n_ids"], float(max_iou), counts.ctypes.data, sp))
    c = {k: (float(v) if k in ("sum_distance", "idtp", "idfp", "idfn") else int(v)) for k, v in zip(clearmot.SUM_FIELDS, counts)}
    return {"hota": hota.pack(res, frames=float(T)), "clear": c}


def combine(per_sequence: dict) -> dict:
    """{name: evaluate_sequence(...)} -> {'sequences': {name: metrics}, 'combined': metrics} (float summaries only)."""
    def metrics(h, c):
        m = dict(hota.finalize(h)["summary"])
        cm = clearmot.finalize(c)
        m.update({k.upper() if k in ("mota", "motp", "idf1", "idp", "idr") else k: cm[k] for k in
                  ("mota", "motp", "idf1", "idp", "idr", "precision", "recall", "num_switches", "num_false_positives", "num_misses",
                   "num_fragmentations", "mostly_tracked", "mostly_lost", "num_objects", "num_frames")})
        return {k: float(v) for k, v in m.items()}
    out = {"sequences": {n: metrics(r["hota"], r["clear"]) for n, r in per_sequence.items()}}
    if per_sequence:
        hsum = np.sum([r["hota"] for r in per_sequence.values()], axis=0)
        csum = clearmot.unpack(np.sum([clearmot.pack(r["clear"]) for r in per_sequence.values()], axis=0))
        out["combined"] = metrics(hsum, csum)
    return out


def evaluate_folders(gt_dir: str, pred_dir: str, device: str = "cpu") -> dict:
    res = {}
    for name in sorted(os.listdir(gt_dir)):
        if name.endswith(".txt") and os.path.exists(os.path.join(pred_dir, name)):
            res[name[:-4]] = evaluate_sequence(mot_io.load_mot(os.path.join(gt_dir, name)), mot_io.load_mot(os.path.join(pred_dir, name)), device=device)
    return combine(res)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--gpu"]
    if len(args) != 2:
        sys.exit(__doc__)
    print(json.dumps(evaluate_folders(args[0], args[1], device="gpu" if "--gpu" in sys.argv[1:] else "cpu"), indent=1))
