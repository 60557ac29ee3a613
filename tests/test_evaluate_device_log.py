"""The re-arrangement of the engine's per-video table into the evaluators' layout (tracklab_amd.evaluate.device_log_tracks) is torch ops without
data-dependent shapes, so it runs on CPU tensors too: checked here against the HOST path it replaces -- engine.DetectionTable.append_step fed with
pipeline.track_columns, i.e. the reference's merge of a module's rows into the detections by index -- for every row format the banks emit,
including rows that name a detection of an earlier frame of the step (a coasting plain-StrongSORT track: the last row wins), of an earlier
step or beyond the frame's count (dropped). The evaluators themselves (tlk_*_sequence_dev_f64) need the GPU: tests/test_gpu_engine.py."""
import numpy as np
import pytest
import torch

from tracklab_amd import _lib, evaluate
from tracklab_amd import gpu_pipeline as gp
from tracklab_amd.engine import DetectionTable, DeviceStepLog

FORMATS = {"oc_sort": None, "bpbreid": _lib.BPBSS_ROW, "strong_sort": _lib.SSORT_ROW, "byte_track": _lib.BYTETRACK_ROW, "bot_sort": _lib.BOTSORT_ROW,
           "deep_oc_sort": _lib.DEEPOCSORT_ROW}


class _Pipe:
    def __init__(self, rd, maxd):
        self.row_dtype, self.maxd = rd, maxd
    rows_array_np = gp.DetTrackPipeline.rows_array_np

    def track_columns(self, rows, ocnt):
        cls = gp.DetTrackPipeline if (self.row_dtype is None or self.row_dtype is _lib.BYTETRACK_ROW) else gp.DetReidTrackPipeline
        return cls.track_columns(self, rows, ocnt)


@pytest.mark.parametrize("id_offset", [0, 5_000_000])
@pytest.mark.parametrize("name", list(FORMATS))
def test_device_log_tracks_equals_the_detection_table(name, id_offset):
    """id_offset: a LATER video of a run with `reset_ids_per_video: false` -- ByteTrack / BoT-SORT keep counting, so the ids of a short video
    start far above the number of tracks the video itself can hold (they used to collapse into one dense id)"""
    rd = FORMATS[name]
    F, cap, maxd = 3, 10, 8
    pipe = _Pipe(rd, maxd)
    rng = np.random.default_rng(len(name))
    log, table = DeviceStepLog(chunk=2), DetectionTable()
    t0 = 0
    for k, n in enumerate([3, 3, 3, 2]):                       # a partial last step; chunks of two steps
        id_base = 1000 + t0 * maxd                            # (frames_done * max_dets of the pipeline, plus an offset nobody should assume is 0)
        rows = np.zeros((F, cap, 8)) if rd is None else np.zeros((F, cap), dtype=rd)
        ocnt = rng.integers(0, cap + 1, F).astype(np.int32)
        dcnt = rng.integers(maxd // 2, maxd + 1, F).astype(np.int32)
        if k == 1:
            ocnt[1] = 0                                        # a frame without rows
        for f in range(F):
            for i in range(cap):
                tid = float(id_offset + rng.integers(1, 40))
                b = rng.uniform(0, 100, 4); b[2:] += b[:2]
                u = rng.random()
                tf = f if u < 0.7 else (max(f - 1, 0) if u < 0.85 else (f - 4 if u < 0.93 else f + 1))      # this frame / the previous one / an earlier step / a later frame
                det = id_base + tf * maxd + int(rng.integers(0, maxd))
                if rd is None:
                    rows[f, i, :4] = b; rows[f, i, 4] = tid; rows[f, i, 7] = det; rows[f, i, 6] = 0.9
                else:
                    rows[f, i]["track_id"] = tid; rows[f, i]["det_id"] = det
                    if "kf_ltwh" in rd.names:
                        rows[f, i]["kf_ltwh"] = [b[0], b[1], b[2] - b[0], b[3] - b[1]]
                    else:
                        rows[f, i]["ltrb"] = b
        ltwh = rng.uniform(0, 50, (F, maxd, 4)).astype(np.float32)
        raw = rows if rd is None else rows.view(np.uint8).reshape(F, cap, -1)
        log.sink(t0, n, id_base)({"rows": torch.from_numpy(raw.copy()), "ocnt": torch.from_numpy(ocnt), "ltwh": torch.from_numpy(ltwh), "dcnt": torch.from_numpy(dcnt)})
        table.append_step(t0, n, id_base, maxd, ltwh, dcnt, pipe.track_columns(raw, ocnt.astype(np.int64)))
        t0 += n
    df = table.to_dataframe(0)
    tracked = df[df.track_id.notna()]
    exp_ids = tracked.track_id.to_numpy().astype(np.int64)
    exp_ltwh = np.stack(tracked.track_bbox_ltwh.to_list()) if len(tracked) else np.zeros((0, 4))
    exp_off = np.concatenate([[0], np.cumsum(np.bincount(tracked.image_id.to_numpy().astype(int), minlength=t0))])
    tr = evaluate.device_log_tracks(log, pipe)
    u, inv = np.unique(exp_ids, return_inverse=True)
    assert len(exp_ids) > 12
    assert (tr["n_boxes"], tr["n_ids"], tr["n_frames"], tr["cap"]) == (len(exp_ids), len(u), t0, maxd)
    np.testing.assert_array_equal(tr["ids"].numpy()[:len(inv)], inv)                    # dense ids in the sorted order of the track ids
    np.testing.assert_allclose(tr["ltwh"].numpy()[:len(inv)], exp_ltwh, rtol=0, atol=1e-12)
    np.testing.assert_allclose(tr["ltrb"].numpy()[:len(inv)], np.column_stack([exp_ltwh[:, 0], exp_ltwh[:, 1], exp_ltwh[:, 0] + exp_ltwh[:, 2], exp_ltwh[:, 1] + exp_ltwh[:, 3]]),
                               rtol=0, atol=1e-12)
    np.testing.assert_array_equal(tr["off"].numpy(), exp_off)


def test_device_log_tracks_refuses_ids_that_span_more_than_one_video_can_create():
    """never clamp an id into another one: ids 1 and 10 000 in a 2-frame, 4-slot table cannot both come from this video"""
    pipe = _Pipe(None, 4)
    rows = torch.zeros((2, 4, 8), dtype=torch.float64)
    rows[0, 0, 4], rows[0, 0, 7] = 1.0, 0.0
    rows[1, 0, 4], rows[1, 0, 7] = 10_000.0, 4.0
    log = DeviceStepLog(chunk=2)
    log.sink(0, 2, 0)({"rows": rows, "ocnt": torch.tensor([1, 1], dtype=torch.int32), "dcnt": torch.tensor([4, 4], dtype=torch.int32)})
    with pytest.raises(RuntimeError, match="span"):
        evaluate.device_log_tracks(log, pipe)


def test_device_log_tracks_refuses_an_overflowed_step():
    pipe = _Pipe(None, 4)
    log = DeviceStepLog(chunk=2)
    log.sink(0, 2, 0)({"rows": torch.zeros((2, 4, 8), dtype=torch.float64), "ocnt": torch.tensor([1, -3], dtype=torch.int32), "dcnt": torch.tensor([4, 4], dtype=torch.int32)})
    with pytest.raises(RuntimeError, match="capacity"):
        evaluate.device_log_tracks(log, pipe)


def test_an_empty_video_evaluates_to_zero_counts():
    pipe = _Pipe(None, 4)
    res = evaluate.evaluate_device_log({"frame": np.zeros(0, int), "track_id": np.zeros(0, int), "ltwh": np.zeros((0, 4))}, DeviceStepLog(), pipe)
    assert not res["hota"].any() and res["clear"]["num_frames"] == 0
