"""The re-arrangement of the engine's per-video table into the evaluators' layout (tracklab_amd.evaluate.device_log_tracks) is torch ops without
data-dependent shapes, so it runs on CPU tensors too: checked here against a plain loop for every row format the banks emit. The evaluators
themselves (tlk_*_sequence_dev_f64) need the GPU: tests/test_gpu_engine.py."""
import numpy as np
import pytest
import torch

from tracklab_amd import _lib, evaluate
from tracklab_amd.engine import DeviceStepLog

FORMATS = {"oc_sort": None, "bpbreid": _lib.BPBSS_ROW, "strong_sort": _lib.SSORT_ROW, "byte_track": _lib.BYTETRACK_ROW, "bot_sort": _lib.BOTSORT_ROW,
           "deep_oc_sort": _lib.DEEPOCSORT_ROW}


@pytest.mark.parametrize("name", list(FORMATS))
def test_device_log_tracks_equals_a_plain_loop(name):
    rd = FORMATS[name]
    pipe = type("P", (), {"row_dtype": rd, "maxd": 8})()
    rng = np.random.default_rng(len(name))
    log = DeviceStepLog(chunk=2)
    F, cap = 3, 8
    ids, boxes, off = [], [], [0]
    for k, n in enumerate([3, 3, 3, 2]):                       # a partial last step; chunks of two steps
        rows = np.zeros((F, cap, 8)) if rd is None else np.zeros((F, cap), dtype=rd)
        ocnt = rng.integers(0, cap + 1, F).astype(np.int32)
        if k == 1:
            ocnt[1] = 0                                        # a frame without rows
        for f in range(F):
            for i in range(cap):
                tid = float(rng.integers(1, 40))
                b = rng.uniform(0, 100, 4); b[2:] += b[:2]
                if rd is None:
                    rows[f, i, :4] = b; rows[f, i, 4] = tid
                else:
                    rows[f, i]["track_id"] = tid
                    if "kf_ltwh" in rd.names:
                        rows[f, i]["kf_ltwh"] = [b[0], b[1], b[2] - b[0], b[3] - b[1]]
                    else:
                        rows[f, i]["ltrb"] = b
                if f < n and i < ocnt[f]:
                    ids.append(int(tid)); boxes.append(b.copy())
            if f < n:
                off.append(len(ids))
        t = torch.from_numpy(rows) if rd is None else torch.from_numpy(rows.view(np.uint8).reshape(F, cap, -1))
        log.sink(k * F, n, 0)({"rows": t, "ocnt": torch.from_numpy(ocnt)})
    tr = evaluate.device_log_tracks(log, pipe)
    u, inv = np.unique(np.array(ids), return_inverse=True)
    assert (tr["n_boxes"], tr["n_ids"], tr["n_frames"], tr["cap"]) == (len(ids), len(u), 11, cap)
    np.testing.assert_array_equal(tr["ids"].numpy()[:len(inv)], inv)                    # dense ids in the sorted order of the track ids
    np.testing.assert_allclose(tr["ltrb"].numpy()[:len(inv)], np.array(boxes), rtol=0, atol=1e-12)
    b = np.array(boxes)
    np.testing.assert_allclose(tr["ltwh"].numpy()[:len(inv)], np.column_stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]]), rtol=0, atol=1e-12)
    np.testing.assert_array_equal(tr["off"].numpy(), off)


def test_device_log_tracks_refuses_an_overflowed_step():
    pipe = type("P", (), {"row_dtype": None, "maxd": 4})()
    log = DeviceStepLog(chunk=2)
    log.sink(0, 2, 0)({"rows": torch.zeros((2, 4, 8), dtype=torch.float64), "ocnt": torch.tensor([1, -3], dtype=torch.int32)})
    with pytest.raises(RuntimeError, match="capacity"):
        evaluate.device_log_tracks(log, pipe)
