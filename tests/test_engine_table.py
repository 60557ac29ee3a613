"""CPU: the engine's columnar detection table (tracklab_amd/engine.py) -- the vectorised per-step append against a per-row restatement of what
the reference's merge_dataframes chain would leave behind (tracklab/engine/engine.py:18-41: detector rows indexed by detection id, tracker
columns joined by index)."""
import numpy as np

from tracklab_amd.engine import DetectionTable


def test_append_step_matches_per_row_merge():
    rng = np.random.default_rng(3)
    maxd, F = 16, 5
    table = DetectionTable(capacity=8)               # forces growth
    expect = {}
    first, id_base = 0, 0
    for step in range(7):
        n = F if step < 6 else 3                     # partial last step: frames >= n are padding
        dcnt = rng.integers(0, maxd + 1, F).astype(np.int32)
        ltwh = rng.uniform(0, 500, (F, maxd, 4)).astype(np.float32)
        tf, tdet, tid, tl, tconf = [], [], [], [], []
        for f in range(F):
            for i in rng.permutation(int(dcnt[f]))[:int(rng.integers(0, dcnt[f] + 1))]:       # the tracker reports a subset, in its own order
                tf.append(f); tdet.append(id_base + f * maxd + int(i)); tid.append(float(rng.integers(1, 99)))
                tl.append(rng.uniform(0, 500, 4)); tconf.append(rng.uniform())
        tf.append(0); tdet.append(id_base + maxd - 1 if dcnt[0] < maxd else id_base - 1); tid.append(777.0)        # a row whose detection id is
        tl.append(np.zeros(4)); tconf.append(0.0)                                                                   # not in the frame: ignored
        trk = (np.array(tf, dtype=np.int64), np.array(tdet, dtype=np.int64), np.array(tid), np.array(tl).reshape(-1, 4), np.array(tconf))
        table.append_step(first, n, id_base, maxd, ltwh, dcnt, trk)
        for f in range(n):
            for i in range(int(dcnt[f])):
                expect[id_base + f * maxd + i] = [first + f, ltwh[f, i], np.nan, None, np.nan]
        for k in range(len(tf) - 1):
            if tf[k] < n:
                e = expect[tdet[k]]
                e[2], e[3], e[4] = tid[k], tl[k], tconf[k]
        first += n
        id_base += F * maxd
    df = table.to_dataframe(video_id=9)
    assert list(df.index) == sorted(expect) and len(df) == len(expect)
    assert (df.video_id == 9).all() and (df.category_id == 1).all() and (df.bbox_conf == 1.0).all()
    for did, (img, box, tid, tl, tconf) in expect.items():
        row = df.loc[did]
        assert row.image_id == img and np.array_equal(row.bbox_ltwh, box)
        if tl is None:
            assert np.isnan(row.track_id) and np.isnan(row.track_bbox_ltwh).all() and np.isnan(row.track_bbox_conf)
        else:
            assert row.track_id == tid and np.array_equal(row.track_bbox_ltwh, tl) and row.track_bbox_conf == tconf


def test_capacity_errors_are_loud():
    import pytest
    t = DetectionTable()
    with pytest.raises(RuntimeError):
        t.append_step(0, 1, 0, 4, np.zeros((1, 4, 4), np.float32), np.array([-4], np.int32),
                      (np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0), np.zeros((0, 4)), np.zeros(0)))


def test_tracking_engine_decodes_ahead_in_order_with_worker_threads():
    """HipTrackingEngine._decode_ahead: frames come back in video order, several decodes are in flight at once, and no more than the window runs ahead."""
    import threading
    import time
    from types import SimpleNamespace
    from tracklab_amd.engine import HipTrackingEngine
    eng = HipTrackingEngine.__new__(HipTrackingEngine)                    # host logic only: no pipeline, no device
    eng.video_engine = SimpleNamespace(F=4)
    lock, state = threading.Lock(), {"live": 0, "peak": 0, "started": 0}

    def loader(p):
        with lock:
            state["live"] += 1; state["started"] += 1
            state["peak"] = max(state["peak"], state["live"])
        time.sleep(0.01 * (1 + int(p) % 3))                               # uneven decode times: completion order != submission order
        with lock:
            state["live"] -= 1
        return int(p)
    eng.image_loader = loader
    paths = [str(k) for k in range(40)]
    for workers in (0, 4):
        eng.num_workers = workers
        state.update(live=0, peak=0, started=0)
        got, ahead = [], 0
        for k, v in enumerate(eng._decode_ahead(paths)):
            got.append(v)
            ahead = max(ahead, state["started"] - (k + 1))
        assert got == list(range(40))
        if workers:
            assert 2 <= state["peak"] <= workers and ahead <= max(2 * 4, workers)
        else:
            assert state["peak"] == 1 and ahead == 0
    eng.num_workers = 4
    assert list(eng._decode_ahead([])) == []
