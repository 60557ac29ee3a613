"""CPU: the ByteTrack oracle (oracle/src/bytetrack.c) against runs of the reference's own BYTETracker.update
(tests/golden/make_golden.py gen_bytetrack: plugins/track/byte_track imported with `lap` shimmed by its documented embedding)."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

RUNS = sorted(os.path.basename(p)[10:-4] for p in glob.glob(os.path.join(GOLDEN, "bytetrack_*.npz")))


def replay(name, make_tracker, check_lists=None):
    g = np.load(os.path.join(GOLDEN, f"bytetrack_{name}.npz"))
    trk = make_tracker(json.loads(str(g["config"])))
    do, oo = g["det_offsets"], g["out_offsets"]
    min_conf = float(g["min_confidence"])
    for f in range(len(do) - 1):
        dets = g["dets"][do[f]:do[f + 1]]
        if len(dets) == 0:
            continue                                           # wrapper: process() returns [] (byte_track_api.py:55-56)
        out = trk.update(dets[dets[:, 4] > min_conf])          # byte_track_api.py:58
        exp = g["rows"][oo[f]:oo[f + 1]]
        assert out.shape == exp.shape, f"{name} frame {f}"
        np.testing.assert_array_equal(out[:, 4:], exp[:, 4:], err_msg=f"{name} frame {f}")        # track id, class, score, tracklab id
        np.testing.assert_allclose(out[:, :4], exp[:, :4], rtol=0, atol=0, err_msg=f"{name} frame {f}")
        if check_lists is not None and f"f{f}_trk_ids" in g.files:
            check_lists(trk, g, f)


def check_lists_exact(trk, g, f):
    for which, ln in ((0, "trk"), (1, "lost")):
        ids, mean, cov, st = trk.tracks(which)
        np.testing.assert_array_equal(ids, g[f"f{f}_{ln}_ids"])          # list membership AND order
        np.testing.assert_array_equal(st, g[f"f{f}_{ln}_state"])         # state, is_activated, frame_id, start_frame, tracklet_len
        np.testing.assert_allclose(mean, g[f"f{f}_{ln}_mean"], rtol=0, atol=0)
        np.testing.assert_allclose(cov, g[f"f{f}_{ln}_cov"], rtol=0, atol=0)


@pytest.mark.parametrize("name", RUNS)
def test_bytetrack_oracle_matches_reference(orc, name):
    replay(name, lambda hp: orc.ByteTrack(**hp), check_lists_exact)
