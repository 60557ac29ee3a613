"""Oracle (oracle/src/*.c) vs golden vectors produced by importing the reference
(tests/golden/make_golden.py): OC-SORT end-to-end, IoU family, KalmanBoxTracker replays."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

OCSORT_FILES = sorted(glob.glob(os.path.join(GOLDEN, "ocsort_*.npz")))
RTOL, ATOL = 0, 0            # r03: boxes and Kalman states BIT-exact (np.linalg.inv / np.dot in LAPACK / BLAS operation order, oracle/src/lapack_order.h)


def run_ocsort(tracker_cls, step_fn, g, on_frame=None):
    cfg = json.loads(str(g["config"]))
    trk = tracker_cls(**cfg["hyper"])
    do, oo = g["det_offsets"], g["out_offsets"]
    for f in range(len(do) - 1):
        dets = g["dets"][do[f]:do[f + 1]]
        exp = g["out"][oo[f]:oo[f + 1]]
        got = step_fn(trk, dets, cfg["min_confidence"])
        assert got.shape == exp.shape, f"frame {f}: rows {got.shape} vs {exp.shape}"
        # integer-valued columns exactly: track id, class, tracklab (detection) id
        np.testing.assert_array_equal(got[:, [4, 5, 7]], exp[:, [4, 5, 7]], err_msg=f"frame {f} ids")
        np.testing.assert_allclose(got[:, [0, 1, 2, 3, 6]], exp[:, [0, 1, 2, 3, 6]], rtol=RTOL, atol=ATOL,
                                   err_msg=f"frame {f} boxes")
        if on_frame:
            on_frame(f, trk)
    return trk


@pytest.mark.parametrize("path", OCSORT_FILES, ids=[os.path.basename(p)[7:-4] for p in OCSORT_FILES])
def test_ocsort_oracle_matches_reference(orc, path):
    g = np.load(path)

    def check_state(f, trk):
        if f"f{f}_kf_x" in g:
            x, P, ids = trk.tracks()
            np.testing.assert_array_equal(ids, g[f"f{f}_ids"])
            np.testing.assert_array_equal(x, g[f"f{f}_kf_x"])
            np.testing.assert_array_equal(P, g[f"f{f}_kf_P"])

    run_ocsort(orc.OCSort, orc.ocsort_wrapper_step, g, check_state)


def test_iou_family(orc):
    g = np.load(os.path.join(GOLDEN, "iou_family.npz"))
    for tag in "abcd":
        b1, b2 = g[f"{tag}_b1"], g[f"{tag}_b2"]
        for fn, var in (("iou_batch", "iou"), ("giou_batch", "giou"), ("diou_batch", "diou"),
                        ("ciou_batch", "ciou"), ("ct_dist", "ct_dist")):
            got = orc.iou_matrix(b1, b2, var)
            exp = g[f"{tag}_{fn}"]
            if var in ("iou", "giou", "diou"):
                np.testing.assert_array_equal(got, exp, err_msg=f"{tag} {fn}")       # same op order -> bit-exact
            else:
                np.testing.assert_allclose(got, exp, rtol=1e-13, atol=1e-15, err_msg=f"{tag} {fn}")


def test_kalman_box_tracker_replays(orc):
    g = np.load(os.path.join(GOLDEN, "kf7_cases.npz"))
    for c in range(int(g["n_cases"])):
        obs, pat = g[f"c{c}_obs"], g[f"c{c}_pattern"]
        k = orc.KalmanBoxTracker(obs[0], 1.0, delta_t=3)
        for t, seen in enumerate(pat, start=1):
            pos = k.predict()
            np.testing.assert_array_equal(pos, g[f"c{c}_pred"][t - 1])
            k.update(obs[t] if seen else None, 1.0)
            x, P, v = k.state()
            np.testing.assert_array_equal(x, g[f"c{c}_x"][t - 1], err_msg=f"case {c} t {t}")
            np.testing.assert_array_equal(P, g[f"c{c}_P"][t - 1], err_msg=f"case {c} t {t}")
            np.testing.assert_array_equal(v, g[f"c{c}_vel"][t - 1])


def test_cosine_gallery_oracle_matches_reference(orc):
    g = np.load(os.path.join(GOLDEN, "cosine_gallery.npz"))
    for c in range(int(g["n_cases"])):
        out = orc.cosine_gallery_min(g[f"c{c}_gallery"], g[f"c{c}_offsets"], g[f"c{c}_dets"])
        np.testing.assert_allclose(out, g[f"c{c}_cost"], rtol=0, atol=3e-6)      # fp32 dot products, different summation order
