"""CPU: the Deep-OC-SORT oracle (oracle/src/deepocsort.c) against runs of the reference's own OCSort.update
(tests/golden/make_golden.py gen_deepocsort: plugins/track/deep_oc_sort imported, cmc_off, scipy LSA, the ReID forward replaced by
synthetic float32 torch embeddings)."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

RUNS = sorted(os.path.basename(p)[11:-4] for p in glob.glob(os.path.join(GOLDEN, "deepocsort_*.npz")))


def replay(name, make_tracker, check_state=None):
    g = np.load(os.path.join(GOLDEN, f"deepocsort_{name}.npz"))
    hp, D = json.loads(str(g["config"])), int(g["dim"])
    trk = make_tracker(D, hp)
    do, oo = g["det_offsets"], g["out_offsets"]
    min_conf = float(g["min_confidence"])
    for f in range(len(do) - 1):
        dets, emb = g["dets"][do[f]:do[f + 1]], g["embeddings"][do[f]:do[f + 1]]
        if len(dets) == 0:
            continue                                           # wrapper: process() returns [] (deep_oc_sort_api.py:59-60)
        keep = dets[:, 4] > min_conf                           # deep_oc_sort_api.py:62
        out = trk.update(dets[keep], emb[keep])
        exp = g["rows"][oo[f]:oo[f + 1]]
        assert out.shape == exp.shape, f"{name} frame {f}"
        np.testing.assert_array_equal(out, exp, err_msg=f"{name} frame {f}")      # boxes are copies of detections: exact
        if check_state is not None and f"f{f}_ids" in g.files:
            check_state(trk, g, f)


def check_state(trk, g, f, x_tol=1e-12, emb_tol=1e-6):
    # r03: np.linalg.inv / np.dot are reproduced in library operation order (lapack_order.h), which leaves ONE source of last-bit differences:
    # new_kf_process_noise / new_kf_measurement_noise (ocsort.py:82-93) square with `** 2`, i.e. libm pow(x, 2), which rounds differently from
    # x * x in 0.08 % of the cases on this glibc; oracle and kernels use x * x (a GPU cannot reproduce one libm's mis-roundings).
    ids, x, P, emb, st, vel, last = trk.tracks()
    np.testing.assert_array_equal(ids, g[f"f{f}_ids"])
    np.testing.assert_array_equal(st, g[f"f{f}_state"])        # time_since_update, hits, hit_streak, age, frozen, kf.observed
    np.testing.assert_array_equal(last, g[f"f{f}_last"])
    np.testing.assert_allclose(vel, g[f"f{f}_vel"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(x, g[f"f{f}_x"], rtol=x_tol, atol=x_tol)
    np.testing.assert_allclose(P, g[f"f{f}_P"], rtol=1e-12, atol=1e-12)
    assert not g[f"f{f}_emb_f64"].any()                        # the reference's embeddings stay float32 tensors
    np.testing.assert_allclose(emb, g[f"f{f}_emb"], rtol=0, atol=emb_tol)         # float32 EMA + renorm, summation order of the norm


@pytest.mark.parametrize("name", RUNS)
def test_deepocsort_oracle_matches_reference(orc, name):
    replay(name, lambda D, hp: orc.DeepOCSort(D, **hp), check_state)


def test_deepocsort_oracle_with_camera_motion_warps_matches_reference(orc):
    """apply_affine_correction on non-identity warps: OCSort.update run by the reference with cmc_off False and
    CMCComputer.compute_affine patched to return a synthetic (2,3) warp per frame (gen_deepocsort_cmc); the oracle gets the same warps."""
    g = np.load(os.path.join(GOLDEN, "cmc_deepocsort.npz"))
    trk = orc.DeepOCSort(int(g["dim"]), **json.loads(str(g["config"])))
    do, oo = g["det_offsets"], g["out_offsets"]
    for f in range(len(do) - 1):
        out = trk.update(g["dets"][do[f]:do[f + 1]], g["embeddings"][do[f]:do[f + 1]], warp=g["warps"][f])
        exp = g["rows"][oo[f]:oo[f + 1]]
        assert out.shape == exp.shape, f
        np.testing.assert_array_equal(out[:, 4:], exp[:, 4:], err_msg=f"frame {f}")
        np.testing.assert_allclose(out[:, :4], exp[:, :4], rtol=1e-12, atol=1e-10, err_msg=f"frame {f}")
        if f"f{f}_ids" in g.files:
            ids, x, P, emb, st, vel, last = trk.tracks()
            np.testing.assert_array_equal(ids, g[f"f{f}_ids"])
            np.testing.assert_array_equal(st, g[f"f{f}_state"])
            np.testing.assert_allclose(last, g[f"f{f}_last"], rtol=1e-12, atol=1e-10)
            np.testing.assert_allclose(vel, g[f"f{f}_vel"], rtol=0, atol=1e-12)
            np.testing.assert_allclose(x, g[f"f{f}_x"], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(P, g[f"f{f}_P"], rtol=1e-8, atol=1e-8)
