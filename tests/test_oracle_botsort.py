"""CPU: the BoT-SORT oracle (oracle/src/botsort.c) against runs of the reference's own BoTSORT.update
(tests/golden/make_golden.py gen_botsort: plugins/track/bot_sort imported, cmc_method "none", `lap` shimmed by its documented
embedding, the ReID forward replaced by synthetic embeddings)."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

RUNS = sorted(os.path.basename(p)[8:-4] for p in glob.glob(os.path.join(GOLDEN, "botsort_*.npz")))


def replay(name, make_tracker, check_lists=None, box_tol=(1e-11, 1e-10)):
    """Feed a golden run frame by frame to `make_tracker(D, hyper)`.update(dets, emb) and assert the output rows."""
    from tracklab_amd.synth import SyntheticStream
    g = np.load(os.path.join(GOLDEN, f"botsort_{name}.npz"))
    hp, D = json.loads(str(g["config"])), int(g["dim"])
    trk = make_tracker(D, hp)
    stream = SyntheticStream(int(g["seed"]), int(g["n_objects"]), int(g["n_frames"]), parts=1, dim=D, with_embeddings=True,
                             **json.loads(str(g["stream_kwargs"])))
    do, oo = g["det_offsets"], g["out_offsets"]
    min_conf = float(g["min_confidence"])
    for f, fr in enumerate(stream):
        dets = g["dets"][do[f]:do[f + 1]]
        emb = fr["embeddings"][:, 0, :].astype(np.float32)
        if f % 41 == 13:
            emb = emb[:0]
        assert len(emb) == len(dets)
        if len(dets) == 0:
            continue                                           # wrapper: process() returns [] (bot_sort_api.py:59-60)
        keep = dets[:, 4] > min_conf                           # bot_sort_api.py:62
        out = trk.update(dets[keep], emb[keep])
        exp = g["rows"][oo[f]:oo[f + 1]]
        assert out.shape == exp.shape, f"{name} frame {f}"
        np.testing.assert_array_equal(out[:, 4:], exp[:, 4:], err_msg=f"{name} frame {f}")        # track id, class, score, tracklab id
        np.testing.assert_allclose(out[:, :4], exp[:, :4], rtol=box_tol[0], atol=box_tol[1], err_msg=f"{name} frame {f}")
        if check_lists is not None and f"f{f}_trk_ids" in g.files:
            check_lists(trk, g, f)


def check_lists_exact(trk, g, f):
    for which, ln in ((0, "trk"), (1, "lost")):
        ids, mean, cov, st, feat = trk.tracks(which)
        np.testing.assert_array_equal(ids, g[f"f{f}_{ln}_ids"])          # list membership AND order
        np.testing.assert_array_equal(st, g[f"f{f}_{ln}_state"])         # state, is_activated, frame_id, start_frame, tracklet_len
        np.testing.assert_allclose(mean, g[f"f{f}_{ln}_mean"], rtol=0, atol=0)
        np.testing.assert_allclose(cov, g[f"f{f}_{ln}_cov"], rtol=0, atol=0)
        np.testing.assert_allclose(feat, g[f"f{f}_{ln}_feat"], rtol=0, atol=5e-7)    # float32 EMA + renorm, summation order of the norm


@pytest.mark.parametrize("name", RUNS)
def test_botsort_oracle_matches_reference(orc, name):
    replay(name, lambda D, hp: orc.BoTSORT(D, **hp), check_lists_exact)


def test_botsort_oracle_with_camera_motion_warps_matches_reference(orc):
    """multi_gmc on non-identity warps (bot_sort.py:93-109): BoTSORT.update run by the reference with GMC.apply patched to return a
    synthetic (2,3) warp per frame (tests/golden/make_golden.py gen_botsort_gmc); the oracle gets the same warps."""
    g = np.load(os.path.join(GOLDEN, "gmc_botsort.npz"))
    trk = orc.BoTSORT(int(g["dim"]), **json.loads(str(g["config"])))
    do, oo = g["det_offsets"], g["out_offsets"]
    for f in range(len(do) - 1):
        out = trk.update(g["dets"][do[f]:do[f + 1]], g["embeddings"][do[f]:do[f + 1]], warp=g["warps"][f])
        exp = g["rows"][oo[f]:oo[f + 1]]
        assert out.shape == exp.shape, f
        np.testing.assert_array_equal(out[:, 4:], exp[:, 4:], err_msg=f"frame {f}")
        np.testing.assert_allclose(out[:, :4], exp[:, :4], rtol=0, atol=0, err_msg=f"frame {f}")
        if f"f{f}_trk_ids" in g.files:
            for which, ln in ((0, "trk"), (1, "lost")):
                ids, mean, cov = trk.tracks(which)[:3]
                np.testing.assert_array_equal(ids, g[f"f{f}_{ln}_ids"])
                np.testing.assert_allclose(mean, g[f"f{f}_{ln}_mean"], rtol=0, atol=0)
                np.testing.assert_allclose(cov, g[f"f{f}_{ln}_cov"], rtol=0, atol=0)
