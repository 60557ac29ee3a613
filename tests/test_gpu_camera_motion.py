"""-m gpu: applying an externally estimated camera-motion warp to the tracker state on the device (the application half):
tlk_ssort_camera_update and tlk_deepocsort_affine_correction against runs of the reference with its cv2 estimator patched to return
synthetic warps (tests/golden/camera_ssort.npz, cmc_deepocsort.npz) and, state for state, against the oracle."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_plain_strongsort_camera_update_on_device(orc):
    from tracklab_amd._lib import SsortBank
    g = np.load(os.path.join(GOLDEN, "camera_ssort.npz"))
    hp, D = json.loads(str(g["config"])), int(g["dim"])
    gpu, cpu = SsortBank(D, **hp), orc.PlainStrongSORT(D, **hp)
    do, oo = g["det_offsets"], g["out_offsets"]
    for f in range(len(do) - 1):
        if f > 0:                                              # strong_sort_api.py:62-65
            gpu.camera_update(g["warps"][f]); cpu.camera_update(g["warps"][f])
        d, e = g["dets"][do[f]:do[f + 1]], g["embeddings"][do[f]:do[f + 1]]
        r = gpu.update(d, e)
        out = np.column_stack([r["ltrb"], r["track_id"], r["class_id"], r["conf"], r["det_id"]]).astype(np.float64).reshape(-1, 8)
        np.testing.assert_array_equal(out, g["rows"][oo[f]:oo[f + 1]], err_msg=f"frame {f}")          # the reference's rows
        np.testing.assert_array_equal(out, cpu.update(d, e))
        gi, gm = gpu.tracks()[:2]
        ci, cm = cpu.tracks()[:2]
        np.testing.assert_array_equal(gi, ci)
        np.testing.assert_array_equal(gm, cm)                  # same fp64 / fp32 operation order: bit-identical means
        if f"f{f}_track_ids" in g.files:
            np.testing.assert_allclose(gm, g[f"f{f}_mean"], rtol=1e-11, atol=1e-10)


def test_deepocsort_affine_correction_on_device(orc):
    from tracklab_amd._lib import DeepOCSortBank
    g = np.load(os.path.join(GOLDEN, "cmc_deepocsort.npz"))
    hp, D = json.loads(str(g["config"])), int(g["dim"])
    gpu, cpu = DeepOCSortBank(D, **hp), orc.DeepOCSort(D, **hp)
    do, oo = g["det_offsets"], g["out_offsets"]
    for f in range(len(do) - 1):
        d, e = g["dets"][do[f]:do[f + 1]], g["embeddings"][do[f]:do[f + 1]]
        gpu.affine_correction(g["warps"][f])                   # ocsort.py:425-428: every frame, ahead of predict
        out = gpu.update(d, e)
        exp = g["rows"][oo[f]:oo[f + 1]]
        assert out.shape == exp.shape, f
        np.testing.assert_array_equal(out[:, 4:], exp[:, 4:], err_msg=f"frame {f}")
        np.testing.assert_allclose(out[:, :4], exp[:, :4], rtol=1e-12, atol=1e-10)
        np.testing.assert_array_equal(out, cpu.update(d, e, warp=g["warps"][f]))
        gi, gx, gP, ge, gs, gv, gl = gpu.tracks()
        ci, cx, cP, ce, cs, cv, cl = cpu.tracks()
        np.testing.assert_array_equal(gi, ci); np.testing.assert_array_equal(gs, cs)
        np.testing.assert_array_equal(gl, cl); np.testing.assert_array_equal(gv, cv)
        np.testing.assert_array_equal(gx, cx); np.testing.assert_array_equal(gP, cP)
        if f"f{f}_ids" in g.files:
            np.testing.assert_allclose(gx, g[f"f{f}_x"], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(gl, g[f"f{f}_last"], rtol=1e-12, atol=1e-10)
