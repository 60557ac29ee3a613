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


def test_botsort_multi_gmc_on_device(orc):
    """BoT-SORT's STrack.multi_gmc inside the frame kernel (between multi_predict and the association, bot_sort.py:341-343): the reference's
    own run with GMC.apply patched to synthetic (2,3) warps (tests/golden/gmc_botsort.npz) replayed through tlk_botsort_update_gmc; rows and
    list order exact, Kalman state bit-identical to the oracle's; and the batched device entry point with per-frame warps gives the same rows."""
    import torch
    from tracklab_amd._lib import BOTSORT_ROW, BoTSORTBank
    g = np.load(os.path.join(GOLDEN, "gmc_botsort.npz"))
    hp, D = json.loads(str(g["config"])), int(g["dim"])
    gpu, cpu = BoTSORTBank(D, **hp), orc.BoTSORT(D, **hp)
    do, oo = g["det_offsets"], g["out_offsets"]
    nf = len(do) - 1
    cols = lambda r: np.column_stack([r["ltrb"], r["track_id"], r["cls"], r["score"], r["det_id"]]).astype(np.float64).reshape(-1, 8)
    all_rows = []
    for f in range(nf):
        d, e, w = g["dets"][do[f]:do[f + 1]], g["embeddings"][do[f]:do[f + 1]], g["warps"][f]
        out = cols(gpu.update(d, e, warp=w))
        exp = g["rows"][oo[f]:oo[f + 1]]
        assert out.shape == exp.shape, f
        np.testing.assert_array_equal(out[:, 4:], exp[:, 4:], err_msg=f"frame {f}")
        np.testing.assert_allclose(out[:, :4], exp[:, :4], rtol=1e-11, atol=1e-9, err_msg=f"frame {f}")
        np.testing.assert_array_equal(out, cpu.update(d, e, warp=w))
        all_rows.append(out)
        for which in (0, 1):
            gi, gm, gc = gpu.tracks(which)[:3]
            ci, cm, cc = cpu.tracks(which)[:3]
            np.testing.assert_array_equal(gi, ci); np.testing.assert_array_equal(gm, cm); np.testing.assert_array_equal(gc, cc)
        if f"f{f}_trk_ids" in g.files:
            ids, mean, cov = gpu.tracks(0)[:3]
            np.testing.assert_array_equal(ids, g[f"f{f}_trk_ids"])
            np.testing.assert_allclose(mean, g[f"f{f}_trk_mean"], rtol=1e-11, atol=1e-10)
            np.testing.assert_allclose(cov, g[f"f{f}_trk_cov"], rtol=1e-9, atol=1e-9)
    # batched: (1 stream, nf frames) with a warp per frame
    MAXD = 128
    dets = np.zeros((1, nf, MAXD, 7)); feats = np.zeros((1, nf, MAXD, D), np.float32); cnt = np.zeros((1, nf), np.int32)
    for f in range(nf):
        n = do[f + 1] - do[f]
        dets[0, f, :n], feats[0, f, :n], cnt[0, f] = g["dets"][do[f]:do[f + 1]], g["embeddings"][do[f]:do[f + 1]], n
    bank = BoTSORTBank(D, **hp, max_dets=MAXD)
    d_d, d_f, d_c = torch.from_numpy(dets).cuda(), torch.from_numpy(feats).cuda(), torch.from_numpy(cnt).cuda()
    d_w = torch.from_numpy(np.ascontiguousarray(g["warps"][:nf], dtype=np.float64).reshape(1, nf, 6)).cuda()
    cap = 256
    rows = torch.zeros((1, nf, cap, BOTSORT_ROW.itemsize), dtype=torch.uint8, device="cuda"); oc = torch.zeros((1, nf), dtype=torch.int32, device="cuda")
    bank.update_dev(d_d.data_ptr(), d_f.data_ptr(), d_c.data_ptr(), nf, rows.data_ptr(), cap, oc.data_ptr(), warps=d_w.data_ptr())
    torch.cuda.synchronize()
    got, n_out = rows.cpu().numpy().view(BOTSORT_ROW).reshape(nf, cap), oc.cpu().numpy()[0]
    for f in range(nf):
        assert n_out[f] == len(all_rows[f])
        np.testing.assert_array_equal(cols(got[f, :n_out[f]]), all_rows[f])


def test_botsort_with_a_camera_motion_method_demands_the_warp():
    from tracklab_amd._lib import BoTSORTBank, TlkError
    b = BoTSORTBank(32, cmc_method="sparseOptFlow")
    with pytest.raises(TlkError):
        b.update(np.zeros((1, 7)), np.ones((1, 32), np.float32))                  # no warp: refusing beats silently using the identity
    assert len(b.update(np.array([[10., 10, 50, 90, 0.9, 1, 0]]), np.ones((1, 32), np.float32), warp=np.eye(2, 3))) == 1
