"""CPU: the oracle's stateless KF7 step and the StrongSORT motion costs against vectors produced by the reference itself
(tests/golden/make_golden.py gen_kf7 / gen_motion_costs)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def bbox_to_z(b):
    """oc_sort/ocsort.py:21-34 convert_bbox_to_z."""
    w, h = b[2] - b[0], b[3] - b[1]
    return np.array([b[0] + w / 2.0, b[1] + h / 2.0, w * h, w / float(h + 1e-6)])


def kf7_chain_steps(g):
    """(x_prev, P_prev, z, x_next, P_next) for every step of the KalmanBoxTracker replays in which plain predict + update(z)
    links two stored states: the step and the two before it observed (no freeze/unfreeze in between)."""
    for c in range(int(g["n_cases"])):
        obs, pat = g[f"c{c}_obs"], g[f"c{c}_pattern"]
        for t in range(2, len(pat) + 1):          # state index t-1 is after step t (1-based)
            if pat[t - 1] and pat[t - 2] and (t < 3 or pat[t - 3]):
                yield g[f"c{c}_x"][t - 2], g[f"c{c}_P"][t - 2], bbox_to_z(obs[t][:4]), g[f"c{c}_x"][t - 1], g[f"c{c}_P"][t - 1]


def test_kf7_stateless_matches_reference_states(orc):
    g = np.load(os.path.join(GOLDEN, "kf7_cases.npz"))
    n = 0
    for x0, P0, z, x1, P1 in kf7_chain_steps(g):
        x, P = orc.kf7_predict(x0, P0)
        x, P = orc.kf7_update(x, P, z)
        np.testing.assert_allclose(x, x1, rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(P, P1, rtol=1e-9, atol=1e-9)
        n += 1
    assert n >= 20


def test_motion_costs_match_reference(orc):
    g = np.load(os.path.join(GOLDEN, "motion_costs.npz"))
    for c in range(int(g["n_cases"])):
        iou_c = orc.iou_ltwh_cost(g[f"c{c}_trk_ltwh"], g[f"c{c}_det_ltwh"])
        np.testing.assert_array_equal(iou_c, g[f"c{c}_iou_cost"])                       # same op order -> bit-exact
        oks_c = orc.oks_cost(g[f"c{c}_trk_kps"], g[f"c{c}_det_kps"])
        np.testing.assert_allclose(oks_c, g[f"c{c}_oks_cost"], rtol=1e-12, atol=1e-15, equal_nan=True)   # exp() of libm vs numpy
    assert np.isnan(g["c0_oks_cost"][1]).all()         # the single-visible-keypoint track: scale < 0.1 -> NaN (oks_matching.py:80-81)


def test_pil_reid_preprocess_matches_pillow(orc):
    """crop box, Pillow bilinear resize (bit-exact) and ToTensor+Normalize (bit-exact float32)."""
    g = np.load(os.path.join(GOLDEN, "pil_preprocess.npz"))
    img = g["image"]
    H, W = img.shape[:2]
    np.testing.assert_array_equal(orc.ssort_crop_box(g["boxes"], W, H), g["boxes_int"])
    for i, b in enumerate(g["boxes"]):
        out, u8 = orc.ssort_reid_preprocess(img, b)
        np.testing.assert_array_equal(u8, g["resized"][i], err_msg=f"box {i}")
        if f"norm{i}" in g.files:
            np.testing.assert_array_equal(out, g[f"norm{i}"])
        for c in range(3):                         # every crop: normalisation = the (value, channel) table torch produced
            np.testing.assert_array_equal(out[c], g["norm_lut"][c][u8[:, :, c]])


def _pil_sweep_frame(w_img, H=720):
    """tests/golden/make_golden.py::pil_sweep_image restated (the fixture stores hashes, not the 2.7 MB frames)."""
    yy, xx = np.mgrid[0:H, 0:w_img].astype(np.int64)
    tex = ((xx * 1103515245 + yy * 12345 + xx * yy * 7) >> 3) % 13 - 6
    img = np.stack([(xx * 255 // w_img) + tex, (yy * 255 // H) - tex, ((xx * 3 + yy * 5) % 256) + tex // 2], axis=2)
    for k in range(60):
        x, y = (k * 197) % (w_img - 90), (k * 113) % (H - 170)
        w, h = 10 + (k * 31) % 70, 10 + (k * 53) % 150
        img[y:y + h, x:x + w] = [(k * 37) % 256, (k * 91) % 256, (k * 151) % 256]
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_pil_reid_preprocess_matches_pillow_over_a_sweep_of_crop_sizes(orc, tag):
    """r03 fixture pil_sweep.npz: 40 crops of 2 .. 200 x 2 .. 600 px through Pillow itself (2 / 3 / 5 taps per axis, scales around 1.2 and 2, clipped
    and tiny crops; frame widths 1280 and 1283): SHA-256 of the full resized array and every 8th row equal."""
    import hashlib
    g = np.load(os.path.join(GOLDEN, "pil_sweep.npz"))
    img = _pil_sweep_frame(int(g[f"width_{tag}"]))
    np.testing.assert_array_equal(orc.ssort_crop_box(g[f"boxes_{tag}"], img.shape[1], img.shape[0]), g[f"boxes_int_{tag}"])
    for i, b in enumerate(g[f"boxes_{tag}"]):
        out, u8 = orc.ssort_reid_preprocess(img, b)
        np.testing.assert_array_equal(u8[::8], g[f"rows_{tag}"][i], err_msg=f"box {i}")
        assert hashlib.sha256(np.ascontiguousarray(u8).tobytes()).hexdigest() == str(g[f"sha_{tag}"][i]), f"box {i}"
        for c in range(3):
            np.testing.assert_array_equal(out[c], g["norm_lut"][c][u8[:, :, c]])
