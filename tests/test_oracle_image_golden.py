"""CPU: oracle/src/image.c and pose.c against outputs of OpenCV / rtmlib / albumentations THEMSELVES (tests/golden/image_opencv.npz, written by
tests/golden/make_image_golden.py wherever cv2 is installed).  The build container has none of the three, so the file is absent there and these
tests skip -- rows W3 / N1 / N2 stay "parity unpinned" until someone runs the generator; the first run turns them into pinned rows without
touching a line of product code.  Reference call sites: wrappers/bbox_detector/rtmlib_api.py:27-46, wrappers/reid/kpreid_api.py:115-144,
wrappers/pose_estimator/rtmlib_api.py:27-33."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

PATH = os.path.join(GOLDEN, "image_opencv.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(PATH), reason="tests/golden/image_opencv.npz absent: run tests/golden/make_image_golden.py where OpenCV is installed")


@pytest.fixture(scope="module")
def g():
    return np.load(PATH)


def test_letterbox_equals_rtmlib_yolox_preprocess(orc, g):
    for i in range(int(g["lb_n"])):
        out, ratio = orc.letterbox(g[f"lb{i}_img"], 640)
        assert ratio == float(g[f"lb{i}_ratio"]), i
        np.testing.assert_array_equal(out, g[f"lb{i}_padded"].transpose(2, 0, 1).astype(np.float32), err_msg=f"frame size {g[f'lb{i}_img'].shape}")


def test_reid_crop_resize_equals_cv2_and_albumentations(orc, g):
    frame = g["rz_frame"]
    for k in range(int(g["rz_n"])):
        l, t, r, b = g[f"rz{k}_ltrb"]
        oh, ow = g[f"rz{k}_hw"]
        np.testing.assert_array_equal(orc.cv_resize_linear_u8(frame[t:b, l:r], int(oh), int(ow)), g[f"rz{k}_resized"], err_msg=f"crop {k}")
        if f"rz{k}_normalized" in g.files:
            got = orc.crop_resize_norm(frame, np.array([[l, t, r, b]]), int(oh), int(ow))[0]
            np.testing.assert_array_equal(got, g[f"rz{k}_normalized"].transpose(2, 0, 1), err_msg=f"crop {k} normalised")


def test_pose_warp_equals_cv2_warpaffine_and_rtmlib_preprocess(orc, g):
    frame = g["wa_frame"]
    for k in range(int(g["wa_n"])):
        np.testing.assert_array_equal(orc.cv_warp_affine_linear(frame, g[f"wa{k}_M"], 256, 192), g[f"wa{k}_warped"], err_msg=f"box {k}")
        if f"wa{k}_crop" in g.files:                          # rtmlib RTMPose.preprocess itself
            crop, c, s = orc.rtmpose_preprocess(frame, g[f"wa{k}_box"])
            np.testing.assert_array_equal(c, g[f"wa{k}_center"]); np.testing.assert_array_equal(s, g[f"wa{k}_scale"])
            np.testing.assert_allclose(orc.rtmpose_warp_matrix(c, s), g[f"wa{k}_M"], rtol=0, atol=1e-12)
            np.testing.assert_array_equal(crop, np.float32(g[f"wa{k}_crop"]).transpose(2, 0, 1))
        if f"wa{k}_kpts" in g.files:                          # rtmlib RTMPose.postprocess (SimCC decode)
            kps, sc = orc.simcc_decode(g[f"wa{k}_simcc_x"], g[f"wa{k}_simcc_y"], g[f"wa{k}_center"], g[f"wa{k}_scale"])
            np.testing.assert_array_equal(sc, g[f"wa{k}_scores"]); np.testing.assert_allclose(kps, g[f"wa{k}_kpts"], rtol=0, atol=1e-9)


def test_yolox_postprocess_equals_rtmlib(orc, g):
    if "yx_head" not in g.files:
        pytest.skip("fixture generated without rtmlib")
    boxes, scores, cls = orc.yolox_postprocess(g["yx_head"], 640, float(np.float32(g["yx_ratio"])))
    keep = (scores > 0.3) & (cls == 0)                        # rtmlib's final filter (wrappers/bbox_detector/rtmlib_api.py consumes these boxes)
    np.testing.assert_allclose(boxes[keep], g["yx_boxes"], rtol=0, atol=1e-4)
