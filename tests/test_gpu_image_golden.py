"""-m gpu: the HIP image kernels against outputs of OpenCV / rtmlib / albumentations themselves (tests/golden/image_opencv.npz; skipped while the
fixture is absent, see tests/test_oracle_image_golden.py and tests/golden/make_image_golden.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

PATH = os.path.join(GOLDEN, "image_opencv.npz")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(PATH), reason="tests/golden/image_opencv.npz absent: run tests/golden/make_image_golden.py where OpenCV is installed")]


def test_hip_letterbox_equals_rtmlib_yolox_preprocess():
    import torch
    from tracklab_amd import _lib
    g = np.load(PATH)
    for i in range(int(g["lb_n"])):
        x, ratio = _lib.letterbox(torch.from_numpy(g[f"lb{i}_img"][None]).cuda(), 640, "nchw", torch.float32)
        assert ratio == float(g[f"lb{i}_ratio"])
        np.testing.assert_array_equal(x[0].cpu().numpy(), g[f"lb{i}_padded"].transpose(2, 0, 1).astype(np.float32))


def test_hip_reid_crops_equal_cv2_resize_and_albumentations_normalize():
    import torch
    from tracklab_amd import _lib
    g = np.load(PATH)
    if "rz0_normalized" not in g.files:
        pytest.skip("fixture generated without albumentations")
    frame = torch.from_numpy(g["rz_frame"][None]).cuda()
    for k in range(int(g["rz_n"])):
        l, t, r, b = [float(v) for v in g[f"rz{k}_ltrb"]]
        oh, ow = [int(v) for v in g[f"rz{k}_hw"]]
        ltwh = torch.tensor([[[l, t, r - l, b - t]]], dtype=torch.float32).cuda()
        crops = _lib.roi_crop_resize_norm(frame, ltwh, torch.tensor([1], dtype=torch.int32).cuda(), oh, ow, "nchw", torch.float32)
        np.testing.assert_array_equal(crops[0].cpu().numpy(), g[f"rz{k}_normalized"].transpose(2, 0, 1), err_msg=f"crop {k}")


def test_hip_pose_crops_equal_rtmlib_preprocess():
    import torch
    from tracklab_amd import _lib
    g = np.load(PATH)
    if "wa0_crop" not in g.files:
        pytest.skip("fixture generated without rtmlib")
    frame = torch.from_numpy(g["wa_frame"][None]).cuda()
    n = int(g["wa_n"])
    boxes = torch.from_numpy(np.stack([g[f"wa{k}_box"] for k in range(n)])[None]).cuda()
    crops, meta = _lib.pose_crop_warp_norm(frame, boxes, torch.tensor([n], dtype=torch.int32).cuda(), 192, 256, "nchw", torch.float32)
    for k in range(n):
        np.testing.assert_array_equal(crops[k].cpu().numpy(), np.float32(g[f"wa{k}_crop"]).transpose(2, 0, 1), err_msg=f"box {k}")
