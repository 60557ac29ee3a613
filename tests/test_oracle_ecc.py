"""CPU: the restatement of StrongSORT's camera-motion estimator (oracle/src/ecc.c) -- Track.ECC of plugins/track/strong_sort/sort/track.py:129-211.
cv2.findTransformECC is OpenCV and OpenCV is not installed: PARITY UNPINNED (tests/golden/make_cmc_golden.py writes the fixture that pins it
wherever cv2 exists). What can be pinned here: the estimator recovers the KNOWN Euclidean motion between synthetic frames, is the identity
for identical frames, and reports failure where OpenCV raises."""
import os

import numpy as np
import pytest

from test_oracle_cmc import _textured, _warp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def ecc_pair(seed, theta, tx, ty, h=1080, w=1920):
    """Two (h, w, 3) frames: the second one's content moved by p -> R(theta) p + t."""
    rng = np.random.default_rng(seed)
    base = _textured(rng, h // 4 + 60, w // 4 + 60)
    base = np.kron(base, np.ones((4, 4, 1), np.float32))                     # coarse texture: it must survive the 0.1 downscale
    k = np.ones(9) / 9
    for ax in (0, 1):
        base = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), ax, base)
    f0 = np.clip(np.rint(base[100:100 + h, 100:100 + w]), 0, 255).astype(np.uint8)
    f1 = _warp(base[100:, 100:], np.cos(theta), np.sin(theta), tx, ty, h, w)
    return f0, f1


def test_ecc_recovers_a_known_euclidean_motion(orc):
    theta, tx, ty = 0.004, 13.0, -8.0
    f0, f1 = ecc_pair(5, theta, tx, ty, 540, 960)
    warp, it = orc.ecc_frames(f0, f1)
    assert warp is not None and warp.dtype == np.float32 and 2 <= it <= 100      # (eps 1e-5 is below the jitter of the 1/32 px warp grid: often all 100)
    # template = previous frame, input = current: current(W x) = previous(x), and the content moved by x -> R x + t, so W = (R, t)
    np.testing.assert_allclose(warp[:, :2], [[np.cos(theta), -np.sin(theta)], [np.sin(theta), np.cos(theta)]], atol=5e-4)
    np.testing.assert_allclose(warp[:, 2], [tx, ty], atol=0.6)
    assert warp[0, 0] == warp[1, 1] and warp[0, 1] == -warp[1, 0]             # MOTION_EUCLIDEAN: a rotation
    back, _ = orc.ecc_frames(f1, f0)                                         # the inverse motion the other way round
    np.testing.assert_allclose(back[:, 2], -(np.array([[np.cos(theta), np.sin(theta)], [-np.sin(theta), np.cos(theta)]]) @ [tx, ty]), atol=0.6)


def test_ecc_identity_and_failure_cases(orc):
    f0, _ = ecc_pair(6, 0.0, 0.0, 0.0, 300, 400)
    warp, it = orc.ecc_frames(f0, f0)
    assert it >= 1 and np.abs(warp - np.eye(2, 3)).max() < 1e-4               # identical frames: rho = 1 at the identity
    small = np.random.default_rng(0).integers(0, 255, (30, 40), dtype=np.uint8)
    w, it, rho = orc.ecc_find_transform(small, small)
    assert it >= 1 and abs(rho - 1.0) < 1e-6 and np.abs(w - np.eye(2, 3)).max() < 1e-5
    flat = np.full((30, 40), 90, np.uint8)
    _, it, _ = orc.ecc_find_transform(flat, small)                           # zero-variance template: NaN correlation -> cv2.error
    assert it == -1
    assert orc.ecc_frames(np.repeat(flat[..., None], 3, 2).repeat(10, 0).repeat(10, 1), np.repeat(small[..., None], 3, 2).repeat(10, 0).repeat(10, 1))[0] is None


def test_ecc_iteration_cap_and_epsilon(orc):
    f0, f1 = ecc_pair(7, -0.003, -6.0, 4.0, 400, 600)
    a = orc.cmc_resize_gray(orc.cmc_gray(f0), 40, 60); b = orc.cmc_resize_gray(orc.cmc_gray(f1), 40, 60)
    w1, it1, _ = orc.ecc_find_transform(a, b, max_iter=1)
    w3, it3, _ = orc.ecc_find_transform(a, b, max_iter=3)
    wf, itf, rho = orc.ecc_find_transform(a, b)
    assert it1 == 1 and it3 == 3 and 3 < itf <= 100 and rho > 0.9
    assert np.abs(w1 - wf).max() > np.abs(w3 - wf).max()                    # more iterations: closer to the converged warp
    _, it_loose, _ = orc.ecc_find_transform(a, b, eps=1e-2)
    assert it_loose < 10 and it_loose <= itf


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLDEN, "cmc_opencv.npz")), reason="needs the OpenCV fixture (tests/golden/make_cmc_golden.py, run where cv2 is installed)")
def test_oracle_against_opencv_ecc_fixture(orc):
    g = np.load(os.path.join(GOLDEN, "cmc_opencv.npz"))
    if "ecc_warp" not in g:
        pytest.skip("fixture predates the ECC entries")
    w, it, rho = orc.ecc_find_transform(g["ecc_small0"], g["ecc_small1"])
    for iters in (1, 2, 5):
        np.testing.assert_allclose(orc.ecc_find_transform(g["ecc_small0"], g["ecc_small1"], max_iter=iters, eps=-1)[0], g[f"ecc_warp_small_it{iters}"], atol=1e-5)
    np.testing.assert_allclose(w, g["ecc_warp_small"], atol=1e-5)
    assert abs(rho - float(g["ecc_rho"])) < 1e-6
