"""CPU: the pose pre/post-processing oracle (oracle/src/pose.c). rtmlib / cv2 are not installed and no reference test pins this
arithmetic (PARITY UNPINNED); what can be checked here are the properties the published algorithms guarantee."""
import numpy as np


def test_warp_affine_identity_shift_and_interpolation(orc):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 255, (120, 160, 3), dtype=np.uint8)
    np.testing.assert_array_equal(orc.cv_warp_affine_linear(img, [[1, 0, 0], [0, 1, 0]], 120, 160), img)
    out = orc.cv_warp_affine_linear(img, [[1, 0, 5], [0, 1, -3]], 120, 160)           # dst(x, y) = src(x - 5, y + 3)
    np.testing.assert_array_equal(out[0:117, 5:160], img[3:120, 0:155])
    assert not out[117:].any() and not out[:, :5].any()                                # constant-0 border
    up = orc.cv_warp_affine_linear(img, [[2, 0, 0], [0, 2, 0]], 240, 320)
    np.testing.assert_array_equal(up[0:238:2, 0:318:2], img[:119, :159])              # integer positions reproduce the source
    mid = (img[:119, :159].astype(int) + img[:119, 1:160].astype(int) + 1) >> 1        # half-way taps: 16384-weight pairs
    np.testing.assert_array_equal(up[0:238:2, 1:318:2], mid)


def test_rtmpose_center_scale_and_matrix(orc):
    crop, c, s = orc.rtmpose_preprocess(np.zeros((300, 400, 3), np.uint8), [100, 50, 200, 250])
    np.testing.assert_array_equal(c, [150, 150])
    np.testing.assert_array_equal(s, [187.5, 250.0])           # 100x200 box * 1.25 -> (125, 250) -> aspect 0.75 -> (187.5, 250)
    M = orc.rtmpose_warp_matrix(c, s)
    np.testing.assert_allclose(M, [[1.024, 0, -57.6], [0, 1.024, -25.6]], atol=1e-9)   # 192 / 187.5; centre -> (96, 128)
    assert crop.shape == (3, 256, 192)
    np.testing.assert_array_equal(crop[:, 0, 0], np.float32((0 - np.array(orc.RTMPOSE_MEAN)) / np.array(orc.RTMPOSE_STD)))
    _, c2, s2 = orc.rtmpose_preprocess(np.zeros((300, 400, 3), np.uint8), [100, 100, 300, 150])       # wide box: width rules
    np.testing.assert_array_equal(s2, [250.0, 250.0 / 0.75])


def test_simcc_decode_semantics(orc):
    K, Wx, Wy = 17, 384, 512
    sx, sy = np.full((K, Wx), -1.0, np.float32), np.full((K, Wy), -1.0, np.float32)
    sx[0, 100] = 0.9; sy[0, 300] = 0.4            # score = the smaller maximum
    sx[1, 7] = 0.5; sx[1, 9] = 0.5; sy[1, 0] = 2  # first maximum wins
    sx[2, 50] = 0.3; sy[2, 60] = -0.2             # non-positive score -> location -1
    c, s = np.array([500.0, 400.0]), np.array([150.0, 200.0])
    kps, sc = orc.simcc_decode(sx, sy, c, s)
    np.testing.assert_array_equal(sc[:3], np.float32([0.4, 0.5, -0.2]))
    np.testing.assert_allclose(kps[0], [50 / 192 * 150 + 500 - 75, 150 / 256 * 200 + 400 - 100])
    np.testing.assert_allclose(kps[1], [3.5 / 192 * 150 + 500 - 75, 0 / 256 * 200 + 400 - 100])
    np.testing.assert_allclose(kps[2], [-0.5 / 192 * 150 + 500 - 75, -0.5 / 256 * 200 + 400 - 100])
    rng = np.random.default_rng(1)
    for n in (3, 8, 17, 33):
        a = rng.normal(0, 1, n).astype(np.float32)
        assert orc.mean_f32_numpy(a) == np.mean(a)           # numpy's own float32 pairwise mean, bit for bit
