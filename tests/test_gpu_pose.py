"""-m gpu: pose pre/post-processing kernels (tlk_pose_crop_warp_norm, tlk_simcc_decode) through the C ABI vs the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(seed, n):
    from tracklab_amd.synth import SyntheticStream, render_frame
    rng = np.random.default_rng(seed)
    fr = SyntheticStream(seed, n, 1).step()
    frame = render_frame(rng, fr["gt_boxes"])
    frame[::5, ::3] = rng.integers(0, 255, frame[::5, ::3].shape)        # texture so that interpolation errors show
    dets = fr["dets"].copy()
    dets[0, :4] = [-30.0, -20.0, 90.0, 260.0]            # leaves the image: constant-0 border
    dets[1, :4] = [1800.0, 900.0, 1990.0, 1150.0]
    dets[2, :4] = [400.0, 300.0, 900.0, 420.0]           # wide box: the width fixes the scale
    return frame, dets


@pytest.mark.parametrize("dtype_name,layout", [("float32", "nchw"), ("float32", "nhwc"), ("float16", "nhwc")])
def test_pose_crop_warp_norm_bit_exact(orc, dtype_name, layout):
    import torch
    from tracklab_amd import _lib
    frame, dets = _scene(3, 40)
    n = len(dets)
    frames = torch.from_numpy(np.stack([frame, frame[::-1].copy()])).cuda()
    boxes = torch.zeros((2, n + 3, 7), dtype=torch.float64, device="cuda")
    boxes[0, :n] = torch.from_numpy(dets).cuda(); boxes[1, :5] = torch.from_numpy(dets[:5]).cuda()
    counts = torch.tensor([n, 5], dtype=torch.int32, device="cuda")
    dt = getattr(torch, dtype_name)
    out, meta = _lib.pose_crop_warp_norm(frames, boxes, counts, 192, 256, layout, dt)
    out, meta = out.float().cpu().numpy(), meta.cpu().numpy()
    for b, (img, cnt) in enumerate(((frame, n), (frame[::-1].copy(), 5))):
        for i in range(cnt):
            exp, c, s = orc.rtmpose_preprocess(img, dets[i, :4])
            slot = b * (n + 3) + i
            np.testing.assert_array_equal(meta[slot, :2], c); np.testing.assert_array_equal(meta[slot, 2:4], s)
            if dt == torch.float16:
                exp = exp.astype(np.float16).astype(np.float32)
            np.testing.assert_array_equal(out[slot], exp, err_msg=f"frame {b} box {i}")
        assert not out[b * (n + 3) + cnt:(b + 1) * (n + 3)].any()


def test_simcc_decode_matches_oracle_and_feeds_the_tracker_layout(orc):
    import torch
    from tracklab_amd import _lib
    rng = np.random.default_rng(5)
    n, K, Wx, Wy = 37, 17, 384, 512
    sx = rng.normal(0, 1, (n, K, Wx)).astype(np.float32); sy = rng.normal(0, 1, (n, K, Wy)).astype(np.float32)
    sx[3, 4] = -1.0; sx[3, 4, 10] = -0.5                  # non-positive maximum -> location -1
    sx[5, 0, 20] = sx[5, 0].max() + 1; sx[5, 0, 300] = sx[5, 0, 20]      # tie: first maximum
    meta = np.zeros((n, 10)); meta[:, 0] = rng.uniform(100, 1800, n); meta[:, 1] = rng.uniform(100, 900, n)
    meta[:, 3] = rng.uniform(100, 400, n); meta[:, 2] = meta[:, 3] * 0.75
    out = _lib.simcc_decode(torch.from_numpy(sx).cuda(), torch.from_numpy(sy).cuda(), torch.from_numpy(meta).cuda())
    kps, sc, conf = out["kps_xyc"].cpu().numpy(), out["scores"].cpu().numpy(), out["conf"].cpu().numpy()
    for i in range(n):
        ek, es = orc.simcc_decode(sx[i], sy[i], meta[i, :2], meta[i, 2:4])
        np.testing.assert_array_equal(kps[i, :, :2], ek); np.testing.assert_array_equal(sc[i], es)
        np.testing.assert_array_equal(kps[i, :, 2], es.astype(np.float64))
        assert conf[i] == np.mean(es)                     # keypoints_conf = np.mean(scores) in float32
    assert kps.shape == (n, 17, 3) and kps.dtype == np.float64      # = the keypoints argument of tlk_bpbss_update


def test_pose_entry_points_reject_bad_arguments():
    from tracklab_amd._lib import TlkError, _bind_pose, check, lib
    L = lib(); _bind_pose(L)
    with pytest.raises(TlkError):
        check(L.tlk_pose_crop_warp_norm(None, 1, 1080, 1920, None, 7, None, 4, 190, 256, None, None, 0, 0, None, None, None))   # in_w % 8
    with pytest.raises(TlkError):
        check(L.tlk_simcc_decode(None, None, 3, 17, 384, 512, 2.0, None, 192, 256, None, None, None, None))
    check(L.tlk_simcc_decode(None, None, 0, 17, 384, 512, 2.0, None, 192, 256, None, None, None, None))
