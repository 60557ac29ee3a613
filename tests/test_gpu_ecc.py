"""-m gpu: StrongSORT's camera-motion estimator on the device (tlk_ecc.hip) against its CPU restatement (oracle/src/ecc.c): Track.ECC of
plugins/track/strong_sort/sort/track.py:129-211. Both restate cv2.findTransformECC (PARITY UNPINNED, see the oracle's header); the kernel
follows the oracle operation for operation, including the summation order, so the comparison is exact up to the last bit of the three
libm calls per iteration (asin / cos / sin in double, rounded to float32)."""
import numpy as np
import pandas as pd
import pytest
from torch.utils.data.dataloader import default_collate

from test_oracle_ecc import ecc_pair

pytestmark = pytest.mark.gpu


class NS(dict):
    __getattr__ = dict.get


def test_find_transform_on_device_equals_the_oracle(orc):
    from tracklab_amd import _lib
    for seed, (theta, tx, ty), (h, w) in ((1, (0.004, 13.0, -8.0), (540, 960)), (2, (-0.01, -25.0, 11.0), (1080, 1920)), (3, (0.0, 0.0, 0.0), (300, 400))):
        f0, f1 = ecc_pair(seed, theta, tx, ty, h, w)
        dh, dw = int(np.rint(h * 0.1)), int(np.rint(w * 0.1))
        a, b = orc.cmc_resize_gray(orc.cmc_gray(f0), dh, dw), orc.cmc_resize_gray(orc.cmc_gray(f1), dh, dw)
        for max_iter in (1, 4, 100):
            exp, it_e, rho_e = orc.ecc_find_transform(a, b, max_iter=max_iter)
            got, it_g, rho_g = _lib.ecc_find_transform(a, b, max_iter=max_iter)
            assert it_g == it_e, (seed, max_iter, it_g, it_e)
            np.testing.assert_allclose(got, exp, rtol=0, atol=2e-6, err_msg=str((seed, max_iter)))
            assert abs(rho_g - rho_e) < 1e-9
    flat = np.full(a.shape, 90, np.uint8)
    assert _lib.ecc_find_transform(flat, a)[1] == -1                         # zero-variance template -> NaN correlation: where cv2 raises


def test_estimator_on_frames_equals_the_oracle_and_recovers_the_motion(orc):
    import torch
    from tracklab_amd import _lib
    theta, tx, ty = 0.003, 9.0, -14.0
    f0, f1 = ecc_pair(4, theta, tx, ty)
    est = _lib.EccEstimator(1080, 1920)
    assert est.apply(f0) is None and est.iterations == 0                      # first frame of a video: no previous frame, no update
    got = est.apply(f1)
    exp, it = orc.ecc_frames(f0, f1)
    assert est.iterations == it and got.dtype == np.float32
    np.testing.assert_allclose(got, exp, rtol=0, atol=2e-5)
    np.testing.assert_allclose(got[:, 2], [tx, ty], atol=0.6)
    np.testing.assert_allclose(got[1, 0], np.sin(theta), atol=5e-4)
    warp_dev, status_dev = est.apply_dev(torch.from_numpy(np.ascontiguousarray(f0)).cuda())          # device entry: f1 -> f0, results stay in HBM
    back, it_b = orc.ecc_frames(f1, f0)
    torch.cuda.synchronize()
    assert int(status_dev.item()) == it_b
    np.testing.assert_allclose(warp_dev.cpu().numpy().reshape(2, 3), back, rtol=0, atol=2e-5)
    est.reset()
    assert est.apply(f1) is None
    est.close()


def test_hip_strongsort_module_with_ecc_true_equals_the_oracle_chain(orc):
    """`ecc: true` (the reference's default, strong_sort.yaml:13) end to end through the plugin API: frames from a panning camera, the warp
    estimated on the GPU and applied to the tracks on the GPU, against the oracle tracker driven with the oracle's ECC in the reference's
    order (strong_sort_api.py:60-72)."""
    from test_modules_host import _frame_df
    from tracklab_amd.synth import SyntheticStream
    from tracklab_amd.wrappers import HipStrongSORT
    hyper = dict(ema_alpha=0.9, max_age=10, max_dist=0.3, max_iou_dist=0.7, max_unmatched_preds=7, mc_lambda=0.995, n_init=2, nn_budget=10)
    m = HipStrongSORT(NS(min_confidence=0.4, ecc=True, feature_dim=64, hyperparams=hyper), "cuda:0", tracking_dataset=None)
    ref = orc.PlainStrongSORT(64, **hyper, img_w=1920, img_h=1080)
    feats_seen = []
    orig = m._features
    m._features = lambda image, dets: feats_seen.append(orig(image, dets)) or feats_seen[-1]
    pans = [(0.0, 0.0, 0.0), (0.001, 6.0, -2.0), (0.002, 12.0, -5.0), (0.002, 19.0, -7.0), (0.001, 25.0, -8.0), (0.0, 30.0, -8.0), (-0.001, 33.0, -6.0), (-0.001, 35.0, -3.0)]
    frames = [ecc_pair(9, *p)[1] for p in pans]                              # the same scene under a moving camera
    prev, n_rows, n_warps = None, 0, 0
    for fr, img in zip(SyntheticStream(6, 15, len(frames), miss_prob=0.05), frames):
        df = _frame_df(fr, np.float64, id0=200)
        sample = m.preprocess(img, df, pd.Series({"frame": fr["frame"]}))
        out = m.process(default_collate([sample]), df, pd.DataFrame({"file_path": ["unused"]}))
        if prev is not None:
            w, _ = orc.ecc_frames(prev, img)
            if w is not None:
                ref.camera_update(w); n_warps += 1
        prev = img
        f = feats_seen[-1]
        keep = sample["input"][:, 4] > 0.4
        exp = ref.update(sample["input"][keep], f[keep])
        assert len(out) == len(exp)
        if len(exp):
            np.testing.assert_array_equal(out.index.to_numpy(), exp[:, 7].astype(int))
            np.testing.assert_array_equal(out.track_id.to_numpy(), exp[:, 4])
            np.testing.assert_allclose(np.stack(out.track_bbox_ltwh.to_list())[:, :2], exp[:, :2], rtol=0, atol=1e-3)    # (a last-bit float32 warp difference moves a box by < 1e-4 px)
            n_rows += len(exp)
    assert n_rows > 40 and n_warps == len(frames) - 1
