"""-m gpu: camera-motion estimation on the device (tlk_cmc_*, GMC.applySparseOptFlow of plugins/track/bot_sort/gmc.py:239-303) against
the C oracle stage by stage -- grey + resize, eigenvalue image, corners, pyramid, Scharr derivatives, Lucas-Kanade tracks bit-identical
(integer arithmetic / the same float32 operations), the warp to 1e-9 (the inlier sums run in another order) -- against the known camera
motion of synthetic frame pairs, and through the BoT-SORT module (estimator + multi_gmc on the device == oracle chain).
PARITY UNPINNED with respect to OpenCV itself (not installed; tests/golden/make_cmc_golden.py produces fixtures where it is)."""
import numpy as np
import pandas as pd
import pytest

from test_oracle_cmc import _textured, _warp

pytestmark = pytest.mark.gpu


def _pair(seed, H_, W_, a, b, tx, ty):
    rng = np.random.default_rng(seed)
    base = _textured(rng, H_ + 80, W_ + 80)
    f0 = np.clip(np.rint(base[40:40 + H_, 40:40 + W_]), 0, 255).astype(np.uint8)
    return f0, _warp(base[40:, 40:], a, b, tx, ty, H_, W_)


def test_cmc_stages_equal_the_oracle(orc):
    from tracklab_amd._lib import CmcEstimator
    H_, W_ = 360, 640
    f0, f1 = _pair(5, H_, W_, np.cos(0.008), np.sin(0.008), 5.0, -3.0)
    est = CmcEstimator(H_, W_, downscale=2)
    np.testing.assert_array_equal(est.apply(f0), np.eye(2, 3))
    g0 = orc.cmc_resize_gray(orc.cmc_gray(f0), H_ // 2, W_ // 2)
    np.testing.assert_array_equal(est.debug(0), g0)
    np.testing.assert_array_equal(est.debug(1), orc.cmc_min_eigen(g0))
    c0 = orc.cmc_good_features(g0)
    np.testing.assert_array_equal(est.debug(2), c0)
    p0 = orc.CmcPyramid(g0)
    for l, (img, der) in enumerate(p0.levels()):
        np.testing.assert_array_equal(est.debug(10 + l)[:img.shape[0]], img)
        np.testing.assert_array_equal(est.debug(20 + l)[:img.shape[0]], der)
    warp = est.apply(f1)
    g1 = orc.cmc_resize_gray(orc.cmc_gray(f1), H_ // 2, W_ // 2)
    nxt, st = orc.cmc_lk(p0, orc.CmcPyramid(g1), c0)
    np.testing.assert_array_equal(est.debug(4), st)
    np.testing.assert_array_equal(est.debug(3)[st], nxt[st])
    ref = orc.SparseOptFlowGMC(H_, W_, 2)
    ref.apply(f0)
    exp = ref.apply(f1)
    assert est.inliers == ref.inliers and ref.inliers > 50
    np.testing.assert_allclose(warp, exp, rtol=0, atol=1e-9)
    est.close()


@pytest.mark.parametrize("angle,scale,tx,ty", [(0.0, 1.0, 6.0, -4.0), (0.01, 1.0, -9.0, 5.0), (-0.006, 1.01, 3.0, 8.0)])
def test_cmc_recovers_a_known_camera_motion_and_follows_the_oracle_over_a_sequence(orc, angle, scale, tx, ty):
    from tracklab_amd._lib import CmcEstimator
    H_, W_ = 360, 640
    a, b = scale * np.cos(angle), scale * np.sin(angle)
    f0, f1 = _pair(11, H_, W_, a, b, tx, ty)
    est, ref = CmcEstimator(H_, W_, 2), orc.SparseOptFlowGMC(H_, W_, 2)
    for fr in (f0, f1, f0, f1):                                  # forth and back: the state (previous pyramid, corners) rolls over
        w, e = est.apply(fr), ref.apply(fr)
        np.testing.assert_allclose(w, e, rtol=0, atol=1e-9)
    np.testing.assert_allclose(w[:, :2], [[a, -b], [b, a]], atol=2e-3)
    np.testing.assert_allclose(w[:, 2], [tx, ty], atol=0.35)
    est.reset()
    np.testing.assert_array_equal(est.apply(f1), np.eye(2, 3))    # new video: the first frame only initialises
    est.close()


def test_cmc_device_entry_point_leaves_the_warp_on_the_device(orc):
    import torch
    from tracklab_amd._lib import CmcEstimator
    H_, W_ = 360, 640
    f0, f1 = _pair(7, H_, W_, 1.0, 0.0, -4.0, 2.0)
    est, ref = CmcEstimator(H_, W_, 2), orc.SparseOptFlowGMC(H_, W_, 2)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for fr in (f0, f1):
            w = est.apply_dev(torch.from_numpy(np.ascontiguousarray(fr)).cuda())
    side.synchronize()
    ref.apply(f0)
    np.testing.assert_allclose(w.cpu().numpy().reshape(2, 3), ref.apply(f1), rtol=0, atol=1e-9)
    est.close()


def test_botsort_module_with_sparse_optical_flow_matches_the_oracle_chain(orc):
    """HipBoTSORT with the reference's default cmc_method: the frames of a panning camera (every box and the background shift together), the
    module's rows == oracle BoT-SORT fed with the oracle estimator's warps and the module's own ReID features."""
    from types import SimpleNamespace as NS
    from torch.utils.data.dataloader import default_collate
    from test_modules_host import _frame_df
    from tracklab_amd.synth import SyntheticStream
    from tracklab_amd.wrappers import HipBoTSORT
    H_, W_, D = 360, 640, 32
    hyper = dict(track_high_thresh=0.5, new_track_thresh=0.6, track_buffer=10, match_thresh=0.8, proximity_thresh=0.5, appearance_thresh=0.25,
                 cmc_method="sparseOptFlow", frame_rate=30, lambda_=0.985)
    core = {k: v for k, v in hyper.items() if k != "cmc_method"}
    m = HipBoTSORT(NS(min_confidence=0.4, feature_dim=D, hyperparams=hyper, max_dets=64), "cuda:0", tracking_dataset=None)
    ref, gmc = orc.BoTSORT(D, **core), orc.SparseOptFlowGMC(H_, W_, 2)
    rng = np.random.default_rng(2)
    base = _textured(rng, H_ + 200, W_ + 200)
    feats_seen = []
    orig = m._features

    def features(image, dets):
        f = orig(image, dets)
        feats_seen.append(f)
        return f
    m._features = features
    rows = 0
    for t, fr in enumerate(SyntheticStream(3, 12, 14, miss_prob=0.05)):
        ox, oy = 60 + 4 * t, 50 + 2 * t                               # the camera pans: the view window slides over the scene
        frame = np.clip(np.rint(base[oy:oy + H_, ox:ox + W_]), 0, 255).astype(np.uint8)
        d = fr["dets"].copy()
        d[:, [0, 2]] = d[:, [0, 2]] / 3.0 - 4 * t + 40                  # 1080p boxes squeezed into the small frame, moving with the pan
        d[:, [1, 3]] = d[:, [1, 3]] / 3.0 - 2 * t + 20
        d[:, [0, 2]] = np.clip(d[:, [0, 2]], 1, W_ - 2); d[:, [1, 3]] = np.clip(d[:, [1, 3]], 1, H_ - 2)
        d = d[(d[:, 2] - d[:, 0] > 4) & (d[:, 3] - d[:, 1] > 4)]
        df = _frame_df(dict(fr, dets=d), np.float64, id0=100 * t)
        feats_seen.clear()
        sample = m.preprocess(frame, df, pd.Series({"frame": t}))
        out = m.process(default_collate([sample]), df, pd.DataFrame({"file_path": ["unused"]}))
        inp = sample["input"]
        keep = inp[:, 4] > 0.4
        hi = keep & (inp[:, 4] > 0.5)
        f = np.zeros((len(inp), D), np.float32)
        if hi.any():
            f[hi] = feats_seen[0]
        exp = ref.update(inp[keep], f[keep], warp=gmc.apply(frame))
        assert len(out) == len(exp), t
        if len(exp):
            order = np.argsort(exp[:, 7], kind="stable")
            got = out.sort_index(kind="stable")
            np.testing.assert_array_equal(got.index.to_numpy(), exp[order, 7].astype(int))
            np.testing.assert_array_equal(got.track_id.to_numpy(), exp[order, 4])
            rows += len(exp)
    assert rows > 60


def test_cmc_is_exact_while_a_resnet_forward_runs_on_another_stream():
    """The estimator runs on its own stream under the ReID forward in the fused step (gpu_pipeline.DetReidTrackPipeline(camera_motion=True)).
    With packed-FP32 code in the LK kernel (v_pk_mul_f32 / v_pk_add_f32 for its dx, dy update) ~5 % of the tracked points came out different
    whenever bf16 convolutions shared the compute units, and exact alone (profiles/r02_pk_f32_overlap.md); csrc/build.sh now compiles every
    side-stream kernel without packed-FP32 instructions.  Here: the same frames through an estimator under load and one on the idle GPU."""
    import ctypes as C
    import torch
    from tracklab_amd._lib import CmcEstimator
    from tracklab_amd.backbones.reid import part_based_reid
    H_, W_ = 1080, 1920
    rng = np.random.default_rng(11)
    frames = []
    for k in range(3):                                        # unrelated noisy frames: every LK point iterates to its limit
        f0, f1 = _pair(20 + k, H_, W_, np.cos(0.004 * k), np.sin(0.004 * k), 3.0 + k, -2.0)
        frames += [f0, f1]
    frames = [np.ascontiguousarray(np.clip(f.astype(np.int16) + rng.integers(-12, 13, f.shape), 0, 255).astype(np.uint8)) for f in frames]
    reid = part_based_reid(1, 128, device="cuda", dtype=torch.bfloat16, channels_last=True)
    crops = torch.randn(96, 3, 256, 128, device="cuda", dtype=torch.bfloat16).to(memory_format=torch.channels_last)
    with torch.no_grad():
        reid(crops)
    dev = torch.from_numpy(np.stack(frames)).cuda()
    busy, idle = CmcEstimator(H_, W_, 2), CmcEstimator(H_, W_, 2)
    side = torch.cuda.Stream()
    out = torch.zeros(len(frames), 6, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    n_pts = 0
    for k in range(len(frames)):
        with torch.no_grad():
            for _ in range(2):
                reid(crops)                                   # queued first: the estimator's kernels run under it
        busy.apply_dev(dev[k], stream_ptr=C.c_void_p(side.cuda_stream), out=out[k])
        torch.cuda.synchronize()
        w = idle.apply(frames[k])
        np.testing.assert_array_equal(busy.debug(3), idle.debug(3), err_msg=f"LK points of frame {k}")
        np.testing.assert_array_equal(busy.debug(4), idle.debug(4))
        np.testing.assert_array_equal(out[k].cpu().numpy().reshape(2, 3), w)
        n_pts += len(idle.debug(3))
    assert n_pts > 500                                        # the comparison covered the tracked points of five frame pairs
    busy.close(), idle.close()
