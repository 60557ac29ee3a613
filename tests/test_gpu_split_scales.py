"""-m gpu: r06, scaled split planes (tlk_conv2d_nhwc_16s, tlk_split_f32_planes_s, tlk_merge_planes_f32_s, tlk_split_scale_update): value =
scale * (hi + lo * 2^-11) with a power-of-two scale per tensor.  (a) scale 1 is the unscaled call, bit for bit; (b) activations far beyond
float16's range go through a convolution chain with the SAME fp64 error bound as the exact-fp32 kernel (2e-6 * |x| conv |w|); (c) the update rule."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cl(t):
    import torch
    return t.contiguous(memory_format=torch.channels_last)


def _state(scale=1.0):
    import torch
    return torch.tensor([scale, 0.0], dtype=torch.float32, device="cuda")


@pytest.mark.parametrize("cin,cout,k,res", [(64, 256, 1, True), (256, 64, 1, False), (64, 64, 3, False), (512, 2048, 1, True), (128, 128, 3, False)])
def test_scale_one_is_the_unscaled_convolution_bit_for_bit(cin, cout, k, res):
    import torch
    from tracklab_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(cin + cout)
    n, h, w = (40, 24, 8) if cout >= 1024 else (96, 24, 8)
    x = _cl(torch.randn(n, cin, h, w, device="cuda", generator=g) * 3)
    wt = _cl(torch.randn(cout, cin, k, k, device="cuda", generator=g) * (1.0 / (cin * k * k)) ** 0.5)
    b = torch.randn(cout, device="cuda", generator=g)
    r = _cl(torch.randn(n, cout, h, w, device="cuda", generator=g)) if res else None
    xh, xl = _lib.split_planes(x); wh, wl = _lib.split_planes(wt)
    rh, rl = _lib.split_planes(r) if res else (None, None)
    y0 = _lib.conv2d_nhwc_16(xh, wh, b, "relu", rh, x_lo=xl, weight_lo=wl, residual_lo=rl)
    si, sr, so = _state(), _state(), _state()
    y1 = _lib.conv2d_nhwc_16(xh, wh, b, "relu", rh, x_lo=xl, weight_lo=wl, residual_lo=rl, in_scale=si, res_scale=sr if res else None, out_state=so)
    assert torch.equal(y0[0], y1[0]) and torch.equal(y0[1], y1[1])
    full = _lib.merge_planes(*y0)
    assert float(so[0]) == 1.0 and abs(float(so[1]) - float(full.abs().max())) <= 1e-6 * float(full.abs().max())      # the recorded maximum is the largest |output|


def test_convolutions_far_beyond_float16s_range_keep_the_fp64_bound():
    """x up to ~4e5, outputs up to ~1e7: planes with calibrated scales through conv -> conv(+residual), against torch's fp64 convolution"""
    import torch
    import torch.nn.functional as F
    from tracklab_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(9)
    n, c, h, w = 64, 128, 24, 8
    x = _cl(torch.randn(n, c, h, w, device="cuda", generator=g) * 1e5)
    w1 = _cl(torch.randn(256, c, 3, 3, device="cuda", generator=g) * 0.2)
    w2 = _cl(torch.randn(c, 256, 1, 1, device="cuda", generator=g) * 0.2)
    b1, b2 = torch.randn(256, device="cuda", generator=g) * 1e4, torch.randn(c, device="cuda", generator=g) * 1e4
    states = torch.zeros((3, 2), device="cuda"); states[:, 0] = 1.0
    changed = torch.zeros(1, dtype=torch.int32, device="cuda")
    w1p, w2p = _lib.split_planes(w1), _lib.split_planes(w2)

    def run():
        xh, xl = _lib.split_planes(x, state=states[0])
        y1 = _lib.conv2d_nhwc_16(xh, w1p[0], b1, "relu", x_lo=xl, weight_lo=w1p[1], in_scale=states[0], out_state=states[1])
        y2 = _lib.conv2d_nhwc_16(y1[0], w2p[0], b2, None, xh, x_lo=y1[1], weight_lo=w2p[1], residual_lo=xl, in_scale=states[1], res_scale=states[0], out_state=states[2])
        return _lib.merge_planes(y2[0], y2[1], scale=states[2])
    passes = 0
    for _ in range(6):
        changed.zero_()
        y = run()
        _lib.split_scale_update(states, changed)
        passes += 1
        if int(changed.item()) == 0:
            break
    assert 2 <= passes <= 4, passes                                    # one pass per saturating layer, then a clean one
    sc = states[:, 0].cpu().numpy()
    assert np.all(sc > 1) and np.all(np.log2(sc) == np.round(np.log2(sc)))        # powers of two, every tensor here is beyond 65504
    assert bool(torch.isfinite(y).all())
    x64, w164, w264 = x.double(), w1.double(), w2.double()
    t = F.relu(F.conv2d(x64, w164, b1.double(), padding=1))
    ref = F.conv2d(t, w264, b2.double()) + x64
    bound = F.conv2d(F.conv2d(x64.abs(), w164.abs(), b1.abs().double(), padding=1), w264.abs(), b2.abs().double()) + x64.abs()
    err = (y.double() - ref).abs()
    assert float((err / bound).max()) <= 2e-6, float((err / bound).max())
    assert float(ref.abs().max()) > 1e6                                 # the experiment really left float16's range


def test_scale_update_rule():
    import torch
    from tracklab_amd import _lib
    f = lambda v: torch.tensor(v, dtype=torch.float32, device="cuda")      # noqa: E731
    #                  scale, recorded max
    st = f([[1.0, 100.0],          # fits: stays 1
            [1.0, 16384.0],        # 2^14 exactly: frexp gives 2^15 * 0.5 -> needs 2
            [1.0, 3.0e5],          # grows to 32 (3e5 / 32 = 9375 <= 16384)
            [64.0, 3.0e5],         # needs 32: within the hysteresis band, stays 64
            [1024.0, 3.0e5],       # needs 32, 32 * 8 <= 1024: shrinks to 64
            [8.0, 0.0],            # nothing recorded (no live rows): unchanged
            [4.0, float("inf")],   # overflow upstream: unchanged, counted
            [1.0, 16383.0]])       # just below 2^14: stays 1
    ch = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.split_scale_update(st, ch)
    got = st.cpu().numpy()
    np.testing.assert_array_equal(got[:, 0], [1, 2, 32, 64, 64, 8, 4, 1])
    np.testing.assert_array_equal(got[:, 1], 0)
    assert int(ch.item()) == 3                                          # two grew, one was not finite


def test_detector_in_split_mode_is_fp32_class():
    """YOLOX with every convolution in split-precision mode ((hi, lo) planes through the CSP slice-concatenations, the
    up-sampling, the SPP block and the decoupled head) against the same network on the exact-fp32 kernels"""
    import importlib
    import torch
    ymod = importlib.import_module("tracklab_amd.backbones.yolox")
    for size in ("s", "m"):
        net = ymod.yolox(size, 1, device="cuda", dtype=torch.float32)
        x = (torch.rand(3, 12, 160, 160, device="cuda") * 255).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            a = net(x, focused=True)
            b = net(x, focused=True, split=True)
        assert a.shape == b.shape and b.dtype == torch.float32 and bool(torch.isfinite(b).all())
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(a.abs().max())), (size, float((a - b).abs().max()), float(a.abs().max()))


def test_pipeline_with_both_networks_in_split_mode_keeps_the_oracles_ids(orc):
    import torch
    from tracklab_amd import gpu_pipeline as gp
    from tracklab_amd.synth import SyntheticStream, render_frame, synth_yolox_head
    from test_gpu_pipeline_configs import _detector_rows
    F, steps, nobj, maxd = 4, 4, 40, 48
    pipe = gp.DetReidTrackPipeline("m", n_streams=1, frames_per_step=F, max_dets=maxd, use_graph=True, dtype=torch.float32,
                                   reid_split_precision=True, detector_split_precision=True)
    assert pipe.det_split and pipe.reid.split_precision and pipe.check_finite
    rng = np.random.default_rng(12)
    stream = list(SyntheticStream(5, nobj, F * steps))
    heads = np.stack([synth_yolox_head(rng, fr["dets"][:, :4], ratio=pipe.ratio) for fr in stream])
    d_frames = torch.from_numpy(np.stack([render_frame(rng, stream[i]["gt_boxes"]) for i in range(F)])).cuda()
    d_heads = torch.from_numpy(heads).cuda().reshape(steps, F, -1, 6)
    ref = orc.StrongSORT(pipe.K, pipe.D, **pipe.tracker_cfg)
    for k in range(steps):
        h_rows, h_cnt = pipe.step(d_frames, d_heads[k])
        pipe.synchronize()
        rows, _ = pipe.rows_numpy(h_rows, h_cnt)
        emb = pipe.last["emb"].cpu().numpy().reshape(F, maxd, pipe.K, pipe.D)
        vis = pipe.last["vis"].cpu().numpy().reshape(F, maxd, pipe.K)
        for f in range(F):
            ltwh = _detector_rows(orc, heads[k * F + f], pipe.ratio)
            n = len(ltwh)
            exp = ref.update((k * F + f) * maxd + np.arange(n), ltwh.astype(np.float64), emb[f, :n], vis[f, :n], np.ones(n))
            got = rows[0][f]
            assert len(got) == len(exp)
            np.testing.assert_array_equal(got["det_id"], exp["det_id"])
            np.testing.assert_array_equal(got["track_id"], exp["track_id"])
    sc = pipe.reid._split_scales
    assert sc is not None and sc.calibrated and bool((sc.buf[:, 0] == 1).all())      # a random-init network fits float16: every scale stays 1
    pipe.close()
