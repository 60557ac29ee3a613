"""-m gpu: r06, the split-precision route of HRNet-W32 (the ReID backbone tracklab/configs/modules/reid/bpbreid.yaml:53 selects).
(a) tlk_split_fuse_sum -- the element-wise joint of the exchange units: [relu](sum of plane / fp32 terms at mixed resolutions) as scaled planes --
against the composition of torch passes it replaces (fp32 reference of the same op: the sum in term order, so the only difference is the
2^-22 relative rounding of the planes); (b) the whole network in split mode against the same network on the exact-fp32 kernels."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cl(t):
    import torch
    return t.contiguous(memory_format=torch.channels_last)


def _state(scale=1.0):
    import torch
    return torch.tensor([scale, 0.0], dtype=torch.float32, device="cuda")


def _up(t, s):
    import torch.nn.functional as F
    return t if s == 0 else F.interpolate(t, scale_factor=2 ** s, mode="nearest")


@pytest.mark.parametrize("c,shifts,kinds,relu", [(32, (0, 1, 2), "pff", True), (64, (0, 0, 1, 2), "fpff", True), (256, (0, 0, 0, 0), "fffp", True),
                                                  (128, (0,), "p", False), (32, (0, 3), "pp", False), (8, (1, 0), "fp", True)])
def test_fuse_sum_is_the_torch_composition_to_plane_rounding(c, shifts, kinds, relu):
    import torch
    from tracklab_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(c + len(shifts))
    n, h, w = 5, 48, 16
    terms, ref = [], None
    for s, kd in zip(shifts, kinds):
        t = _cl(torch.randn(n, c, h >> s, w >> s, device="cuda", generator=g) * 7)
        if kd == "p":
            hi, lo = _lib.split_planes(t)
            t = _lib.merge_planes(hi, lo)              # the value the planes hold
            terms.append((hi, lo, None))
        else:
            terms.append(t)
        u = _up(t, s)
        ref = u if ref is None else ref + u
    if relu:
        ref = torch.relu(ref)
    st = _state()
    yh, yl = _lib.split_fuse_sum(terms, relu=relu, out_state=st)
    y = _lib.merge_planes(yh, yl)
    assert y.shape == ref.shape
    err = (y - ref).abs()
    assert float((err - ref.abs() * 2.0 ** -21).max()) <= 1e-9, float(err.max())      # (absolute term: the lo plane of a tiny value is a float16 subnormal)
    assert float(st[0]) == 1.0 and float(st[1]) == float(ref.abs().max())


def test_fuse_sum_scaled_terms_scaled_output_slices_and_dynamic_batch():
    """terms far beyond float16's range (scaled planes), the output scaled too; written into a channel slice of a wider tensor (the concatenation in
    front of the head); only the live images of a dynamic batch are touched and recorded"""
    import torch
    from tracklab_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(3)
    n, c, h, w = 6, 64, 24, 8
    a = _cl(torch.randn(n, c, h, w, device="cuda", generator=g) * 3e5)
    b = _cl(torch.randn(n, c, h // 2, w // 2, device="cuda", generator=g) * 2e6)
    sa, sb, so = _state(64.0), _state(512.0), _state(1024.0)
    ah, al = _lib.split_planes(a, state=sa)
    bh, bl = _lib.split_planes(b, state=sb)
    av, bv = _lib.merge_planes(ah, al, scale=sa), _lib.merge_planes(bh, bl, scale=sb)
    ref = av + _up(bv, 1)
    wide_h = torch.full((n, 160, h, w), 7.0, dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
    wide_l = wide_h.clone()
    live = torch.tensor([4], dtype=torch.int32, device="cuda")
    _lib.conv_set_dynamic_batch(live)
    try:
        _lib.split_fuse_sum([(ah, al, sa), (bh, bl, sb)], out=(wide_h[:, 32:96], wide_l[:, 32:96]), out_state=so, dynamic_batch=True)
    finally:
        _lib.conv_set_dynamic_batch(None)
    torch.cuda.synchronize()
    y = _lib.merge_planes(_cl(wide_h[:, 32:96]), _cl(wide_l[:, 32:96]), scale=so)
    err = (y[:4] - ref[:4]).abs()
    assert float((err - ref[:4].abs() * 2.0 ** -21).max()) <= 1e-9 * 1024, float(err.max())
    assert float(so[1]) == float(ref[:4].abs().max())                     # the maximum of the LIVE images only
    assert bool((wide_h[4:] == 7).all()) and bool((wide_l[4:] == 7).all())            # images beyond the live count: untouched
    assert bool((wide_h[:, :32] == 7).all()) and bool((wide_h[:, 96:] == 7).all())    # channels beside the slice: untouched


def test_fuse_sum_rejects_what_it_cannot_do():
    import torch
    from tracklab_amd import _lib
    t = _cl(torch.zeros(1, 8, 6, 4, device="cuda"))
    u = _cl(torch.zeros(1, 8, 3, 2, device="cuda"))
    _lib.split_fuse_sum([t, u])
    with pytest.raises(AssertionError):
        _lib.split_fuse_sum([t, _cl(torch.zeros(1, 8, 2, 4, device="cuda"))])
    with pytest.raises(_lib.TlkError):
        _lib.split_fuse_sum([_cl(torch.zeros(1, 12, 6, 4, device="cuda"))])       # channels not a multiple of 8


@pytest.mark.parametrize("scales", [True, False])
def test_hrnet32_in_split_mode_is_fp32_class(scales, monkeypatch):
    """the whole part-based ReID network on HRNet-W32: split mode against the exact-fp32 kernels, same weights, same crops"""
    import importlib
    import torch
    rmod = importlib.import_module("tracklab_amd.backbones.reid")
    monkeypatch.setattr(rmod, "USE_SPLIT_SCALES", scales)
    exact = rmod.part_based_reid(6, 256, device="cuda", dtype=torch.float32, arch="hrnet32")
    split = rmod.part_based_reid(6, 256, device="cuda", dtype=torch.float32, arch="hrnet32", split_precision=True)
    # (random-init HRNet on 0..255 pixels leaves float16's range: the unscaled planes get 0..1 pixels)
    x = _cl(torch.rand(6, 3, 384, 128, device="cuda") * (255 if scales else 1))
    with torch.no_grad():
        fa = exact.features(x)
        fb = split.features(x)
        fb2 = split.features(x)                     # steady state (calibrated scales)
        ea, va = exact.head(fa)
        eb, vb = split.head(fb)
    assert fb.dtype == torch.float32 and fa.shape == fb.shape and bool(torch.isfinite(fb).all())
    assert torch.equal(fb, fb2)
    tol = 2e-5 * max(1.0, float(fa.abs().max()))
    assert float((fa - fb).abs().max()) <= tol, (float((fa - fb).abs().max()), float(fa.abs().max()))
    cos = torch.nn.functional.cosine_similarity(ea.double().flatten(1), eb.double().flatten(1), dim=1)
    assert float((1 - cos).max()) <= 1e-6
    assert torch.equal(va, vb)
    assert (getattr(split, "_split_scales", None) is not None) == scales


def test_hrnet32_split_mode_beyond_float16s_range():
    """weights scaled so that the activations leave float16's range: the scaled planes follow (calibration on the first forward), the unscaled
    route would saturate"""
    import importlib
    import torch
    rmod = importlib.import_module("tracklab_amd.backbones.reid")
    exact = rmod.part_based_reid(6, 256, device="cuda", dtype=torch.float32, arch="hrnet32")
    split = rmod.part_based_reid(6, 256, device="cuda", dtype=torch.float32, arch="hrnet32", split_precision=True)
    with torch.no_grad():
        for m in (exact, split):
            m.backbone.stem[0].conv.weight.mul_(4096.0)        # ReLU network, zero biases: every activation behind it scales by 4096
    x = _cl(torch.rand(4, 3, 384, 128, device="cuda") * 255)
    with torch.no_grad():
        fa = exact.features(x)
        fb = split.features(x)
    assert float(fa.abs().max()) > 65504.0
    assert bool(torch.isfinite(fb).all())
    assert float((fa - fb).abs().max()) <= 2e-5 * float(fa.abs().max())
    sc = split._split_scales.buf[:, 0].cpu().numpy()
    assert sc.max() > 1 and np.all(np.log2(sc) == np.round(np.log2(sc)))


def test_pipeline_with_hrnet32_in_split_mode_keeps_the_oracles_ids(orc):
    """the fused step (hipGraph, dense dynamic ReID batch) with the yaml's backbone in split mode and the detector in split mode: the tracker's rows
    equal the oracle chain's on the embeddings the step produced, and those embeddings are the exact-fp32 pipeline's to cos 1e-6"""
    import torch
    from tracklab_amd import gpu_pipeline as gp
    from tracklab_amd.synth import SyntheticStream, render_frame, synth_yolox_head
    from test_gpu_pipeline_configs import _detector_rows
    F, steps, nobj, maxd = 4, 3, 40, 48
    kw = dict(n_streams=1, frames_per_step=F, max_dets=maxd, use_graph=True, dtype=torch.float32, reid_arch="hrnet32")
    pipe = gp.DetReidTrackPipeline("m", reid_split_precision=True, detector_split_precision=True, **kw)
    exact = gp.DetReidTrackPipeline("m", **kw)
    assert pipe.reid.split_precision and pipe.reid.arch == "hrnet32" and pipe.check_finite
    rng = np.random.default_rng(21)
    stream = list(SyntheticStream(8, nobj, F * steps))
    heads = np.stack([synth_yolox_head(rng, fr["dets"][:, :4], ratio=pipe.ratio) for fr in stream])
    d_frames = torch.from_numpy(np.stack([render_frame(rng, stream[i]["gt_boxes"]) for i in range(F)])).cuda()
    d_heads = torch.from_numpy(heads).cuda().reshape(steps, F, -1, 6)
    ref = orc.StrongSORT(pipe.K, pipe.D, **pipe.tracker_cfg)
    for k in range(steps):
        h_rows, h_cnt = pipe.step(d_frames, d_heads[k])
        pipe.synchronize()
        exact.step(d_frames, d_heads[k])
        exact.synchronize()
        rows, _ = pipe.rows_numpy(h_rows, h_cnt)
        emb = pipe.last["emb"].cpu().numpy().reshape(F, maxd, pipe.K, pipe.D)
        vis = pipe.last["vis"].cpu().numpy().reshape(F, maxd, pipe.K)
        emb_x = exact.last["emb"].cpu().numpy().reshape(F, maxd, pipe.K, pipe.D)
        for f in range(F):
            ltwh = _detector_rows(orc, heads[k * F + f], pipe.ratio)
            n = len(ltwh)
            exp = ref.update((k * F + f) * maxd + np.arange(n), ltwh.astype(np.float64), emb[f, :n], vis[f, :n], np.ones(n))
            got = rows[0][f]
            assert len(got) == len(exp)
            np.testing.assert_array_equal(got["det_id"], exp["det_id"])
            np.testing.assert_array_equal(got["track_id"], exp["track_id"])
            a, b = emb[f, :n].reshape(n, -1).astype(np.float64), emb_x[f, :n].reshape(n, -1).astype(np.float64)
            cos = (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
            assert float((1 - cos).max()) <= 1e-6
    sc = pipe.reid._split_scales
    assert sc is not None and sc.calibrated
    pipe.close(); exact.close()


@pytest.mark.parametrize("c,shifts,relu", [(32, (0, 1, 2, 3), True), (64, (0, 0, 1, 2), True), (256, (0, 0, 0, 0), True), (128, (2,), False), (8, (1, 0), False)])
def test_fuse_sum_f32_is_torchs_interpolate_add_relu_bit_for_bit(c, shifts, relu):
    """tlk_fuse_sum_f32 (the joint of the EXACT fp32 route): ((t0 + t1) + t2) + t3 over nearest-up-sampled fp32 tensors, then ReLU == torch's passes"""
    import torch
    from tracklab_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(7 * c + len(shifts))
    n, h, w = 5, 48, 16
    terms = [_cl(torch.randn(n, c, h >> s, w >> s, device="cuda", generator=g) * 3) for s in shifts]
    ref = None
    for t, s in zip(terms, shifts):
        u = _up(t, s)
        ref = u if ref is None else ref + u
    if relu:
        ref = torch.relu(ref)
    if len(shifts) == 1:                                  # one term: up-sampling into a channel slice of a wider tensor (the concatenation)
        wide = torch.full((n, c + 64, h, w), 7.0, device="cuda").contiguous(memory_format=torch.channels_last)
        _lib.fuse_sum_f32(terms, relu=relu, out=wide[:, 32:32 + c])
        assert torch.equal(wide[:, 32:32 + c], ref) and bool((wide[:, :32] == 7).all()) and bool((wide[:, 32 + c:] == 7).all())
        return
    y = _lib.fuse_sum_f32(terms, relu=relu)
    assert y.shape == ref.shape and torch.equal(y, ref)


def test_hrnet32_exact_route_with_fused_joints_equals_the_torch_passes(monkeypatch):
    """the exact-fp32 HRNet-W32 forward with tlk_fuse_sum_f32 for the exchange units and the concatenation == the same forward on torch's
    interpolate / add / relu / cat passes, bit for bit (the exact route's arithmetic is untouched: only HBM passes are saved)"""
    import importlib
    import torch
    rmod = importlib.import_module("tracklab_amd.backbones.reid")
    net = rmod.part_based_reid(6, 256, device="cuda", dtype=torch.float32, arch="hrnet32")
    x = _cl(torch.rand(5, 3, 384, 128, device="cuda"))
    with torch.no_grad():
        monkeypatch.setattr(rmod, "USE_TLK_FUSE32", True)
        a = net.features(x)
        monkeypatch.setattr(rmod, "USE_TLK_FUSE32", False)
        b = net.features(x)
    assert a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize("c,shifts,relu", [(32, (0, 1, 2, 3), True), (64, (0, 0, 1, 2), True), (128, (2,), False)])
def test_fuse_sum_f16_is_torchs_half_precision_chain_bit_for_bit(c, shifts, relu):
    """tlk_fuse_sum_f16: every partial sum rounded to float16, as torch's half `y = y + t` does"""
    import torch
    from tracklab_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(11 * c + len(shifts))
    n, h, w = 5, 48, 16
    terms = [_cl((torch.randn(n, c, h >> s, w >> s, device="cuda", generator=g) * 30).half()) for s in shifts]
    ref = None
    for t, s in zip(terms, shifts):
        u = _up(t, s)
        ref = u if ref is None else ref + u
    if relu:
        ref = torch.relu(ref)
    if len(shifts) == 1:
        wide = torch.full((n, c + 64, h, w), 7.0, device="cuda", dtype=torch.float16).contiguous(memory_format=torch.channels_last)
        _lib.fuse_sum_f16(terms, relu=relu, out=wide[:, 32:32 + c])
        assert torch.equal(wide[:, 32:32 + c], ref) and bool((wide[:, :32] == 7).all()) and bool((wide[:, 32 + c:] == 7).all())
        return
    y = _lib.fuse_sum_f16(terms, relu=relu)
    assert y.dtype == torch.float16 and torch.equal(y, ref)


def test_hrnet32_f16_route_with_fused_joints_equals_the_torch_passes(monkeypatch):
    import importlib
    import torch
    rmod = importlib.import_module("tracklab_amd.backbones.reid")
    net = rmod.part_based_reid(6, 256, device="cuda", dtype=torch.float16, arch="hrnet32")
    x = _cl(torch.rand(5, 3, 384, 128, device="cuda").half())
    with torch.no_grad():
        monkeypatch.setattr(rmod, "USE_TLK_FUSE16", True)
        a = net.features(x)
        monkeypatch.setattr(rmod, "USE_TLK_FUSE16", False)
        b = net.features(x)
    assert a.dtype == torch.float16 and a.shape == b.shape and torch.equal(a, b)
