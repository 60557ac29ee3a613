"""-m gpu: the r06 prediction-head kernels (tlk_yolox_head_nhwc, tlk_reid_part_head) through the C ABI against oracle/src/heads.c -- bit-exact where
no exp is involved, 2e-6 relative behind the device's exp -- and the pipeline-level behaviours they were built for: padding slots of the hand-off
are zero, the non-finite check sees live rows only (an all-empty f16 step used to raise a spurious 'not finite' error)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _nhwc(t_np, dtype):
    """(B, hw_h, hw_w, C) numpy -> logical (B, C, H, W) cuda tensor in channels_last memory"""
    import torch
    return torch.from_numpy(t_np).cuda().to(dtype).permute(0, 3, 1, 2)


@pytest.mark.parametrize("dtype_name", ["float32", "float16"])
@pytest.mark.parametrize("C,ncls,sizes", [(192, 1, ((20, 20), (10, 10), (5, 5))), (64, 3, ((9, 7), (4, 5))), (320, 80, ((8, 8),))],
                         ids=["yolox_m_width_3_levels", "ragged_chunks_3_classes", "yolox_x_width_coco_classes"])
def test_yolox_head_equals_oracle(orc, dtype_name, C, ncls, sizes):
    import torch
    from tracklab_amd import _lib
    dtype = getattr(torch, dtype_name)
    rng = np.random.default_rng(C + ncls)
    B = 3
    cfs, rfs, ws, bs, cfn, rfn = [], [], [], [], [], []
    for (h, w) in sizes:
        c_np, r_np = rng.standard_normal((B, h, w, C)).astype(np.float32), rng.standard_normal((B, h, w, C)).astype(np.float32)
        ct, rt = _nhwc(c_np, dtype), _nhwc(r_np, dtype)
        cfs.append(ct); rfs.append(rt)
        cfn.append(ct.permute(0, 2, 3, 1).float().cpu().numpy().reshape(B, h * w, C)); rfn.append(rt.permute(0, 2, 3, 1).float().cpu().numpy().reshape(B, h * w, C))
        ws.append((rng.standard_normal((5 + ncls, C)) * 0.1).astype(np.float32)); bs.append(rng.standard_normal(5 + ncls).astype(np.float32))
    got = _lib.yolox_head(cfs, rfs, [torch.from_numpy(w).cuda() for w in ws], [torch.from_numpy(b).cuda() for b in bs], ncls).cpu().numpy()
    exp = orc.yolox_head(cfn, rfn, ws, bs, ncls)
    assert got.shape == exp.shape == (B, sum(h * w for h, w in sizes), 5 + ncls)
    np.testing.assert_array_equal(got[..., :4], exp[..., :4])                       # one fmaf chain + bias: the same bits
    np.testing.assert_allclose(got[..., 4:], exp[..., 4:], rtol=2e-6, atol=1e-7)    # behind the device's exp
    # and the head means what the library route computed: 1 x 1 convolutions + sigmoid (fp32 reference, summation order free: a tolerance)
    off = 0
    for cf, rf, w, b, (h, wd) in zip(cfn, rfn, ws, bs, sizes):
        ref_reg = rf.astype(np.float64) @ w[:5].T.astype(np.float64) + b[:5]
        ref_cls = cf.astype(np.float64) @ w[5:].T.astype(np.float64) + b[5:]
        ref = np.concatenate([ref_reg[..., :4], 1 / (1 + np.exp(-ref_reg[..., 4:5])), 1 / (1 + np.exp(-ref_cls))], -1)
        np.testing.assert_allclose(got[:, off:off + h * wd], ref, rtol=2e-5, atol=2e-5)
        off += h * wd


def test_yolox_network_head_route_agrees_with_the_library_route():
    """the whole detector with the fused head against itself on the library head (TLK_HEADS=0 route: nn.Conv2d + sigmoid + cat), fp32"""
    import importlib
    import torch
    ymod = importlib.import_module("tracklab_amd.backbones.yolox")       # (the package re-exports a function of the same name)
    net = ymod.yolox("s", 1, device="cuda", dtype=torch.float32)
    x = (torch.rand(2, 12, 160, 160, device="cuda") * 255).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        a = net(x, focused=True)
        ymod.USE_TLK_HEADS = False
        try:
            b = net(x, focused=True)
        finally:
            ymod.USE_TLK_HEADS = True
    assert a.shape == b.shape and a.dtype == torch.float32
    assert float((a - b).abs().max()) <= 2e-4 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("dtype_name", ["float32", "float16"])
@pytest.mark.parametrize("D,K,h,w", [(256, 6, 24, 8), (512, 6, 24, 8), (64, 1, 7, 7), (256, 8, 10, 13)], ids=["bpbreid_256", "kpr_512", "one_part_ragged", "eight_parts_ragged"])
def test_reid_part_head_equals_oracle(orc, dtype_name, D, K, h, w):
    import torch
    from tracklab_amd import _lib
    dtype = getattr(torch, dtype_name)
    rng = np.random.default_rng(D + K)
    N = 7
    f_np = rng.standard_normal((N, h, w, D)).astype(np.float32)
    ft = _nhwc(f_np, dtype)
    fn = ft.permute(0, 2, 3, 1).float().cpu().numpy().reshape(N, h * w, D)
    wc, bc = (rng.standard_normal((K, D)) * 0.2).astype(np.float32), rng.standard_normal(K).astype(np.float32)
    thr = 0.5 / K
    emb, vis = _lib.reid_part_head(ft, torch.from_numpy(wc).cuda(), torch.from_numpy(bc).cuda(), thr)
    e_emb, e_vis, bad = orc.reid_part_head(fn, wc, bc, thr)
    assert not bad
    np.testing.assert_allclose(emb.cpu().numpy(), e_emb, rtol=3e-6, atol=1e-7)
    np.testing.assert_array_equal(vis.cpu().numpy(), e_vis)
    # torch statement of the same head (what PartBasedReID.head computes off the GPU route)
    f64 = torch.from_numpy(fn).double()
    att = torch.softmax(f64 @ torch.from_numpy(wc).double().T + torch.from_numpy(bc).double(), dim=-1)           # (N, hw, K)
    ref = torch.einsum("npk,npd->nkd", att, f64) / att.sum(1).clamp_min(1e-6)[..., None]
    np.testing.assert_allclose(emb.cpu().numpy(), ref.numpy(), rtol=2e-5, atol=2e-6)


def test_reid_part_head_hand_off_layout_padding_and_flag(orc):
    """dense batch -> (frame, slot) rows; padding rows zero; a NaN in a LIVE row sets the flag, one in a row nobody reads does not"""
    import torch
    from tracklab_amd import _lib
    rng = np.random.default_rng(3)
    frames, maxd, D, K, h, w = 4, 5, 256, 6, 6, 4
    counts = np.array([3, 0, 5, 1], np.int32)
    base = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int32)
    ncap = frames * maxd
    f_np = rng.standard_normal((ncap, h, w, D)).astype(np.float32)
    f_np[int(counts.sum()):] = np.nan                                # rows beyond the live total: never convolved in the pipeline -- garbage by contract
    wc, bc = (rng.standard_normal((K, D)) * 0.2).astype(np.float32), rng.standard_normal(K).astype(np.float32)
    ft = _nhwc(f_np, torch.float32)
    flag = torch.zeros(1, dtype=torch.bool, device="cuda")
    emb = torch.full((frames, maxd, K, D), 7.0, device="cuda")
    vis = torch.full((frames, maxd, K), 9, dtype=torch.uint8, device="cuda")
    args = (torch.from_numpy(wc).cuda(), torch.from_numpy(bc).cuda(), 0.5 / K, torch.from_numpy(counts).cuda(), torch.from_numpy(base).cuda(), maxd)
    _lib.reid_part_head(ft, *args, out_emb=emb, out_vis=vis, flag=flag)
    e_emb, e_vis, bad = orc.reid_part_head(f_np.reshape(ncap, h * w, D), wc, bc, 0.5 / K, counts, base, maxd)
    assert not bad and not bool(flag.item())
    np.testing.assert_allclose(emb.cpu().numpy().reshape(ncap, K, D), e_emb, rtol=3e-6, atol=1e-7)
    np.testing.assert_array_equal(vis.cpu().numpy().reshape(ncap, K), e_vis)
    g = emb.cpu().numpy()
    for b in range(frames):
        assert np.all(g[b, counts[b]:] == 0) and np.all(vis.cpu().numpy()[b, counts[b]:] == 0)
        assert np.all(np.abs(g[b, :counts[b]]).reshape(counts[b], -1).max(-1) > 0) if counts[b] else True
    # slot layout without a dense batch (TLK_DENSE_REID=0): row r reads feature row r
    f2 = rng.standard_normal((ncap, h, w, D)).astype(np.float32)
    emb2, vis2 = _lib.reid_part_head(_nhwc(f2, torch.float32), args[0], args[1], args[2], args[3], None, maxd)
    e2, v2, _ = orc.reid_part_head(f2.reshape(ncap, h * w, D), wc, bc, 0.5 / K, counts, None, maxd)
    np.testing.assert_allclose(emb2.cpu().numpy(), e2, rtol=3e-6, atol=1e-7)
    np.testing.assert_array_equal(vis2.cpu().numpy(), v2)
    # a non-finite LIVE row raises the flag
    f_np[1, 2, 1, 17] = np.inf
    _lib.reid_part_head(_nhwc(f_np, torch.float32), *args, out_emb=emb, out_vis=vis, flag=flag)
    assert bool(flag.item())
    assert orc.reid_part_head(f_np.reshape(ncap, h * w, D), wc, bc, 0.5 / K, counts, base, maxd)[2]


@pytest.mark.parametrize("arch", ["resnet50", "hrnet32"])
def test_part_based_reid_fused_head_agrees_with_the_torch_head(arch):
    import importlib
    import torch
    rmod = importlib.import_module("tracklab_amd.backbones.reid")
    net = rmod.part_based_reid(6, 256, device="cuda", dtype=torch.float32, arch=arch)
    x = torch.randn(3, 3, 384, 128, device="cuda").contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        f = net.features(x)
        emb, vis = net.head(f)
        rmod.USE_TLK_HEADS = False
        try:
            emb_t, vis_t = net.head(f)
        finally:
            rmod.USE_TLK_HEADS = True
    assert emb.shape == emb_t.shape == (3, 6, 256) and vis.dtype == torch.bool and vis.shape == vis_t.shape
    assert float((emb - emb_t).abs().max()) <= 2e-5 * max(1.0, float(emb_t.abs().max()))
    assert bool((vis == vis_t).all())


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "graph"])
@pytest.mark.parametrize("dtype_name,split", [("float16", False), ("float32", True)], ids=["f16", "split"])
def test_a_step_without_detections_is_not_a_precision_error(use_graph, dtype_name, split):
    """with the dense ReID batch a step whose frames have NO detections convolves nothing, so every row of the network's
    output is stale / uninitialised memory; the non-finite check must look at live rows only.  The feature buffers are poisoned with NaN first."""
    import torch
    from tracklab_amd import gpu_pipeline as gp
    from tracklab_amd.synth import SyntheticStream, render_frame, synth_yolox_head
    F, maxd = 2, 16
    pipe = gp.DetReidTrackPipeline("s", n_streams=1, frames_per_step=F, max_dets=maxd, use_graph=use_graph, dtype=getattr(torch, dtype_name),
                                   reid_split_precision=split)
    assert pipe.check_finite and pipe.dense_reid
    rng = np.random.default_rng(5)
    stream = list(SyntheticStream(3, 10, F))
    frames = torch.from_numpy(np.stack([render_frame(rng, fr["gt_boxes"]) for fr in stream])).cuda()
    full = torch.from_numpy(np.stack([synth_yolox_head(rng, fr["dets"][:, :4], ratio=pipe.ratio) for fr in stream])).cuda()
    empty = torch.zeros_like(full)                                    # no anchor passes the score threshold
    # a normal step first (graphs captured, every buffer written once), then poison what an all-empty step will NOT overwrite
    pipe.step(frames, full); pipe.synchronize()
    assert int(pipe.last["counts"].sum()) > 0
    for st_ in pipe.sets:                                             # (auto stage overlap at two frames per step: the crops exist twice)
        st_["crops"].fill_(float("nan"))
    for b_ in pipe.bufs:
        b_["emb"].fill_(float("nan"))
    pipe.step(frames, empty)
    pipe.synchronize()                                                # must not raise
    assert int(pipe.last["counts"].sum()) == 0
    assert bool((pipe.last["emb"] == 0).all())                        # padding slots are zero-filled by the head kernel
    # and the error still fires when a LIVE embedding is not finite (the head's cached classifier weight poisoned: the head launch is eager)
    pipe.reid._head_w[1].fill_(float("nan"))
    pipe.step(frames, full)
    with pytest.raises(Exception, match="not finite"):
        pipe.synchronize()
    pipe.close()


def test_f16_focus_stem_on_libtlk_agrees_with_the_library_route():
    """r06: the f16 Focus stem (12 channels = one and a half 16-byte groups) zero-padded to 16 channels on tlk_conv2d_nhwc_16 -- the detector's last
    library convolution -- against the library route (TLK_FOCUS16=0: CK / MIOpen convolution + epilogue pass)"""
    import importlib
    import torch
    ymod = importlib.import_module("tracklab_amd.backbones.yolox")
    net = ymod.yolox("m", 1, device="cuda", dtype=torch.float16)
    x = (torch.rand(2, 12, 160, 160, device="cuda") * 255).half().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        a = net.backbone.stem(x, True)
        ymod.USE_TLK_FOCUS16 = False
        try:
            b = net.backbone.stem(x, True)
        finally:
            ymod.USE_TLK_FOCUS16 = True
    assert a.shape == b.shape == (2, 48, 160, 160) and a.dtype == torch.float16
    assert float((a.float() - b.float()).abs().max()) <= 2.0 ** -9 * max(1.0, float(b.float().abs().max()))
