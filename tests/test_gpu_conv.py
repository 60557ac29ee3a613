"""tlk_conv2d_nhwc_f32 (csrc/tlk_conv.hip): the fp32 MFMA convolution with fused epilogue.
  * bit-exact against oracle/src/conv.c (same fmaf chain) for every tile configuration, ragged shapes, strides, taps outside the image,
    residual, channel-sliced input / output;
  * within fp32 round-off of torch's own convolution (the reference's backbones are third-party fp32 networks: tolerance, stated here);
  * the fp32 ReID / detector forward through the kernel == the same modules on torch's library route, to that tolerance."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # n, h, w, cin, cout, k, stride, act, residual
    (2, 9, 7, 8, 5, 3, 1, "relu", False),
    (1, 12, 10, 4, 64, 7, 2, "relu", False),        # the ResNet stem with the input padded to 4 channels
    (3, 8, 8, 64, 64, 1, 1, "relu", True),
    (2, 11, 5, 32, 96, 3, 2, None, True),
    (1, 6, 6, 36, 130, 3, 1, "relu", False),        # K = 324: not a multiple of 32; Cout ragged
    (2, 5, 9, 128, 256, 1, 2, None, False),         # strided 1x1 (downsample branch)
    (1, 16, 16, 12, 48, 3, 1, "none", False),       # YOLOX Focus stem shape
    (5, 4, 3, 8, 33, 5, 1, "relu", True),
]


def _run(case, cfg):
    import oracle
    from tracklab_amd import _lib
    n, h, w, cin, cout, k, s, act, res = case
    rng = np.random.default_rng(hash(case[:7]) & 0xffff)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((cout, k, k, cin)) * 0.1).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    exp = oracle.conv2d_nhwc_f32(x, wt, b, None, stride=s, act=act)
    r = rng.standard_normal(exp.shape).astype(np.float32) if res else None
    if res:
        exp = oracle.conv2d_nhwc_f32(x, wt, b, r, stride=s, act=act)
    xt = torch.from_numpy(x).cuda().permute(0, 3, 1, 2)
    wtt = torch.from_numpy(wt).cuda().permute(0, 3, 1, 2)
    rt = torch.from_numpy(r).cuda().permute(0, 3, 1, 2) if res else None
    _lib.lib(); _lib.conv2d_nhwc_f32(xt[:0], wtt, None)          # binds the symbols
    _lib.check(_lib.lib().tlk_conv2d_set_config(cfg))
    try:
        y = _lib.conv2d_nhwc_f32(xt, wtt, torch.from_numpy(b).cuda(), act, rt, stride=s)
    finally:
        _lib.lib().tlk_conv2d_set_config(-1)
    return y.permute(0, 2, 3, 1).cpu().numpy(), exp


# 0-6: two-stage tiles of conv_f32_mfma_kernel; 7-9: its one-stage tiles (r05); 21-26: the direct-to-LDS kernels of tlk_conv16x.hip on fp32 tensors
# (r05, MODE_F32: they need Cin % 32 == 0 and say so otherwise)
@pytest.mark.parametrize("cfg", [-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
@pytest.mark.parametrize("case", CASES)
def test_conv_is_bit_exact_with_the_oracle_chain(case, cfg):
    got, exp = _run(case, cfg)
    assert got.shape == exp.shape
    assert np.array_equal(got, exp), f"max |diff| {np.abs(got - exp).max()}"


@pytest.mark.parametrize("live", [0, 1, 5, 12, 13, 40])
def test_dynamic_batch_computes_the_live_images_only(live):
    """r05 (dense ReID batch): with tlk_conv_set_dynamic_batch the kernels take the image count from device memory when they RUN; the n of the
    call is the capacity.  The live images' outputs are bit-identical to a static launch, nothing beyond them is written, and a count beyond
    the capacity is clamped to it.  fp32 kernel and both 16-bit modes."""
    from tracklab_amd import _lib
    torch.manual_seed(live)
    n, h, w, cin, cout = 13, 12, 9, 64, 128
    x = torch.randn(n, cin, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, 3, 3, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device="cuda")
    r = torch.randn(n, cout, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    full = _lib.conv2d_nhwc_f32(x, wt, b, "relu", r, stride=1, pad=1)
    n_live = torch.tensor([live], dtype=torch.int32, device="cuda")
    out = torch.full_like(full, -3.0)
    _lib.conv_set_dynamic_batch(n_live)
    try:
        _lib.conv2d_nhwc_f32(x, wt, b, "relu", r, stride=1, pad=1, out=out)
    finally:
        _lib.conv_set_dynamic_batch(None)
    torch.cuda.synchronize()
    k = min(live, n)
    assert torch.equal(out[:k], full[:k])
    assert (out[k:] == -3.0).all()
    # 16-bit kernels (f16 mode): same contract
    xh, wh = x.half(), wt.half()
    full16 = _lib.conv2d_nhwc_16(xh, wh, b, "relu", r.half(), stride=1, pad=1)
    out16 = torch.full_like(full16, -3.0)
    _lib.conv_set_dynamic_batch(n_live)
    try:
        _lib.conv2d_nhwc_16(xh, wh, b, "relu", r.half(), stride=1, pad=1, out=out16)
    finally:
        _lib.conv_set_dynamic_batch(None)
    torch.cuda.synchronize()
    assert torch.equal(out16[:k], full16[:k])
    assert (out16[k:] == -3.0).all()


X32_CASES = [
    # n, h, w, cin, cout, k, stride, act, residual  (Cin a multiple of the 32-float K step)
    (3, 8, 8, 64, 64, 1, 1, "relu", True),
    (2, 11, 5, 32, 96, 3, 2, None, True),
    (2, 5, 9, 128, 256, 1, 2, None, False),
    (5, 13, 7, 64, 260, 1, 1, "relu", True),        # Cout ragged against every tile width
    (1, 24, 8, 256, 128, 3, 1, "relu", False),
]


@pytest.mark.parametrize("cfg", [21, 22, 23, 24, 25, 26])
@pytest.mark.parametrize("case", X32_CASES)
def test_direct_to_lds_fp32_kernels_are_bit_exact_with_the_oracle_chain(case, cfg):
    got, exp = _run(case, cfg)
    assert got.shape == exp.shape
    assert np.array_equal(got, exp), f"max |diff| {np.abs(got - exp).max()}"


@pytest.mark.parametrize("cfg", [27, 28, 29])
@pytest.mark.parametrize("case", X32_CASES[:3] + [(3, 9, 9, 64, 32, 3, 1, "relu", True)])
def test_32_wide_direct_to_lds_fp32_tiles_are_bit_exact_with_the_oracle_chain(case, cfg):
    got, exp = _run(case, cfg)
    assert got.shape == exp.shape
    assert np.array_equal(got, exp), f"max |diff| {np.abs(got - exp).max()}"


PATCH_CASES = [
    # 3 x 3 / stride 1 / pad 1 on 32 channels, whole image rows per 256- (128-) pixel tile: Wo = 32, 16, 64, 8; with / without residual
    (3, 16, 32, 32, 32, 3, 1, "relu", True),
    (2, 32, 16, 32, 64, 3, 1, None, False),
    (2, 8, 64, 32, 32, 3, 1, "relu", False),
    (1, 64, 8, 32, 36, 3, 1, "relu", True),           # Cout ragged against the 32-wide tile
]


@pytest.mark.parametrize("cfg", [30, 31, 32, 33])
@pytest.mark.parametrize("case", PATCH_CASES)
def test_patch_resident_3x3_kernel_is_bit_exact_with_the_oracle_chain(case, cfg):
    """r05, conv16x_kernel<..., PATCH = true>: the tile's input rows + halo land in LDS once and the nine taps are nine shifted fragment
    reads; the k order (kh, kw, ci) and the fmaf chain are the implicit GEMM's, so the bits are oracle/src/conv.c's"""
    got, exp = _run(case, cfg)
    assert got.shape == exp.shape
    assert np.array_equal(got, exp), f"max |diff| {np.abs(got - exp).max()}"


PATCH64_CASES = [
    # r06: 3 x 3 / stride 1 / pad 1 on 64 channels (two K steps per pixel: two patch regions): ResNet's layer 1 (Wo = 32) and HRNet's 64-channel branch (Wo = 16)
    (3, 16, 32, 64, 64, 3, 1, "relu", False),
    (2, 48, 16, 64, 64, 3, 1, "relu", True),
    (2, 8, 64, 64, 72, 3, 1, None, False),            # Cout ragged against the 64-wide tile
    (1, 64, 8, 64, 128, 3, 1, "relu", True),
]


@pytest.mark.parametrize("cfg", [34, 35, 36, 37])
@pytest.mark.parametrize("case", PATCH64_CASES)
def test_patch_resident_3x3_kernel_on_64_channels_is_bit_exact_with_the_oracle_chain(case, cfg):
    got, exp = _run(case, cfg)
    assert got.shape == exp.shape
    assert np.array_equal(got, exp), f"max |diff| {np.abs(got - exp).max()}"


def test_patch_kernel_refuses_a_shape_outside_its_contract_and_the_heuristic_routes_the_32_channel_layers_to_it():
    from tracklab_amd import _lib
    L = _lib.lib()
    x = torch.randn(2, 24, 20, 32, device="cuda").permute(0, 3, 1, 2)          # Wo = 20: 256 % 20 != 0
    w = (torch.randn(32, 3, 3, 32, device="cuda") * 0.1).permute(0, 3, 1, 2)
    assert L.tlk_conv2d_set_config(31) == 0
    try:
        with pytest.raises(_lib.TlkError, match="patch kernel"):
            _lib.conv2d_nhwc_f32(x, w, None, "relu", None, stride=1)
    finally:
        L.tlk_conv2d_set_config(-1)
    x = torch.randn(32, 96, 32, 32, device="cuda").permute(0, 3, 1, 2)          # HRNet-W32's high-resolution branch: 32 crops x 96 x 32 = 98304 pixels
    y = _lib.conv2d_nhwc_f32(x, w, None, "relu", None, stride=1)
    assert L.tlk_conv2d_last_config() == 31
    L.tlk_conv2d_set_config(0)
    try:
        y0 = _lib.conv2d_nhwc_f32(x, w, None, "relu", None, stride=1)
    finally:
        L.tlk_conv2d_set_config(-1)
    assert torch.equal(y, y0)


def test_heuristic_routes_the_64_channel_3x3_layers_to_the_patch_kernel():
    """r06: ResNet's layer 1 / HRNet's 64-channel branch (3 x 3, stride 1, 64 -> 64, whole image rows per 128-pixel tile, a launch of >= 64K pixels)"""
    from tracklab_amd import _lib
    L = _lib.lib()
    x = torch.randn(32, 96, 32, 64, device="cuda").permute(0, 3, 1, 2)
    w = (torch.randn(64, 3, 3, 64, device="cuda") * 0.05).permute(0, 3, 1, 2)
    r = torch.randn(32, 96, 32, 64, device="cuda").permute(0, 3, 1, 2)
    y = _lib.conv2d_nhwc_f32(x, w, None, "relu", r, stride=1)
    assert L.tlk_conv2d_last_config() == 34
    L.tlk_conv2d_set_config(0)
    try:
        y0 = _lib.conv2d_nhwc_f32(x, w, None, "relu", r, stride=1)
    finally:
        L.tlk_conv2d_set_config(-1)
    assert torch.equal(y, y0)


# (n, h, w, cin, cout, k, stride, residual) at sizes where the r06 rules apply -> the configuration the heuristic must pick
R06_ROUTES = [
    ((24, 80, 80, 96, 96, 1, 1, False), 27),        # 96 wide: three 256 x 32 direct-to-LDS tiles (YOLOX-m / CSPNeXt-m)
    ((24, 80, 80, 96, 96, 3, 1, True), 27),
    ((100, 24, 8, 256, 256, 3, 1, False), 25),      # powers of two on few tiles, a small launch: 64 x 128 tiles (one frame's crops)
    ((24, 40, 40, 192, 384, 3, 2, False), 24),      # 3 x 128 wide on few tiles: the two-stage 128 x 128 tile
    ((700, 24, 8, 256, 256, 3, 1, False), 24),      # K >= 1152 on >= 1024 tiles, no residual (ResNet's layer 3 / 4 3 x 3)
    ((90, 96, 32, 64, 256, 1, 1, False), 7),        # K <= 64 projection onto a 128-multiple width: 128 x 128, one stage
    ((24, 160, 160, 48, 48, 1, 1, False), 8),       # short K on the 128 x 64 tile: one stage
    ((40, 96, 32, 256, 32, 3, 1, False), 28),       # 32 wide with a long K loop: 256 x 32, two stages (HRNet's fuse layers)
    ((700, 24, 8, 128, 64, 1, 1, False), 8),        # 64 wide 1 x 1 below a million pixels: back on the register-staged tile (one stage: K <= 192)
]


@pytest.mark.parametrize("case,want", R06_ROUTES)
def test_r06_heuristic_routes_and_every_route_gives_the_generic_kernels_bits(case, want):
    """the tile heuristic re-derived from tools/sweep_conv_f32.py (profiles/r06_conv_f32_sweep.txt): the configuration it picks for the shapes the
    rules were written for, and -- the contract of every configuration -- the same bits as the 128 x 128 register-staged kernel"""
    from tracklab_amd import _lib
    L = _lib.lib()
    n, h, w, cin, cout, k, s, res = case
    g = torch.Generator(device="cuda").manual_seed(cin + cout + k)
    x = torch.randn((n, h, w, cin), device="cuda", generator=g).permute(0, 3, 1, 2)
    wt = (torch.randn((cout, k, k, cin), device="cuda", generator=g) * 0.05).permute(0, 3, 1, 2)
    b = torch.randn(cout, device="cuda", generator=g)
    y0 = None
    L.tlk_conv2d_set_config(0 if cout % 128 == 0 else 2)
    try:
        y0 = _lib.conv2d_nhwc_f32(x, wt, b, "relu", None, stride=s)
        r = torch.randn(y0.shape, device="cuda", generator=g).contiguous(memory_format=torch.channels_last) if res else None
        if res:
            y0 = _lib.conv2d_nhwc_f32(x, wt, b, "relu", r, stride=s)
    finally:
        L.tlk_conv2d_set_config(-1)
    y = _lib.conv2d_nhwc_f32(x, wt, b, "relu", r, stride=s)
    assert L.tlk_conv2d_last_config() == want, (case, L.tlk_conv2d_last_config())
    assert torch.equal(y, y0)


@pytest.mark.parametrize("case", [(2, 24, 16, 3, 64, 7, 2, "relu", False), (3, 17, 13, 3, 32, 3, 2, "relu", False), (1, 40, 130, 3, 48, 3, 2, None, False),
                                  (2, 9, 70, 3, 64, 7, 2, "relu", False)])
def test_direct_rgb_stem_kernel_is_bit_exact_with_the_oracle_on_the_padded_problem(case):
    """r05, csrc/tlk_conv_stem.hip: Cin = 3 goes to the direct stem kernel; its chain is the implicit-GEMM kernel's on the 4-channel-padded
    problem minus the zero terms, which is what oracle/src/conv.c computes on the padded input -- bit for bit (ragged widths beyond one
    64-column strip, rows beyond a 4-row strip pair, Cout below a 32-wide tile)"""
    import oracle
    from tracklab_amd import _lib
    n, h, w, cin, cout, k, s, act, _ = case
    rng = np.random.default_rng(k * 100 + cout)
    x = rng.standard_normal((n, h, w, 3)).astype(np.float32)
    wt = (rng.standard_normal((cout, k, k, 3)) * 0.1).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    x4 = np.concatenate([x, np.zeros((n, h, w, 1), np.float32)], -1)
    w4 = np.concatenate([wt, np.zeros((cout, k, k, 1), np.float32)], -1)
    exp = oracle.conv2d_nhwc_f32(x4, w4, b, None, stride=s, act=act)
    xt = torch.from_numpy(x).cuda().permute(0, 3, 1, 2)
    wtt = torch.from_numpy(wt).cuda().permute(0, 3, 1, 2)
    y = _lib.conv2d_nhwc_f32(xt, wtt, torch.from_numpy(b).cuda(), act, None, stride=s)
    assert _lib.lib().tlk_conv2d_last_config() == 15, "the call did not reach the direct stem kernel"
    got = y.permute(0, 2, 3, 1).cpu().numpy()
    assert got.shape == exp.shape and np.array_equal(got, exp), f"max |diff| {np.abs(got - exp).max()}"


def test_residual_after_the_activation_is_bit_exact_with_the_oracle():
    """TLK_ACT_RES_AFTER: y = act(conv + bias) + r -- CSPNeXt's identity add riding in the pointwise convolution's epilogue"""
    import oracle
    from tracklab_amd import _lib
    rng = np.random.default_rng(9)
    for n, h, w, cin, cout, k in ((2, 9, 7, 48, 48, 1), (1, 6, 5, 96, 96, 1), (2, 5, 5, 16, 130, 3)):
        x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
        wt = (rng.standard_normal((cout, k, k, cin)) * 0.1).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        r = rng.standard_normal((n, h, w, cout)).astype(np.float32)
        exp = oracle.conv2d_nhwc_f32(x, wt, b, r, act="relu", res_after_act=True)
        assert np.array_equal(exp, np.maximum(oracle.conv2d_nhwc_f32(x, wt, b, None), 0) + r)
        got = _lib.conv2d_nhwc_f32(torch.from_numpy(x).cuda().permute(0, 3, 1, 2), torch.from_numpy(wt).cuda().permute(0, 3, 1, 2), torch.from_numpy(b).cuda(),
                                   "relu", torch.from_numpy(r).cuda().permute(0, 3, 1, 2), residual_after_act=True)
        assert np.array_equal(got.permute(0, 2, 3, 1).cpu().numpy(), exp)


def test_silu_within_exp_roundoff():
    got, exp = _run((2, 10, 10, 16, 40, 3, 1, "silu", False), -1)
    np.testing.assert_allclose(got, exp, rtol=2e-6, atol=1e-6)      # device exp vs libm expf


def test_channel_slices_in_and_out():
    """reads a channel slice of a wider tensor and writes into a slice of a concatenation buffer (pixel strides)"""
    import oracle
    from tracklab_amd import _lib
    rng = np.random.default_rng(3)
    wide = torch.from_numpy(rng.standard_normal((2, 7, 6, 24)).astype(np.float32)).cuda().permute(0, 3, 1, 2)
    x = wide[:, 8:16]
    wt = (rng.standard_normal((12, 3, 3, 8)) * 0.1).astype(np.float32)
    cat = torch.zeros((2, 7, 6, 20), device="cuda").permute(0, 3, 1, 2)
    _lib.conv2d_nhwc_f32(x, torch.from_numpy(wt).cuda().permute(0, 3, 1, 2), None, "relu", out=cat[:, 4:16])
    exp = oracle.conv2d_nhwc_f32(x.permute(0, 2, 3, 1).cpu().numpy(), wt, None, None, act="relu")
    got = cat.permute(0, 2, 3, 1).cpu().numpy()
    assert np.array_equal(got[..., 4:16], exp) and not got[..., :4].any() and not got[..., 16:].any()


@pytest.mark.parametrize("shape", [(64, 64, 96, 32, 64, 3, 1), (64, 256, 48, 16, 128, 1, 1), (32, 512, 24, 8, 512, 3, 1), (4, 96, 80, 80, 96, 3, 2)])
def test_conv_within_fp32_roundoff_of_torch(shape):
    """tolerance vs torch's fp64 convolution: |err| <= 2e-6 * (|x| conv |w|) -- a few fp32 ulps of the absolute-value sum, K up to 4608"""
    from tracklab_amd import _lib
    n, cin, h, w, cout, k, s = shape
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((n, cin, h, w), device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn((cout, cin, k, k), device="cuda", generator=g) * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device="cuda", generator=g)
    y = _lib.conv2d_nhwc_f32(x, wt, b, "relu", stride=s)
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), s, k // 2))
    bound = F.conv2d(x.double().abs(), wt.double().abs(), b.double().abs(), s, k // 2)
    assert y.shape == ref.shape
    assert bool(((y.double() - ref).abs() <= 2e-6 * bound + 1e-30).all()), float(((y.double() - ref).abs() / bound).max())


def test_fp32_networks_on_the_kernel_match_the_library_route():
    """ReID ResNet-50 and YOLOX-s forward in fp32: hand-written convolutions vs torch / MIOpen convolutions, same weights"""
    from tracklab_amd.backbones import common
    from tracklab_amd.backbones.reid import part_based_reid
    from tracklab_amd.backbones.yolox import yolox
    g = torch.Generator(device="cuda").manual_seed(2)
    for net, x in ((part_based_reid(dtype=torch.float32), torch.randn((6, 4, 384, 128), device="cuda", generator=g)[:, :3]),
                   (yolox("s", dtype=torch.float32), torch.rand((2, 3, 640, 640), device="cuda", generator=g) * 255)):
        x = x.contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            got = net(x)
            common.USE_TLK_CONV_F32 = False
            try:
                exp = net(x)
            finally:
                common.USE_TLK_CONV_F32 = True
        got, exp = (got[0], exp[0]) if isinstance(got, tuple) else (got, exp)
        scale = float(exp.abs().max())
        assert float((got - exp).abs().max()) <= 2e-4 * scale, (float((got - exp).abs().max()), scale)
