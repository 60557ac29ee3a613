"""tlk_conv2d_nhwc_16 (csrc/tlk_conv16.hip): convolutions on the 16-bit MFMA.
  * split mode (fp32 values as (hi, lo) f16 pairs, three MFMAs per product pair): within the SAME fp64 bound as the exact-fp32 kernel
    (tests/test_gpu_conv.py: |err| <= 2e-6 * (|x| conv |w|)) -- fp32-class -- on ragged shapes, strides, residuals, K up to 4608;
  * f16 mode: equal to an fp32-accumulated convolution of the f16 operands rounded once to f16 (1 ulp of f16);
  * split / merge are exact inverses to 2^-22;
  * the ReID network in split precision agrees with the exact-fp32 network far below the f16 leg's distance."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(8, 64, 24, 8, 64, 3, 1, True), (4, 8, 40, 24, 64, 7, 2, False), (6, 256, 12, 8, 128, 1, 2, False), (3, 512, 24, 8, 512, 3, 1, True),
          (2, 128, 17, 9, 72, 3, 1, False), (5, 2048, 6, 4, 256, 1, 1, False), (2, 72, 11, 7, 200, 3, 2, True)]


def _inputs(shape, seed=1):
    n, cin, h, w, cout, k, s, res = shape
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn((n, cin, h, w), device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn((cout, cin, k, k), device="cuda", generator=g) * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device="cuda", generator=g)
    ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
    r = torch.randn((n, cout, ho, wo), device="cuda", generator=g).contiguous(memory_format=torch.channels_last) if res else None
    return x, wt, b, r, k, s


def test_split_and_merge_round_trip():
    from tracklab_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(0)
    x = (torch.randn((3, 16, 9, 5), device="cuda", generator=g) * torch.logspace(-4, 3, 16, device="cuda").view(1, -1, 1, 1)).contiguous(memory_format=torch.channels_last)
    hi, lo = _lib.split_planes(x)
    y = _lib.merge_planes(hi, lo)
    # relative 2^-22 while hi is a normal f16 (|x| >= 2^-14); below that hi is subnormal and the pair is exact to an ABSOLUTE 2^-35
    assert bool(((y - x).abs() <= torch.maximum(x.abs() * 2.0 ** -21, torch.full_like(x, 2.0 ** -35))).all())
    hi8, lo8 = _lib.split_planes(x[:, :3], 8)                       # channel slice in, zero-padded channels out
    assert hi8.shape == (3, 8, 9, 5) and not hi8[:, 3:].any() and not lo8[:, 3:].any()
    assert torch.equal(hi8[:, :3], hi[:, :3]) and torch.equal(lo8[:, :3], lo[:, :3])


@pytest.mark.parametrize("shape", SHAPES)
def test_split_mode_is_fp32_class(shape):
    from tracklab_amd import _lib
    x, wt, b, r, k, s = _inputs(shape)
    xh, xl = _lib.split_planes(x)
    wh, wl = _lib.split_planes(wt)
    rh, rl = _lib.split_planes(r) if r is not None else (None, None)
    y32 = _lib.conv2d_nhwc_16(xh, wh, b, "relu", rh, stride=s, x_lo=xl, weight_lo=wl, residual_lo=rl, out_f32=True)
    yh, yl = _lib.conv2d_nhwc_16(xh, wh, b, "relu", rh, stride=s, x_lo=xl, weight_lo=wl, residual_lo=rl)
    ref = F.conv2d(x.double(), wt.double(), b.double(), s, k // 2)
    bound = F.conv2d(x.double().abs(), wt.double().abs(), b.double().abs(), s, k // 2)
    if r is not None:
        ref, bound = ref + r.double(), bound + r.double().abs()
    ref = F.relu(ref)
    assert y32.shape == ref.shape
    err = (y32.double() - ref).abs()
    assert bool((err <= 2e-6 * bound + 1e-30).all()), float((err / bound).max())      # the exact-fp32 kernel's bound (tests/test_gpu_conv.py)
    merged = _lib.merge_planes(yh, yl)
    assert float(((merged - y32).abs() / y32.abs().clamp_min(1e-6)).max()) <= 2.0 ** -20      # the (hi, lo) output is y32 to the pair's precision


@pytest.mark.parametrize("shape", SHAPES)
def test_f16_mode_is_an_fp32_accumulated_f16_convolution(shape):
    from tracklab_amd import _lib
    x, wt, b, r, k, s = _inputs(shape, seed=2)
    xh, wh = x.half(), wt.half()
    rh = r.half() if r is not None else None
    y = _lib.conv2d_nhwc_16(xh, wh, b, "relu", rh, stride=s)
    ref = F.conv2d(xh.double(), wh.double(), b.double(), s, k // 2)
    if r is not None:
        ref = ref + rh.double()
    ref = F.relu(ref)
    bound = F.conv2d(xh.double().abs(), wh.double().abs(), b.double().abs(), s, k // 2)
    err = (y.double() - ref).abs()
    # one f16 rounding of the result (2^-11 relative) + fp32 accumulation round-off of the sum
    assert bool((err <= ref.abs() * 2.0 ** -11 + 2e-6 * bound + 1e-7).all()), float((err / (ref.abs() + 1e-3)).max())


def test_reid_network_in_split_precision_tracks_the_exact_fp32_network():
    from tracklab_amd.backbones.reid import part_based_reid
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((6, 4, 384, 128), device="cuda", generator=g)[:, :3].contiguous(memory_format=torch.channels_last)
    exact = part_based_reid(dtype=torch.float32)
    split = part_based_reid(dtype=torch.float32, split_precision=True)
    half = part_based_reid(dtype=torch.float16)
    with torch.no_grad():
        e0, v0 = exact(x)
        e1, v1 = split(x)
        e2, _ = half(x.half())
    scale = float(e0.abs().max())
    d_split, d_half = float((e1 - e0).abs().max()) / scale, float((e2.float() - e0).abs().max()) / scale
    assert d_split <= 2e-5, d_split                       # fp32-class: the two fp32 routes differ by summation order and 2^-22 operand error
    assert d_split * 20 < d_half, (d_split, d_half)       # ... an order of magnitude or more below the f16 leg's distance
    assert torch.equal(v0, v1)
    cos = F.cosine_similarity(e0.double().flatten(1), e1.double().flatten(1)).min().item()
    assert cos > 1 - 1e-9, cos


@pytest.mark.parametrize("shape", [(3, 48, 16, 12, 48, 1, 1, True), (2, 96, 9, 7, 96, 1, 1, True), (2, 256, 6, 5, 192, 3, 1, True)])
def test_residual_after_the_activation(shape):
    """TLK_ACT_RES_AFTER: y = act(conv + bias) + r (CSPNeXt's identity add) in f16 and split mode, through both kernels (register-staged for
    Cout <= 64 / ragged Cin, direct-to-LDS otherwise)"""
    from tracklab_amd import _lib
    x, wt, b, r, k, s = _inputs(shape, seed=3)
    ref = F.silu(F.conv2d(x.half().double(), wt.half().double(), b.double(), s, k // 2)) + r.half().double()
    y = _lib.conv2d_nhwc_16(x.half(), wt.half(), b, "silu", r.half(), stride=s, residual_after_act=True)
    assert bool(((y.double() - ref).abs() <= 2e-3 * ref.abs() + 2e-3).all())
    xh, xl = _lib.split_planes(x); wh, wl = _lib.split_planes(wt); rh, rl = _lib.split_planes(r)
    y32 = _lib.conv2d_nhwc_16(xh, wh, b, "silu", rh, stride=s, x_lo=xl, weight_lo=wl, residual_lo=rl, out_f32=True, residual_after_act=True)
    ref = F.silu(F.conv2d(x.double(), wt.double(), b.double(), s, k // 2)) + r.double()
    bound = F.conv2d(x.double().abs(), wt.double().abs(), b.double().abs(), s, k // 2) + r.double().abs()
    assert bool(((y32.double() - ref).abs() <= 4e-6 * bound + 1e-30).all())


# ---- r05: the large-tile / one-stage / ring kernels of tlk_conv16x.hip (tlk_conv16_set_config forces a tile configuration) -------------------
X_SHAPES = [(5, 64, 24, 8, 64, 3, 1, True), (3, 256, 12, 8, 256, 1, 1, True), (2, 128, 17, 9, 200, 3, 2, False), (7, 512, 6, 4, 512, 3, 1, False),
            (2, 192, 11, 7, 72, 1, 1, True), (1, 64, 40, 24, 256, 1, 1, True)]


def _force16(cfg):
    from tracklab_amd import _lib
    L = _lib.lib()
    _lib.conv2d_nhwc_16(torch.zeros((0, 8, 2, 2), device="cuda", dtype=torch.float16).contiguous(memory_format=torch.channels_last),
                        torch.zeros((8, 8, 1, 1), device="cuda", dtype=torch.float16))          # binds the symbols
    _lib.check(L.tlk_conv16_set_config(cfg))


@pytest.mark.parametrize("cfg", list(range(1, 17)) + [25, 26])          # (25 / 26, r06: the 32-column tiles)
@pytest.mark.parametrize("shape", X_SHAPES + [(3, 64, 24, 8, 32, 1, 1, False)])
def test_every_f16_tile_configuration_of_the_direct_to_lds_kernels(shape, cfg):
    """one / two / three / four LDS stages, 64 x 64 ... 256 x 256 tiles, residual prefetched or not: the same fp32-accumulated f16 convolution
    (ragged M and Cout, taps outside the image, strides), to the bound of the r04 kernels' test above"""
    from tracklab_amd import _lib
    x, wt, b, r, k, s = _inputs(shape, seed=cfg)
    xh, wh = x.half(), wt.half()
    rh = r.half() if r is not None else None
    try:
        _force16(cfg)
        y = _lib.conv2d_nhwc_16(xh, wh, b, "relu", rh, stride=s)
    finally:
        _force16(0)
    ref = F.conv2d(xh.double(), wh.double(), b.double(), s, k // 2)
    if r is not None:
        ref = ref + rh.double()
    ref = F.relu(ref)
    bound = F.conv2d(xh.double().abs(), wh.double().abs(), b.double().abs(), s, k // 2)
    err = (y.double() - ref).abs()
    assert bool((err <= ref.abs() * 2.0 ** -11 + 2e-6 * bound + 1e-7).all()), float((err / (ref.abs() + 1e-3)).max())


# r06: HALF-STEP tiles (K step = 32 halfs) -- Cin a multiple of 32 but not of 64 (YOLOX-m's 96-channel layers, 160, 32, 288)
HALF_SHAPES = [(4, 96, 20, 20, 96, 3, 1, True), (2, 96, 40, 40, 192, 1, 1, False), (3, 32, 17, 9, 72, 3, 2, False), (2, 160, 12, 8, 128, 1, 1, True),
               (1, 288, 10, 10, 96, 3, 1, False)]


@pytest.mark.parametrize("cfg", [0, 19, 20, 21, 22, 23, 24])
@pytest.mark.parametrize("shape", HALF_SHAPES + [(3, 32, 24, 8, 32, 3, 1, True)])
def test_f16_half_step_tiles(shape, cfg):
    """conv16x_kernel<..., ROWB = 64>: 64-byte LDS rows, 32 halfs per K step; cfg 0 = the heuristic, which routes these shapes to them (they used to
    fall to the r04 register-staged kernel's general loader)"""
    from tracklab_amd import _lib
    x, wt, b, r, k, s = _inputs(shape, seed=60 + cfg)
    xh, wh = x.half(), wt.half()
    rh = r.half() if r is not None else None
    try:
        _force16(cfg)
        y = _lib.conv2d_nhwc_16(xh, wh, b, "silu", rh, stride=s, residual_after_act=True) if r is not None else _lib.conv2d_nhwc_16(xh, wh, b, "relu", None, stride=s)
    finally:
        _force16(0)
    pre = F.conv2d(xh.double(), wh.double(), b.double(), s, k // 2)
    ref = F.silu(pre) + rh.double() if r is not None else F.relu(pre)
    bound = F.conv2d(xh.double().abs(), wh.double().abs(), b.double().abs(), s, k // 2) + (rh.double().abs() if r is not None else 0)
    err = (y.double() - ref).abs()
    assert bool((err <= ref.abs() * 2.0 ** -11 + 4e-6 * bound + 1e-7).all()), float((err / (ref.abs() + 1e-3)).max())
    # the r04 kernel on the same layer: same products, same fp32 accumulation order within a 16-wide slice -- at most the output rounding apart
    try:
        _force16(-1)
        y0 = _lib.conv2d_nhwc_16(xh, wh, b, "silu", rh, stride=s, residual_after_act=True) if r is not None else _lib.conv2d_nhwc_16(xh, wh, b, "relu", None, stride=s)
    finally:
        _force16(0)
    assert float((y.float() - y0.float()).abs().max()) <= 2.0 ** -9 * max(1.0, float(y0.float().abs().max()))


# (n, cin, h, w, cout, k, stride, residual, split) at sizes where the r06 rules of the sweep apply -> the configuration the heuristic must pick
R06_ROUTES16 = [
    ((40, 32, 96, 32, 32, 3, 1, True, False), 23),      # half-step K, <= 32 outputs, >= 2 tiles per CU: 128 x 32 tiles of two wavefronts (HRNet's 32-channel branch)
    ((2, 32, 96, 32, 32, 3, 1, True, False), 19),       # ... a small launch of the same layer: 64 x 64 tiles, four stages
    ((100, 64, 48, 16, 32, 1, 1, False, False), 25),    # full-step K < 1024 into 32 channels (the exchange paths): 128 x 32, one stage
    ((40, 32, 96, 32, 32, 3, 1, True, True), 11),       # split mode, <= 32 outputs: 128 x 32, one stage
    ((40, 256, 96, 32, 32, 3, 1, False, True), 12),     # ... on a long K loop: three stages
    ((24, 96, 80, 80, 48, 1, 1, False, False), 19),
    ((40, 256, 96, 32, 32, 3, 1, False, False), 16),    # 32 wide on a full K step: 64 x 64 tiles
    ((200, 192, 16, 12, 192, 3, 1, False, False), 10),  # 192 wide = three 64-wide tiles: 256 x 64, one stage
    ((600, 256, 12, 4, 256, 3, 1, True, False), 9),     # a long K loop with residual: the tile that reads the residual in its epilogue
    ((24, 192, 40, 40, 192, 3, 1, False, True), 7),     # split mode, the detector's few-tile layers: 128 x 128 of eight wavefronts
    ((24, 384, 40, 40, 192, 1, 1, False, True), 7),
]


@pytest.mark.parametrize("case,want", R06_ROUTES16)
def test_r06_heuristic_routes_of_the_16_bit_kernels(case, want):
    """tools/sweep_conv16.py (profiles/r06_conv16_sweep.txt): the configuration the heuristic picks for the shapes the r06 rules were written for
    (tlk_conv16_last_config), and the result against fp64 on the f16 operands to the bound of the tests above"""
    from tracklab_amd import _lib
    n, cin, h, w, cout, k, s, res, split = case
    g = torch.Generator(device="cuda").manual_seed(cin + cout)
    x = torch.randn((n, h, w, cin), device="cuda", generator=g).permute(0, 3, 1, 2)
    wt = (torch.randn((cout, k, k, cin), device="cuda", generator=g) * 0.05).permute(0, 3, 1, 2)
    b = torch.randn(cout, device="cuda", generator=g)
    _force16(0)
    L = _lib.lib()
    L.tlk_conv16_last_config.restype = C.c_int
    if split:
        xh, xl = _lib.split_planes(x.contiguous(memory_format=torch.channels_last))
        wh, wl = _lib.split_planes(wt.contiguous(memory_format=torch.channels_last))
        y = _lib.conv2d_nhwc_16(xh, wh, b, "silu", None, stride=s, x_lo=xl, weight_lo=wl, out_f32=True)
        assert L.tlk_conv16_last_config() == want, (case, L.tlk_conv16_last_config())
        ref = F.silu(F.conv2d(x.double(), wt.double(), b.double(), s, k // 2))
        bound = F.conv2d(x.double().abs(), wt.double().abs(), b.double().abs(), s, k // 2)
        assert bool(((y.double() - ref).abs() <= 2e-6 * bound + 1e-7).all())
        return
    xh, wh = x.half(), wt.half()
    rh = torch.randn((n, cout, (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1), device="cuda", generator=g).half().contiguous(memory_format=torch.channels_last) if res else None
    y = _lib.conv2d_nhwc_16(xh, wh, b, "relu", rh, stride=s)
    assert L.tlk_conv16_last_config() == want, (case, L.tlk_conv16_last_config())
    ref = F.conv2d(xh.double(), wh.double(), b.double(), s, k // 2)
    if res:
        ref = ref + rh.double()
    ref = F.relu(ref)
    bound = F.conv2d(xh.double().abs(), wh.double().abs(), b.double().abs(), s, k // 2) + (rh.double().abs() if res else 0)
    err = (y.double() - ref).abs()
    assert bool((err <= ref.abs() * 2.0 ** -11 + 4e-6 * bound + 1e-7).all()), float((err / (ref.abs() + 1e-3)).max())


def test_a_full_step_configuration_refuses_a_half_step_layer():
    from tracklab_amd import _lib
    x, wt, b, r, k, s = _inputs(HALF_SHAPES[0])
    try:
        _force16(3)
        with pytest.raises(_lib.TlkError, match="half-step"):
            _lib.conv2d_nhwc_16(x.half(), wt.half(), b, "relu", None, stride=s)
    finally:
        _force16(0)


# (n, cin, h, w, cout, k, stride, residual): 3 x 3 / 1 on exactly 64 channels, whole image rows per 256- / 128-pixel tile
PATCH16_SHAPES = [(3, 64, 16, 32, 64, 3, 1, True), (2, 64, 32, 16, 128, 3, 1, False), (2, 64, 8, 64, 72, 3, 1, True), (1, 64, 64, 8, 64, 3, 1, False)]


@pytest.mark.parametrize("cfg", [0, 17, 18])
@pytest.mark.parametrize("shape", PATCH16_SHAPES)
def test_f16_patch_resident_3x3_kernel(shape, cfg):
    """conv16x_kernel<..., MODE_F16, PATCH>: the tile's input rows + halo land in LDS once, the nine taps are nine shifted fragment reads; cfg 0 =
    the heuristic, which routes these shapes (ResNet-50's layer 1) to it"""
    from tracklab_amd import _lib
    x, wt, b, r, k, s = _inputs(shape, seed=40 + cfg)
    xh, wh = x.half(), wt.half()
    rh = r.half() if r is not None else None
    try:
        _force16(cfg)
        y = _lib.conv2d_nhwc_16(xh, wh, b, "relu", rh, stride=s)
    finally:
        _force16(0)
    ref = F.conv2d(xh.double(), wh.double(), b.double(), s, k // 2)
    if r is not None:
        ref = ref + rh.double()
    ref = F.relu(ref)
    bound = F.conv2d(xh.double().abs(), wh.double().abs(), b.double().abs(), s, k // 2)
    err = (y.double() - ref).abs()
    assert bool((err <= ref.abs() * 2.0 ** -11 + 2e-6 * bound + 1e-7).all()), float((err / (ref.abs() + 1e-3)).max())


# (8..12, r06: the 32-column tiles of HRNet-W32's high-resolution branch -- with a 32-channel layer, with and without residual, and a ragged Cout)
@pytest.mark.parametrize("cfg", list(range(1, 13)))
@pytest.mark.parametrize("shape", [X_SHAPES[0], X_SHAPES[1], X_SHAPES[2], X_SHAPES[3], (3, 32, 24, 8, 32, 3, 1, True), (2, 64, 13, 7, 40, 3, 2, False)])
def test_every_split_tile_configuration_is_fp32_class(shape, cfg):
    from tracklab_amd import _lib
    x, wt, b, r, k, s = _inputs(shape, seed=10 + cfg)
    xh, xl = _lib.split_planes(x)
    wh, wl = _lib.split_planes(wt)
    rh, rl = _lib.split_planes(r) if r is not None else (None, None)
    try:
        _force16(cfg)
        y32 = _lib.conv2d_nhwc_16(xh, wh, b, "relu", rh, stride=s, x_lo=xl, weight_lo=wl, residual_lo=rl, out_f32=True)
        yh, yl = _lib.conv2d_nhwc_16(xh, wh, b, "relu", rh, stride=s, x_lo=xl, weight_lo=wl, residual_lo=rl)
    finally:
        _force16(0)
    ref = F.conv2d(x.double(), wt.double(), b.double(), s, k // 2)
    bound = F.conv2d(x.double().abs(), wt.double().abs(), b.double().abs(), s, k // 2)
    if r is not None:
        ref, bound = ref + r.double(), bound + r.double().abs()
    ref = F.relu(ref)
    err = (y32.double() - ref).abs()
    assert bool((err <= 2e-6 * bound + 1e-30).all()), float((err / bound).max())
    merged = _lib.merge_planes(yh, yl)
    assert float(((merged - y32).abs() / y32.abs().clamp_min(1e-6)).max()) <= 2.0 ** -20


def test_the_default_route_picks_by_launch_size_and_stays_correct():
    """the heuristic of launch16x over the launch sizes it distinguishes: one frame / one hundred crops / a full step"""
    from tracklab_amd import _lib
    for n in (1, 8, 300):
        for shape in ((n, 192, 20, 20, 192, 3, 1, False), (n, 256, 24, 8, 1024, 1, 1, True), (n, 64, 96, 32, 64, 3, 1, False), (n, 1024, 24, 8, 512, 1, 1, False)):
            x, wt, b, r, k, s = _inputs(shape, seed=n)
            xh, wh = x.half(), wt.half()
            rh = r.half() if r is not None else None
            y = _lib.conv2d_nhwc_16(xh, wh, b, "silu", rh, stride=s)
            ref = F.conv2d(xh.double(), wh.double(), b.double(), s, k // 2)
            if r is not None:
                ref = ref + rh.double()
            ref = F.silu(ref)
            assert bool(((y.double() - ref).abs() <= 2e-3 * ref.abs() + 2e-3).all()), shape


STEM16_CASES = [
    # n, h, w, cout, k, act, pool
    (3, 384, 128, 64, 7, "relu", True),         # the ReID crops: 192 x 64 map -> 96 x 32 pooled
    (2, 41, 27, 64, 7, "relu", True),           # ragged: odd map sizes, strips beyond the image, columns beyond Wo
    (2, 50, 128, 32, 7, None, True),            # one cout tile, no activation (negative values through the pool's -inf padding)
    (2, 37, 301, 64, 7, "relu", False),         # no pool: several column strips
    (3, 64, 48, 32, 3, "silu", False),          # 3 x 3 stem (RTMPose / HRNet)
    (1, 256, 192, 64, 3, "relu", False),
    (2, 30, 22, 48, 3, "relu", True),
]


@pytest.mark.parametrize("case", STEM16_CASES)
def test_f16_rgb_stem_kernel_matches_the_two_pass_route(case):
    """r05, csrc/tlk_conv_stem16.hip: stem convolution + bias + activation (+ fused 3 x 3 / 2 max-pool) against fp32 arithmetic on the same
    f16-rounded operands, result rounded to f16 before the pool (max commutes with the rounding): one f16 ulp of slack"""
    import torch.nn.functional as F
    from tracklab_amd import _lib
    n, h, w, cout, k, act, pool = case
    g = torch.Generator(device="cpu").manual_seed(k * 1000 + cout + h)
    x = torch.randn(n, h, w, 3, generator=g).half().cuda().permute(0, 3, 1, 2)
    wt = (torch.randn(cout, k, k, 3, generator=g) * 0.1).half().cuda().permute(0, 3, 1, 2)
    b = torch.randn(cout, generator=g).cuda()
    ref = F.conv2d(x.float(), wt.float(), b, stride=2, padding=k // 2)
    ref = torch.relu(ref) if act == "relu" else (F.silu(ref) if act == "silu" else ref)
    ref = ref.half().float()
    if pool:
        ref = F.max_pool2d(ref, 3, 2, 1)
    packed = _lib.conv_stem16_pack(wt)
    y = _lib.conv_stem16(x, packed, cout, k, b, act, pool=pool)
    assert y.shape == ref.shape and y.dtype == torch.float16 and y.is_contiguous(memory_format=torch.channels_last)
    err = (y.float() - ref).abs()
    tol = 2e-3 * ref.abs() + 2e-3
    assert bool((err <= tol).all()), f"max err {float(err.max())} at {int(err.argmax())}"


def test_f16_stem_honours_the_dynamic_batch_and_the_resnet_module_uses_it():
    from tracklab_amd import _lib
    from tracklab_amd.backbones import reid
    x = torch.randn(5, 64, 32, 3).half().cuda().permute(0, 3, 1, 2)
    wt = (torch.randn(64, 7, 7, 3) * 0.1).half().cuda().permute(0, 3, 1, 2)
    packed = _lib.conv_stem16_pack(wt)
    full = _lib.conv_stem16(x, packed, 64, 7, None, "relu", pool=True)
    live = torch.tensor([3], dtype=torch.int32, device="cuda")
    out = torch.full_like(full, -3.0)
    _lib.conv_set_dynamic_batch(live)
    try:
        _lib.conv_stem16(x, packed, 64, 7, None, "relu", pool=True, out=out)
    finally:
        _lib.conv_set_dynamic_batch(None)
    assert torch.equal(out[:3], full[:3]) and bool((out[3:] == -3.0).all())
    net = reid._ResNet50().cuda().half().eval().to(memory_format=torch.channels_last)
    xin = torch.randn(2, 128, 64, 3).half().cuda().permute(0, 3, 1, 2)
    with torch.no_grad():
        y = net.conv1.stem16(xin, pool=True)
        ref = torch.nn.functional.max_pool2d(torch.relu(torch.nn.functional.conv2d(xin.float(), net.conv1.conv.weight.float(), net.conv1.bias.float(), stride=2, padding=3)).half().float(), 3, 2, 1)
    assert y is not None and y.shape == ref.shape
    assert bool(((y.float() - ref).abs() <= 2e-3 * ref.abs() + 2e-3).all())
