"""tlk_dwconv2d_nhwc (csrc/tlk_dwconv.hip): depthwise k x k convolution + bias + activation, the depthwise halves of RTMPose's CSPNeXt blocks.
  * fp32: bit-exact against oracle/src/conv.c orc_dwconv2d_nhwc_f32 (the same fmaf chain) -- ragged widths, images shorter than the kernel,
    several strips per column, channel-sliced input / output; SiLU within the device exp's error;
  * f16: fp32 accumulation of exact f16 x f16 products, so the result is the fp32 oracle run on the f16-rounded operands, rounded to f16
    (1 ulp of f16 allowed for the final rounding of a differently rounded fp32 sum);
  * both within round-off of torch's own depthwise convolution (the reference's pose network is a third-party fp32 ONNX model behind
    tracklab/wrappers/pose_estimator/rtmlib_api.py:21-36: tolerance, stated here);
  * the RTMPose forward through the kernel == the same module on the library route, to that tolerance."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # n, h, w, c, k, act
    (2, 9, 7, 8, 5, "relu"),
    (3, 64, 48, 48, 5, None),          # CSPNeXt-m stage 1
    (2, 32, 24, 96, 5, "relu"),        # stage 2
    (5, 8, 6, 384, 5, "none"),         # stage 4
    (1, 3, 2, 16, 5, "relu"),          # image smaller than the kernel
    (1, 1, 1, 8, 3, None),
    (2, 40, 5, 24, 3, "relu"),         # small batch: several strips per column
    (1, 70, 33, 12, 5, None),          # strips with a ragged last one
    (7, 6, 11, 4, 3, "relu"),
]


def _inputs(case):
    n, h, w, c, k, act = case
    rng = np.random.default_rng(hash(case[:5]) & 0xffff)
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    wt = (rng.standard_normal((k, k, c)) * 0.3).astype(np.float32)
    b = rng.standard_normal(c).astype(np.float32)
    return x, wt, b


@pytest.mark.parametrize("case", CASES)
def test_dwconv_f32_is_bit_exact_with_the_oracle_chain(case):
    import oracle
    from tracklab_amd import _lib
    x, wt, b = _inputs(case)
    exp = oracle.dwconv2d_nhwc_f32(x, wt, b, case[5])
    y = _lib.dwconv2d_nhwc(torch.from_numpy(x).cuda().permute(0, 3, 1, 2), torch.from_numpy(wt).cuda(), torch.from_numpy(b).cuda(), case[5])
    got = y.permute(0, 2, 3, 1).cpu().numpy()
    assert np.array_equal(got, exp), f"max |diff| {np.abs(got - exp).max()}"
    ref = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2).double(), torch.from_numpy(wt).permute(2, 0, 1).unsqueeze(1).double(),
                   torch.from_numpy(b).double(), 1, case[4] // 2, groups=case[3])
    ref = F.relu(ref) if case[5] == "relu" else ref
    np.testing.assert_allclose(got, ref.permute(0, 2, 3, 1).numpy(), rtol=1e-5, atol=1e-5)


def test_dwconv_f32_silu_within_exp_roundoff_and_no_bias():
    import oracle
    from tracklab_amd import _lib
    x, wt, b = _inputs((2, 16, 12, 48, 5, "silu"))
    xt, wtt = torch.from_numpy(x).cuda().permute(0, 3, 1, 2), torch.from_numpy(wt).cuda()
    got = _lib.dwconv2d_nhwc(xt, wtt, torch.from_numpy(b).cuda(), "silu").permute(0, 2, 3, 1).cpu().numpy()
    np.testing.assert_allclose(got, oracle.dwconv2d_nhwc_f32(x, wt, b, "silu"), rtol=2e-6, atol=1e-6)
    got = _lib.dwconv2d_nhwc(xt, wtt, None, None).permute(0, 2, 3, 1).cpu().numpy()
    assert np.array_equal(got, oracle.dwconv2d_nhwc_f32(x, wt, None, None))


@pytest.mark.parametrize("case", [c for c in CASES if c[3] % 8 == 0])
def test_dwconv_f16_is_the_fp32_chain_on_f16_operands(case):
    import oracle
    from tracklab_amd import _lib
    x, wt, b = _inputs(case)
    xh, wh = x.astype(np.float16), wt.astype(np.float16)
    exp32 = oracle.dwconv2d_nhwc_f32(xh.astype(np.float32), wh.astype(np.float32), b, case[5])
    y = _lib.dwconv2d_nhwc(torch.from_numpy(xh).cuda().permute(0, 3, 1, 2), torch.from_numpy(wh).cuda(), torch.from_numpy(b).cuda(), case[5])
    assert y.dtype == torch.float16
    got = y.permute(0, 2, 3, 1).cpu().numpy()
    assert np.array_equal(got, exp32.astype(np.float16)), f"max |diff| {np.abs(got.astype(np.float32) - exp32).max()}"


@pytest.mark.parametrize("cfg", [1, 2, 3, 4])
@pytest.mark.parametrize("dt", ["f32", "f16"])
def test_dwconv_every_lane_shape_gives_the_same_bits(cfg, dt):
    """r06: columns per lane (1 / 2) x rows loaded ahead (1 / 2) -- odd widths (a lane's second column outside the image), images narrower than
    the kernel, ragged strips; fp32 against the oracle chain, f16 against the chain on the f16 operands rounded once"""
    import oracle
    from tracklab_amd import _lib
    _lib.dwconv_set_config(cfg)
    try:
        for case in CASES:
            if dt == "f16" and case[3] % 8:
                continue
            x, wt, b = _inputs(case)
            if dt == "f16":
                x, wt = x.astype(np.float16), wt.astype(np.float16)
            exp = oracle.dwconv2d_nhwc_f32(x.astype(np.float32), wt.astype(np.float32), b, case[5])
            y = _lib.dwconv2d_nhwc(torch.from_numpy(x).cuda().permute(0, 3, 1, 2), torch.from_numpy(wt).cuda(), torch.from_numpy(b).cuda(), case[5])
            got = y.permute(0, 2, 3, 1).cpu().numpy()
            assert np.array_equal(got, exp.astype(got.dtype)), (case, cfg, dt)
    finally:
        _lib.dwconv_set_config(0)


def test_dwconv_f16_silu_is_within_one_f16_step_of_the_oracle():
    import oracle
    from tracklab_amd import _lib
    for case in ((3, 64, 48, 48, 5), (2, 17, 7, 8, 3), (2, 9, 5, 24, 5)):
        x, wt, b = _inputs(case + (None,))
        xh, wh = x.astype(np.float16), wt.astype(np.float16)
        exp = oracle.dwconv2d_nhwc_f32(xh.astype(np.float32), wh.astype(np.float32), b, "silu")
        y = _lib.dwconv2d_nhwc(torch.from_numpy(xh).cuda().permute(0, 3, 1, 2), torch.from_numpy(wh).cuda(), torch.from_numpy(b).cuda(), "silu")
        got = y.permute(0, 2, 3, 1).cpu().numpy().astype(np.float32)
        # fp32 SiLU within ~3e-7 (device exp + reciprocal), then ONE rounding to f16: equal, or the neighbouring f16 where the fp32 value sits on a tie
        step = np.spacing(np.abs(exp).astype(np.float16)).astype(np.float32)
        assert (np.abs(got - exp) <= 0.5 * step + 4e-7 * np.abs(exp) + 1e-12).all()
        assert (got == exp.astype(np.float16).astype(np.float32)).mean() > 0.999


def test_dwconv_reads_and_writes_channel_slices():
    import oracle
    from tracklab_amd import _lib
    rng = np.random.default_rng(5)
    wide = rng.standard_normal((2, 10, 9, 40)).astype(np.float32)
    wt = rng.standard_normal((5, 5, 16)).astype(np.float32)
    b = rng.standard_normal(16).astype(np.float32)
    xt = torch.from_numpy(wide).cuda().permute(0, 3, 1, 2)
    out = torch.full((2, 48, 10, 9), 7.0, device="cuda").contiguous(memory_format=torch.channels_last)
    _lib.dwconv2d_nhwc(xt[:, 8:24], torch.from_numpy(wt).cuda(), torch.from_numpy(b).cuda(), "relu", out=out[:, 32:48])
    exp = oracle.dwconv2d_nhwc_f32(wide[..., 8:24], wt, b, "relu")
    o = out.permute(0, 2, 3, 1).cpu().numpy()
    assert np.array_equal(o[..., 32:], exp) and np.all(o[..., :32] == 7.0)


def test_dwconv_rejects_what_it_does_not_implement():
    from tracklab_amd import _lib
    x = torch.zeros(1, 8, 4, 4, device="cuda").contiguous(memory_format=torch.channels_last)
    with pytest.raises(_lib.TlkError):
        _lib.dwconv2d_nhwc(x, torch.zeros(7, 7, 8, device="cuda"), None, None)          # k = 7
    x6 = torch.zeros(1, 6, 4, 4, device="cuda").contiguous(memory_format=torch.channels_last)
    with pytest.raises(_lib.TlkError):
        _lib.dwconv2d_nhwc(x6, torch.zeros(3, 3, 6, device="cuda"), None, None)         # 6 fp32 channels: not a multiple of 16 bytes
    with pytest.raises(_lib.TlkError):
        _lib.dwconv_set_config(5)


def _pose_outputs(dtype, x):
    import importlib
    R = importlib.import_module("tracklab_amd.backbones.rtmpose")
    Cm = importlib.import_module("tracklab_amd.backbones.common")
    net = R.rtmpose("m", "cuda", dtype)                      # seeded: the same weights for every dtype before the cast
    switches = ((R, "USE_TLK_DWCONV"), (R, "USE_SLICE_CONCAT"), (R, "USE_FINAL_GEMM"), (Cm, "USE_TLK_SPP"), (Cm, "USE_TLK_CONV_F16_NARROW"))
    with torch.no_grad():
        mine = net(x.to(dtype))
        old = [getattr(m, k) for m, k in switches]
        for m, k in switches:
            setattr(m, k, False)                             # library route: MIOpen / hipBLASLt + separate passes (fp32: libtlk's convolution stays)
        try:
            lib = net(x.to(dtype))
        finally:
            for (m, k), v in zip(switches, old):
                setattr(m, k, v)
    return mine, lib


def test_rtmpose_forward_equals_the_library_route():
    """Every libtlk piece of the pose network on (depthwise kernel, SPP pass, narrow pointwise convolutions with the identity add and the
    concatenation written in place, final layer as a GEMM) against all of them off.
    fp32: the forward through the depthwise kernel == the library route to fp32 round-off.  f16: two f16 evaluations of a 60-layer random
    network differ by their accumulated roundings, so both are measured against the fp32 network: the kernel's route (fp32 accumulation
    inside the depthwise convolution) must not be further from it than the library route is."""
    x = torch.randn(6, 3, 256, 192, device="cuda").contiguous(memory_format=torch.channels_last)
    m32, l32 = _pose_outputs(torch.float32, x)
    for a, r in zip(m32, l32):
        assert (a - r).abs().max().item() <= 2e-4 * r.abs().max().item()
    m16, l16 = _pose_outputs(torch.float16, x)
    for a, b, r in zip(m16, l16, l32):
        err_mine, err_lib = (a - r).abs().max().item(), (b - r).abs().max().item()
        assert err_mine <= 1.5 * err_lib + 1e-3 * r.abs().max().item(), (err_mine, err_lib)
