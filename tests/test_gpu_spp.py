"""tlk_spp_maxpool_nhwc (csrc/tlk_spp.hip): [x | max 5x5 | max 9x9 | max 13x13] in one pass -- the pooling half of the SPPBottleneck of YOLOX's
CSPDarknet and RTMPose's CSPNeXt.  max is exact: bit-identical to oracle/src/conv.c orc_spp_maxpool_nhwc_f32 and to torch's max_pool2d + cat,
in fp32 and f16, for maps smaller than the windows, ragged sizes, channel-sliced input / output."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [(2, 8, 6, 16), (1, 25, 45, 24), (3, 1, 1, 8), (2, 3, 17, 8), (5, 14, 2, 32), (1, 20, 20, 384)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("case", CASES)
def test_spp_equals_the_oracle_and_torch(case, dtype):
    import oracle
    from tracklab_amd import _lib
    n, h, w, c = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    if dtype == torch.float16:
        x = x.astype(np.float16).astype(np.float32)
    xt = torch.from_numpy(x).cuda().to(dtype).permute(0, 3, 1, 2)
    got = _lib.spp_maxpool_nhwc(xt)
    assert got.shape == (n, 4 * c, h, w) and got.is_contiguous(memory_format=torch.channels_last)
    assert np.array_equal(got.permute(0, 2, 3, 1).float().cpu().numpy(), oracle.spp_maxpool_nhwc_f32(x))
    ref = torch.cat([xt] + [F.max_pool2d(xt, k, 1, k // 2) for k in (5, 9, 13)], 1)
    assert torch.equal(got, ref)


def test_spp_reads_and_writes_channel_slices():
    from tracklab_amd import _lib
    wide = torch.randn(2, 40, 9, 7, device="cuda").contiguous(memory_format=torch.channels_last)
    out = torch.full((2, 72, 9, 7), 3.0, device="cuda").contiguous(memory_format=torch.channels_last)
    xs = wide[:, 8:24]
    _lib.spp_maxpool_nhwc(xs, out=out[:, 8:72])
    ref = torch.cat([xs] + [F.max_pool2d(xs, k, 1, k // 2) for k in (5, 9, 13)], 1)
    assert torch.equal(out[:, 8:], ref) and bool((out[:, :8] == 3.0).all())


def test_spp_rejects_channel_counts_that_are_not_16_byte_multiples():
    from tracklab_amd import _lib
    x = torch.zeros(1, 6, 4, 4, device="cuda").contiguous(memory_format=torch.channels_last)
    with pytest.raises(_lib.TlkError):
        _lib.spp_maxpool_nhwc(x)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_sppbottleneck_modules_equal_their_library_route(dtype):
    import importlib
    common, rtmpose, yolox = (importlib.import_module("tracklab_amd.backbones." + m) for m in ("common", "rtmpose", "yolox"))
    for mod, shape in ((yolox.SPPBottleneck(64, 64), (2, 64, 25, 45)), (rtmpose.SPPBottleneck(64, 64), (3, 64, 8, 6))):
        mod = common.finalize(common.random_init_(mod, 1), "cuda", dtype, True)
        x = torch.randn(*shape, device="cuda", dtype=dtype).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            a = mod(x)
            common.USE_TLK_SPP = False
            try:
                b = mod(x)
            finally:
                common.USE_TLK_SPP = True
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 5e-3), (torch.float32, 2e-5)])
def test_pose_final_layer_as_one_gemm(dtype, tol):
    """RTMPoseNet._final_maps: the 7 x 7 convolution on the 8 x 6 map as a dense GEMM with the Toeplitz form of the weight == the convolution"""
    import importlib
    R = importlib.import_module("tracklab_amd.backbones.rtmpose")
    net = R.rtmpose("m", "cuda", dtype)
    with torch.no_grad():
        net.final_layer.bias.normal_()
        f = torch.randn(5, net.final_layer.in_channels, 8, 6, device="cuda", dtype=dtype).contiguous(memory_format=torch.channels_last)
        got = net._final_maps(f)
        ref = net.final_layer(f).flatten(2)
    assert got.shape == ref.shape == (5, 17, 48)
    assert (got.float() - ref.float()).abs().max().item() <= tol * ref.float().abs().max().item()


@pytest.mark.parametrize("dtype_name", ["float32", "float16"])
@pytest.mark.parametrize("shape,k,s,p", [((3, 64, 192, 64), 3, 2, 1), ((2, 16, 7, 5), 3, 2, 1), ((2, 8, 9, 9), 2, 2, 0), ((1, 32, 6, 11), 5, 1, 2)])
def test_maxpool2d_equals_torch_bit_for_bit(shape, k, s, p, dtype_name):
    """tlk_maxpool2d_nhwc (r05): ResNet-50's stem pool in one hand-written pass; max is exact -> torch.equal.  With a dynamic batch set, the
    images beyond the live count are not written."""
    import torch
    import torch.nn.functional as F
    from tracklab_amd import _lib
    dt = getattr(torch, dtype_name)
    x = torch.randn(shape, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    y = _lib.maxpool2d_nhwc(x, k, s, p)
    assert torch.equal(y, F.max_pool2d(x, k, s, p))
    live = torch.tensor([1], dtype=torch.int32, device="cuda")
    out = torch.full_like(y, -5.0)
    _lib.conv_set_dynamic_batch(live)
    try:
        _lib.maxpool2d_nhwc(x, k, s, p, out=out)
    finally:
        _lib.conv_set_dynamic_batch(None)
    assert torch.equal(out[:1], y[:1]) and bool((out[1:] == -5.0).all())
