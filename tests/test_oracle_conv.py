"""oracle/src/conv.c (the checker of tlk_conv2d_nhwc_f32): the stated fmaf chain, and fp32 round-off distance from torch's convolution."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle


@pytest.mark.parametrize("case", [(2, 9, 7, 8, 5, 3, 1), (1, 12, 10, 4, 6, 7, 2), (2, 6, 6, 36, 10, 3, 2), (1, 5, 5, 16, 7, 1, 1)])
def test_oracle_conv_within_roundoff_of_torch_fp64(case):
    n, h, w, cin, cout, k, s = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((cout, k, k, cin)) * 0.1).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    y = oracle.conv2d_nhwc_f32(x, wt, b, None, stride=s, act="relu")
    xt, wtt = torch.from_numpy(x).permute(0, 3, 1, 2).double(), torch.from_numpy(wt).permute(0, 3, 1, 2).double()
    ref = F.relu(F.conv2d(xt, wtt, torch.from_numpy(b).double(), s, k // 2)).permute(0, 2, 3, 1).numpy()
    bound = F.conv2d(xt.abs(), wtt.abs(), torch.from_numpy(b).double().abs(), s, k // 2).permute(0, 2, 3, 1).numpy()
    assert y.shape == ref.shape
    assert np.all(np.abs(y - ref) <= 2e-6 * bound)


def test_oracle_conv_is_the_stated_chain():
    """one output element by hand: k order 0,4,1,5,2,6,3,7 per group of 8 (np.float32 fma emulated in float64: exact for one fma, then rounded)"""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 1, 1, 12)).astype(np.float32)
    wt = rng.standard_normal((1, 1, 1, 12)).astype(np.float32)
    y = oracle.conv2d_nhwc_f32(x, wt, None, None, act=None)
    acc = np.float32(0)
    xs, ws = np.zeros(16, np.float32), np.zeros(16, np.float32)
    xs[:12], ws[:12] = x.ravel(), wt.ravel()
    for g in (0, 8):
        for t in (0, 4, 1, 5, 2, 6, 3, 7):
            acc = np.float32(np.float64(xs[g + t]) * np.float64(ws[g + t]) + np.float64(acc))      # 24 x 24 bit product is exact in float64
    assert y.ravel()[0] == acc


def test_residual_and_activations():
    rng = np.random.default_rng(6)
    x = rng.standard_normal((1, 4, 4, 8)).astype(np.float32)
    wt = rng.standard_normal((3, 1, 1, 8)).astype(np.float32)
    r = rng.standard_normal((1, 4, 4, 3)).astype(np.float32)
    base = oracle.conv2d_nhwc_f32(x, wt, None, None)
    assert np.array_equal(oracle.conv2d_nhwc_f32(x, wt, None, r, act="relu"), np.maximum(base + r, 0))
    v = base + r
    np.testing.assert_allclose(oracle.conv2d_nhwc_f32(x, wt, None, r, act="silu"), v / (1 + np.exp(-v)), rtol=1e-6)
    assert np.array_equal(oracle.conv2d_nhwc_f32(x, wt, None, r, act="relu", res_after_act=True), np.maximum(base, 0) + r)      # TLK_ACT_RES_AFTER


@pytest.mark.parametrize("case", [(2, 9, 7, 8, 5), (1, 3, 2, 4, 5), (3, 6, 11, 12, 3), (1, 1, 1, 8, 3)])
def test_oracle_dwconv_within_roundoff_of_torch_fp64(case):
    n, h, w, c, k = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    wt = rng.standard_normal((k, k, c)).astype(np.float32)
    b = rng.standard_normal(c).astype(np.float32)
    y = oracle.dwconv2d_nhwc_f32(x, wt, b, "relu")
    xt, wtt = torch.from_numpy(x).permute(0, 3, 1, 2).double(), torch.from_numpy(wt).permute(2, 0, 1).unsqueeze(1).double()
    ref = F.relu(F.conv2d(xt, wtt, torch.from_numpy(b).double(), 1, k // 2, groups=c)).permute(0, 2, 3, 1).numpy()
    bound = F.conv2d(xt.abs(), wtt.abs(), torch.from_numpy(b).double().abs(), 1, k // 2, groups=c).permute(0, 2, 3, 1).numpy()
    assert np.all(np.abs(y - ref) <= 2e-6 * bound)
    v = oracle.dwconv2d_nhwc_f32(x, wt, b, None)
    np.testing.assert_allclose(oracle.dwconv2d_nhwc_f32(x, wt, b, "silu"), v / (1 + np.exp(-v)), rtol=1e-6, atol=1e-7)


def test_oracle_dwconv_is_the_stated_chain():
    """the centre pixel of a 5 x 5 image by hand: ky then kx ascending, one fma per tap"""
    rng = np.random.default_rng(8)
    x = rng.standard_normal((1, 5, 5, 4)).astype(np.float32)
    wt = rng.standard_normal((5, 5, 4)).astype(np.float32)
    y = oracle.dwconv2d_nhwc_f32(x, wt, None, None)
    for ch in range(4):
        acc = np.float32(0)
        for ky in range(5):
            for kx in range(5):
                acc = np.float32(np.float64(x[0, ky, kx, ch]) * np.float64(wt[ky, kx, ch]) + np.float64(acc))
        assert y[0, 2, 2, ch] == acc
