"""RTMPose-shaped top-down pose network (CSPNeXt backbone + RTMCC SimCC head) in PyTorch-ROCm.

Architecture after the model the reference downloads for config 4 (configs/modules/pose_estimator/rtmpose_rtmlib.yaml:
rtmpose-m, simcc-body7, 256x192 input): CSPNeXt-m (deepen 0.67, widen 0.75; stage channels 96/192/384/768, depthwise 5x5
blocks, channel attention, SPP in the last stage) and the RTMCC head (7x7 conv to 17 keypoint maps, 8x6 -> 256 token
embedding with ScaleNorm, one gated attention unit with s = 128 and expansion 2, two linear classifiers to 384 / 512 SimCC
bins). Weights are random-initialised (no checkpoints offline): throughput only. BatchNorm is folded (ConvBiasAct).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

import os as _os

from .common import ConvBiasAct, epilogue_, finalize, param_key, random_init_, spp_concat

# depthwise 5 x 5 (+ bias + SiLU) on libtlk's hand-written kernel (tlk_dwconv2d_nhwc); TLK_DWCONV=0 restores the library route for A/B runs
USE_TLK_DWCONV = _os.environ.get("TLK_DWCONV", "1") != "0"
# the 7 x 7 keypoint-map convolution on the 8 x 6 map as one dense GEMM (RTMPoseNet._final_maps); TLK_POSE_FINAL_GEMM=0 = the convolution
USE_FINAL_GEMM = _os.environ.get("TLK_POSE_FINAL_GEMM", "1") != "0"
# CSPLayer: the two halves of the concatenation written in place by the convolutions that produce them (libtlk routes); 0 = torch.cat
USE_SLICE_CONCAT = _os.environ.get("TLK_SLICE_CONCAT", "1") != "0"


class DWConvBiasAct(nn.Module):
    """depthwise k x k (+bias, SiLU) followed by pointwise 1x1 (+bias, SiLU): mmdet DepthwiseSeparableConvModule."""

    def __init__(self, cin, cout, k=5):
        super().__init__()
        self.dw = nn.Conv2d(cin, cin, k, 1, k // 2, groups=cin, bias=False)
        self.dw_bias = nn.Parameter(torch.zeros(cin))
        self.pw = ConvBiasAct(cin, cout, 1, 1, "silu")

    def _taps(self):
        """(k, k, C) taps-major weight + fp32 bias for libtlk's depthwise kernel; cached per device / dtype"""
        c = getattr(self, "_dw_taps", None)
        w = self.dw.weight
        key = param_key(w, self.dw_bias)
        if c is None or c[0] != key:
            k = w.shape[-1]
            c = (key, (w.detach().permute(2, 3, 0, 1).reshape(k, k, w.shape[0]).contiguous(), self.dw_bias.detach().float().contiguous()))
            self._dw_taps = c
        return c[1]

    def forward(self, x, residual=None, residual_after_act=False, out=None):
        if USE_TLK_DWCONV and x.is_cuda and x.dtype in (torch.float16, torch.float32) and x.is_contiguous(memory_format=torch.channels_last) \
                and x.shape[1] % (8 if x.dtype == torch.float16 else 4) == 0:
            # ONE launch of libtlk's depthwise kernel (bias + SiLU inside, every byte moved once) instead of MIOpen's grouped convolution +
            # an epilogue pass: 33 of the 76 ms of the f16 pose forward of config 4 were spent in those two
            from .. import _lib
            wk, b32 = self._taps()
            return self.pw(_lib.dwconv2d_nhwc(x, wk, b32, "silu"), residual, residual_after_act, out)
        return self.pw(epilogue_(self.dw(x), self.dw_bias, "silu"), residual, residual_after_act, out)


class CSPNeXtBlock(nn.Module):
    def __init__(self, c, add_identity=True):
        super().__init__()
        self.conv1 = ConvBiasAct(c, c, 3, 1, "silu")
        self.conv2 = DWConvBiasAct(c, c, 5)
        self.add = add_identity

    def forward(self, x, out=None):
        # mmdet CSPNeXtBlock.forward: out = conv2(conv1(x)); out + identity -- the add rides in the pointwise convolution's epilogue where that is a
        # libtlk kernel (after its activation), a separate pass otherwise
        return self.conv2(self.conv1(x), x if self.add else None, residual_after_act=True, out=out)


class ChannelAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.fc = nn.Conv2d(c, c, 1, bias=True)

    def forward(self, x):
        return x * F.hardsigmoid(self.fc(x.mean((2, 3), keepdim=True)))


class CSPLayer(nn.Module):
    def __init__(self, cin, cout, n, add_identity=True, attention=True):
        super().__init__()
        mid = cout // 2
        self.main = ConvBiasAct(cin, mid, 1, 1, "silu")
        self.short = ConvBiasAct(cin, mid, 1, 1, "silu")
        self.blocks = nn.Sequential(*[CSPNeXtBlock(mid, add_identity) for _ in range(n)])
        self.final = ConvBiasAct(2 * mid, cout, 1, 1, "silu")
        self.att = ChannelAttention(2 * mid) if attention else None

    def forward(self, x):
        if USE_SLICE_CONCAT and self.short.writes_slices(x) and len(self.blocks) > 0:
            # the concatenation is never copied: the last block's pointwise convolution and the shortcut write their halves of it directly
            n, _, h, w = x.shape
            mid = self.short.conv.out_channels
            y = torch.empty((n, 2 * mid, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            t = self.main(x)
            for blk in list(self.blocks)[:-1]:
                t = blk(t)
            self.blocks[-1](t, out=y[:, :mid])
            self.short(x, out=y[:, mid:])
        else:
            y = torch.cat([self.blocks(self.main(x)), self.short(x)], 1)
        if self.att is not None:
            y = self.att(y)
        return self.final(y)


class SPPBottleneck(nn.Module):
    def __init__(self, cin, cout, ks=(5, 9, 13)):
        super().__init__()
        mid = cin // 2
        self.conv1 = ConvBiasAct(cin, mid, 1, 1, "silu")
        self.ks = ks
        self.conv2 = ConvBiasAct(mid * (len(ks) + 1), cout, 1, 1, "silu")

    def forward(self, x):
        x = self.conv1(x)
        return self.conv2(spp_concat(x, self.ks))


class ScaleNorm(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.scale = dim ** -0.5
        self.eps = eps
        self.g = nn.Parameter(torch.ones(1))

    def forward(self, x):
        norm = torch.linalg.norm(x, dim=-1, keepdim=True) * self.scale
        return x / norm.clamp(min=self.eps) * self.g


class GAU(nn.Module):
    """RTMCCBlock (self-attention variant): gated attention unit with relu^2 kernel."""

    def __init__(self, dim=256, s=128, expansion=2):
        super().__init__()
        self.e, self.s = dim * expansion, s
        self.ln = ScaleNorm(dim)
        self.uv = nn.Linear(dim, 2 * self.e + s, bias=False)
        self.gamma = nn.Parameter(torch.rand(2, s))
        self.beta = nn.Parameter(torch.zeros(2, s))
        self.o = nn.Linear(self.e, dim, bias=False)
        self.res_scale = nn.Parameter(torch.ones(dim))
        self.sqrt_s = math.sqrt(s)

    def forward(self, x):
        h = F.silu(self.uv(self.ln(x)))
        u, v, base = torch.split(h, [self.e, self.e, self.s], dim=-1)
        base = base.unsqueeze(2) * self.gamma[None, None] + self.beta
        q, k = base[:, :, 0], base[:, :, 1]
        kernel = torch.square(F.relu(torch.bmm(q, k.transpose(1, 2)) / self.sqrt_s))
        return x * self.res_scale + self.o(u * torch.bmm(kernel, v))


class RTMPoseNet(nn.Module):
    def __init__(self, widen=0.75, deepen=0.67, keypoints=17, input_hw=(256, 192), simcc_split=2.0):
        super().__init__()
        c = [int(v * widen) for v in (64, 128, 256, 512, 1024)]
        n = [max(round(v * deepen), 1) for v in (3, 6, 6, 3)]
        self.stem = nn.Sequential(ConvBiasAct(3, c[0] // 2, 3, 2, "silu"), ConvBiasAct(c[0] // 2, c[0] // 2, 3, 1, "silu"),
                                  ConvBiasAct(c[0] // 2, c[0], 3, 1, "silu"))
        stages = []
        for i in range(4):
            layers = [ConvBiasAct(c[i], c[i + 1], 3, 2, "silu")]
            if i == 3:
                layers.append(SPPBottleneck(c[i + 1], c[i + 1]))
            layers.append(CSPLayer(c[i + 1], c[i + 1], n[i], add_identity=i < 3, attention=True))
            stages.append(nn.Sequential(*layers))
        self.stages = nn.Sequential(*stages)
        self.K = keypoints
        fh, fw = input_hw[0] // 32, input_hw[1] // 32
        self.final_layer = nn.Conv2d(c[4], keypoints, 7, 1, 3)
        self.mlp = nn.Sequential(ScaleNorm(fh * fw), nn.Linear(fh * fw, 256, bias=False))
        self.gau = GAU(256, 128, 2)
        self.cls_x = nn.Linear(256, int(input_hw[1] * simcc_split), bias=False)
        self.cls_y = nn.Linear(256, int(input_hw[0] * simcc_split), bias=False)

    def _final_maps(self, f):
        """final_layer (7 x 7 convolution, 768 -> 17 keypoint maps) on the 8 x 6 map, flattened to (N, K, 48).  A 7 x 7 window on an 8 x 6 map
        covers nearly all of it, so on the GPU the convolution is ONE dense GEMM of the flattened NHWC map (N, 48 * 768) with the Toeplitz form
        of the weight (K * 48, 48 * 768; zeros where a tap falls outside the map) -- the same multiply-adds, handed to hipBLASLt instead of a
        17-output-channel convolution (1.3 ms of the f16 pose forward of config 4); the output rows are (k, y, x): already the layout wanted."""
        if not (USE_FINAL_GEMM and f.is_cuda and f.is_contiguous(memory_format=torch.channels_last)):
            return self.final_layer(f).flatten(2)
        n, c, h, w = f.shape
        cache = getattr(self, "_final_gemm", None)
        wt = self.final_layer.weight
        key = param_key(wt, self.final_layer.bias) + (h, w)
        if cache is None or cache[3] != key:
            k, _, kh, kw = wt.shape
            big = torch.zeros(k, h, w, h, w, c, device=wt.device, dtype=wt.dtype)        # [k, y, x, y', x', c]
            for y in range(h):
                for x in range(w):
                    y0, y1 = max(0, y - kh // 2), min(h, y + kh // 2 + 1)
                    x0, x1 = max(0, x - kw // 2), min(w, x + kw // 2 + 1)
                    big[:, y, x, y0:y1, x0:x1, :] = wt[:, :, y0 - y + kh // 2:y1 - y + kh // 2, x0 - x + kw // 2:x1 - x + kw // 2].permute(0, 2, 3, 1)
            cache = (big.reshape(k * h * w, h * w * c).contiguous(), self.final_layer.bias.detach().repeat_interleave(h * w).contiguous(), (h, w), key)
            self._final_gemm = cache
        return F.linear(f.permute(0, 2, 3, 1).reshape(n, h * w * c), cache[0], cache[1]).view(n, self.K, h * w)

    def forward(self, x):
        """x (N, 3, 256, 192) normalised crops -> (simcc_x (N, K, 384), simcc_y (N, K, 512)) float32."""
        f = self.stages(self.stem(x))
        t = self._final_maps(f)                                  # (N, K, 48)
        t = self.gau(self.mlp(t))
        return self.cls_x(t).float().contiguous(), self.cls_y(t).float().contiguous()


def rtmpose(size="m", device="cuda", dtype=torch.float16, channels_last=True, seed=0):
    widen, deepen = {"t": (0.375, 0.167), "s": (0.5, 0.33), "m": (0.75, 0.67), "l": (1.0, 1.0)}[size]
    net = random_init_(RTMPoseNet(widen, deepen), seed)
    with torch.no_grad():                      # random_init_ zeroes every 1-D parameter (biases); the norm gains are 1-D too
        for m in net.modules():
            if isinstance(m, ScaleNorm):
                m.g.fill_(1.0)
            if isinstance(m, GAU):
                m.res_scale.fill_(1.0)
    return finalize(net, device, dtype, channels_last)
