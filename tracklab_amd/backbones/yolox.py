"""YOLOX (CSPDarknet + PAFPN + decoupled head), inference form with BN folded into the convs.

The reference runs this network through third-party ``rtmlib.YOLOX`` on ONNXRuntime
(tracklab/wrappers/bbox_detector/rtmlib_api.py:21,30; model zoo in
tracklab/configs/modules/bbox_detector/yolox_rtmlib*.yaml). Architecture constants follow the public
YOLOX definition: s = (depth .33, width .50), m = (.67, .75), l = (1, 1), x = (1.33, 1.25).
``forward`` takes the letterboxed 0..255 image (NCHW logical shape, any memory format) or, with
``focused=True``, the space-to-depth tensor (B, 12, S/2, S/2) produced directly by
``tlk_letterbox_u8(TLK_FOCUS_NHWC)``; it returns the raw head tensor (B, A, 5+C) float32 that
``tlk_yolox_decode_nms`` consumes (xy/wh undecoded, obj/cls after sigmoid).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .common import ConvBiasAct as Conv
from .common import SplitAct, finalize, random_init_, spp_concat

import os as _os

# CSPLayer: the two halves of the concatenation written in place by the convolutions that produce them (libtlk routes); 0 = torch.cat
USE_SLICE_CONCAT = _os.environ.get("TLK_SLICE_CONCAT", "1") != "0"

# r06: the nine 1 / 4 / num_classes-channel prediction convolutions + sigmoid / cat / flatten / permute / cast of the head as ONE libtlk launch
# (tlk_yolox_head_nhwc); TLK_HEADS=0 restores the library convolutions + torch glue for A/B runs
USE_TLK_HEADS = _os.environ.get("TLK_HEADS", "1") != "0"
USE_TLK_FOCUS16 = _os.environ.get("TLK_FOCUS16", "1") != "0"       # 0: the f16 Focus stem on the library route (A/B runs)

SIZES = {"tiny": (0.33, 0.375), "s": (0.33, 0.50), "m": (0.67, 0.75), "l": (1.0, 1.0), "x": (1.33, 1.25)}


# r06: split-precision route (fp32 values as (hi, lo) float16 plane pairs, common.SplitAct): the operators between the convolutions act on both
# planes -- concatenation and nearest up-sampling move elements, so they commute with the split exactly; max pooling does not (it compares
# VALUES), so the SPP block merges, pools in fp32 and splits again
def _cat(ts):
    if isinstance(ts[0], SplitAct):
        return SplitAct(torch.cat([t.hi for t in ts], 1), torch.cat([t.lo for t in ts], 1))
    return torch.cat(ts, 1)


def _empty_like_wide(x, channels):
    mk = lambda t: torch.empty((t.shape[0], channels, t.shape[2], t.shape[3]), dtype=t.dtype, device=t.device, memory_format=torch.channels_last)      # noqa: E731
    return SplitAct(mk(x.hi), mk(x.lo)) if isinstance(x, SplitAct) else mk(x)


def _chan(t, a, b):
    return SplitAct(t.hi[:, a:b], t.lo[:, a:b]) if isinstance(t, SplitAct) else t[:, a:b]


class Bottleneck(nn.Module):
    def __init__(self, cin, cout, shortcut=True, expansion=0.5):
        super().__init__()
        hidden = int(cout * expansion)
        self.conv1 = Conv(cin, hidden, 1)
        self.conv2 = Conv(hidden, cout, 3)
        self.add = shortcut and cin == cout

    def forward(self, x, out=None):
        # YOLOX Bottleneck: y = conv2(conv1(x)); y + x -- the add rides in conv2's epilogue (after its activation) where that is a libtlk kernel
        return self.conv2(self.conv1(x), x if self.add else None, residual_after_act=True, out=out)


class CSPLayer(nn.Module):
    def __init__(self, cin, cout, n=1, shortcut=True, expansion=0.5):
        super().__init__()
        hidden = int(cout * expansion)
        self.conv1 = Conv(cin, hidden, 1)
        self.conv2 = Conv(cin, hidden, 1)
        self.conv3 = Conv(2 * hidden, cout, 1)
        self.m = nn.Sequential(*[Bottleneck(hidden, hidden, shortcut, 1.0) for _ in range(n)])

    def forward(self, x):
        if USE_SLICE_CONCAT and (isinstance(x, SplitAct) or self.conv2.writes_slices(x)) and len(self.m) > 0:
            # the concatenation is never copied: the last bottleneck and the shortcut convolution write their halves of it directly
            hid = self.conv2.conv.out_channels
            y = _empty_like_wide(x, 2 * hid)
            t = self.conv1(x)
            for blk in list(self.m)[:-1]:
                t = blk(t)
            self.m[-1](t, out=_chan(y, 0, hid))
            self.conv2(x, out=_chan(y, hid, 2 * hid))
            return self.conv3(y)
        t = self.conv1(x)
        for blk in self.m:
            t = blk(t)
        return self.conv3(_cat((t, self.conv2(x))))


class SPPBottleneck(nn.Module):
    def __init__(self, cin, cout, ks=(5, 9, 13)):
        super().__init__()
        hidden = cin // 2
        self.conv1 = Conv(cin, hidden, 1)
        self.ks = tuple(ks)
        self.conv2 = Conv(hidden * (len(ks) + 1), cout, 1)

    def forward(self, x):
        x = self.conv1(x)
        if isinstance(x, SplitAct):           # max pooling compares values: pooled in fp32 (exact), split again
            return self.conv2(SplitAct.from_f32(spp_concat(x.merge(), self.ks)))
        return self.conv2(spp_concat(x, self.ks))


class Focus(nn.Module):
    def __init__(self, cin, cout, k=3):
        super().__init__()
        self.conv = Conv(cin * 4, cout, k)

    def forward(self, x, focused=False, split=False):
        if not focused:
            tl, tr = x[..., ::2, ::2], x[..., ::2, 1::2]
            bl, br = x[..., 1::2, ::2], x[..., 1::2, 1::2]
            x = torch.cat((tl, bl, tr, br), dim=1)
        if split:                             # the 12-channel space-to-depth image as planes of 16 channels (the 16-bit kernels take 16-byte channel groups)
            x = SplitAct.from_f32(x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last), 16)
        elif USE_TLK_FOCUS16 and x.is_cuda and x.dtype == torch.float16 and x.shape[1] == 12 and self.conv.conv.weight.dtype == torch.float16:
            # r06: f16 -- 12 channels are one and a half 16-byte groups; padded to 16 (one small pass over the letterboxed image) the stem runs on libtlk's
            # 16-bit kernel with bias + SiLU inside, instead of a library convolution + epilogue pass: the last library convolution of the detector
            x = nn.functional.pad(x, (0, 0, 0, 0, 0, 4)).contiguous(memory_format=torch.channels_last)
        return self.conv(x)


class CSPDarknet(nn.Module):
    def __init__(self, dep, wid):
        super().__init__()
        b, d = int(wid * 64), max(round(dep * 3), 1)
        self.stem = Focus(3, b)
        self.dark2 = nn.Sequential(Conv(b, b * 2, 3, 2), CSPLayer(b * 2, b * 2, d))
        self.dark3 = nn.Sequential(Conv(b * 2, b * 4, 3, 2), CSPLayer(b * 4, b * 4, d * 3))
        self.dark4 = nn.Sequential(Conv(b * 4, b * 8, 3, 2), CSPLayer(b * 8, b * 8, d * 3))
        self.dark5 = nn.Sequential(Conv(b * 8, b * 16, 3, 2), SPPBottleneck(b * 16, b * 16),
                                   CSPLayer(b * 16, b * 16, d, shortcut=False))

    def forward(self, x, focused=False, split=False):
        x = self.dark2(self.stem(x, focused, split))
        c3 = self.dark3(x)
        c4 = self.dark4(c3)
        return c3, c4, self.dark5(c4)


class PAFPN(nn.Module):
    def __init__(self, dep, wid):
        super().__init__()
        c3, c4, c5 = int(256 * wid), int(512 * wid), int(1024 * wid)
        n = round(3 * dep)
        self.up = nn.Upsample(scale_factor=2, mode="nearest")
        self.lateral_conv0 = Conv(c5, c4, 1)
        self.C3_p4 = CSPLayer(2 * c4, c4, n, False)
        self.reduce_conv1 = Conv(c4, c3, 1)
        self.C3_p3 = CSPLayer(2 * c3, c3, n, False)
        self.bu_conv2 = Conv(c3, c3, 3, 2)
        self.C3_n3 = CSPLayer(2 * c3, c4, n, False)
        self.bu_conv1 = Conv(c4, c4, 3, 2)
        self.C3_n4 = CSPLayer(2 * c4, c5, n, False)

    def forward(self, feats):
        x2, x1, x0 = feats
        up = lambda t: SplitAct(self.up(t.hi), self.up(t.lo)) if isinstance(t, SplitAct) else self.up(t)      # noqa: E731
        fpn0 = self.lateral_conv0(x0)
        f1 = self.C3_p4(_cat([up(fpn0), x1]))
        fpn1 = self.reduce_conv1(f1)
        p3 = self.C3_p3(_cat([up(fpn1), x2]))
        p4 = self.C3_n3(_cat([self.bu_conv2(p3), fpn1]))
        p5 = self.C3_n4(_cat([self.bu_conv1(p4), fpn0]))
        return p3, p4, p5


class Head(nn.Module):
    def __init__(self, num_classes, wid):
        super().__init__()
        c = int(256 * wid)
        ins = [int(256 * wid), int(512 * wid), int(1024 * wid)]
        self.stems = nn.ModuleList([Conv(i, c, 1) for i in ins])
        self.cls_convs = nn.ModuleList([nn.Sequential(Conv(c, c, 3), Conv(c, c, 3)) for _ in ins])
        self.reg_convs = nn.ModuleList([nn.Sequential(Conv(c, c, 3), Conv(c, c, 3)) for _ in ins])
        self.cls_preds = nn.ModuleList([nn.Conv2d(c, num_classes, 1) for _ in ins])
        self.reg_preds = nn.ModuleList([nn.Conv2d(c, 4, 1) for _ in ins])
        self.obj_preds = nn.ModuleList([nn.Conv2d(c, 1, 1) for _ in ins])

    def _packed_preds(self):
        """per level: (5 + num_classes, C) float32 rows [reg 0..3 | obj | cls...] and their biases, cached on the parameters' version"""
        from .common import param_key
        ps = [t for k in range(len(self.stems)) for m in (self.reg_preds[k], self.obj_preds[k], self.cls_preds[k]) for t in (m.weight, m.bias)]
        key = param_key(*ps)
        c = getattr(self, "_pk", None)
        if c is None or c[0] != key:
            ws, bs = [], []
            for k in range(len(self.stems)):
                mods = (self.reg_preds[k], self.obj_preds[k], self.cls_preds[k])
                ws.append(torch.cat([m.weight.detach().reshape(m.out_channels, -1) for m in mods], 0).float().contiguous())
                bs.append(torch.cat([m.bias.detach() for m in mods], 0).float().contiguous())
            c = (key, ws, bs)
            self._pk = c
        return c[1], c[2]

    def forward(self, feats):
        cfs, rfs = [], []
        for k, x in enumerate(feats):
            x = self.stems[k](x)
            if isinstance(x, SplitAct):       # split route: the branches' last convolutions hand fp32 to the prediction kernel
                self.cls_convs[k][-1].out_f32 = True
                self.reg_convs[k][-1].out_f32 = True
            cfs.append(self.cls_convs[k](x))
            rfs.append(self.reg_convs[k](x))
        x0 = cfs[0]
        if USE_TLK_HEADS and x0.is_cuda and x0.dtype in (torch.float32, torch.float16) and x0.shape[1] % (8 if x0.dtype == torch.float16 else 4) == 0 \
                and x0.shape[1] <= 1024 and all(t.is_contiguous(memory_format=torch.channels_last) for t in cfs + rfs):
            from .. import _lib
            ws, bs = self._packed_preds()
            return _lib.yolox_head(cfs, rfs, ws, bs, self.cls_preds[0].out_channels)      # (B, A, 5+C) float32, one launch
        outs = []
        for k, (cf, rf) in enumerate(zip(cfs, rfs)):
            o = torch.cat([self.reg_preds[k](rf), self.obj_preds[k](rf).sigmoid(), self.cls_preds[k](cf).sigmoid()], 1)
            outs.append(o.flatten(2))
        return torch.cat(outs, 2).permute(0, 2, 1).float().contiguous()      # (B, A, 5+C)


class YOLOX(nn.Module):
    def __init__(self, size="s", num_classes=1):
        super().__init__()
        dep, wid = SIZES[size]
        self.backbone = CSPDarknet(dep, wid)
        self.neck = PAFPN(dep, wid)
        self.head = Head(num_classes, wid)
        self.size, self.num_classes = size, num_classes

    def forward(self, x, focused=False, split=False):
        """split (r06; fp32 weights and input on the GPU): every convolution in split-precision mode -- (hi, lo) float16 plane pairs, three f16 MFMAs
        per product pair, fp32 accumulation (csrc/tlk_conv16*.hip) -- fp32-class results; the head's predictions come out in fp32 as always"""
        return self.head(self.neck(self.backbone(x, focused, split and x.is_cuda and x.dtype == torch.float32)))


def yolox(size="s", num_classes=1, device="cuda", dtype=torch.float16, channels_last=True, seed=0):
    """Random-init (no checkpoints offline) YOLOX-`size`, eval mode, on `device` in `dtype`."""
    return finalize(random_init_(YOLOX(size, num_classes), seed), device, dtype, channels_last)
