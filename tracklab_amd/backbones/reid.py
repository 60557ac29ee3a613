"""Part-based ReID network in the shape of BPBReID (ResNet-50 with last stride 1, or HRNet-W32; 1x1 dim-reduce,
pixel-wise part classifier, attention-weighted pooling per part).

The reference runs this through the third-party torchreid fork
(tracklab/wrappers/reid/kpreid_api.py:147-182, configs/modules/reid/bpbreid.yaml: 384x128 input,
``dim_reduce_output: 256``, five body parts + foreground = K 6) and emits ``embeddings (N,K,D) f32``
and ``visibility_scores (N,K) bool``. Random-init here (no checkpoints offline); BN folded into conv weight + bias;
every conv is followed by ONE fused libtlk epilogue launch (bias + ReLU (+ residual)).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .common import ConvBiasAct, SplitAct, SplitScales, finalize, random_init_

import os as _os
USE_TLK_MAXPOOL = _os.environ.get("TLK_MAXPOOL", "1") != "0"       # 0: torch's max_pool2d (A/B runs)
# r06: the part-based head (6-channel convolution, softmax, attention pooling, visibility, hand-off gather, non-finite check) as ONE libtlk
# launch (tlk_reid_part_head); TLK_HEADS=0 restores the library convolution + torch passes for A/B runs
USE_TLK_HEADS = _os.environ.get("TLK_HEADS", "1") != "0"
# r06: power-of-two plane scales in the split-precision route (activations beyond float16's range); TLK_SPLIT_SCALES=0: the unscaled planes of r05
USE_SPLIT_SCALES = _os.environ.get("TLK_SPLIT_SCALES", "1") != "0"
# r06: HRNet's split route reduces the branches one by one instead of concatenating them (PartBasedReID._reduce_branches); 0: concatenate (A/B runs)
USE_BRANCH_REDUCE = _os.environ.get("TLK_BRANCH_REDUCE", "1") != "0"
# r06: HRNet's exact-fp32 route forms every exchange unit's receiving branch, and the concatenation in front of the head, with tlk_fuse_sum_f32 (one
# pass, bit-identical to the interpolate / add / relu / cat composition it replaces); 0: the torch passes (A/B runs)
USE_TLK_FUSE32 = _os.environ.get("TLK_FUSE32", "1") != "0"


USE_TLK_FUSE16 = _os.environ.get("TLK_FUSE16", "1") != "0"       # the same for the f16 route (tlk_fuse_sum_f16: every partial sum rounded to f16, as torch's half adds)


def _fuse32_ok(t):
    """a tensor the fused joints take: float32 (tlk_fuse_sum_f32) or float16 (tlk_fuse_sum_f16), channels_last, on the GPU"""
    return isinstance(t, torch.Tensor) and t.is_cuda and ((USE_TLK_FUSE32 and t.dtype == torch.float32) or (USE_TLK_FUSE16 and t.dtype == torch.float16)) \
        and t.shape[1] % 8 == 0 and t.is_contiguous(memory_format=torch.channels_last)


def _fuse_plain(terms, relu=False, out=None):
    from .. import _lib
    return (_lib.fuse_sum_f32 if terms[0].dtype == torch.float32 else _lib.fuse_sum_f16)(terms, relu=relu, out=out, dynamic_batch=True)


class _Bottleneck(nn.Module):
    def __init__(self, cin, planes, stride=1, down=False):
        super().__init__()
        self.c1 = ConvBiasAct(cin, planes, 1, 1, "relu")
        self.c2 = ConvBiasAct(planes, planes, 3, stride, "relu")
        self.c3 = ConvBiasAct(planes, planes * 4, 1, 1, "relu")             # ReLU applied after the residual add
        self.down = ConvBiasAct(cin, planes * 4, 1, stride, None) if down else None

    def forward(self, x):
        idt = x if self.down is None else self.down(x)
        return self.c3(self.c2(self.c1(x)), residual=idt)


def _layer(cin, planes, n, stride):
    blocks = [_Bottleneck(cin, planes, stride, True)] + [_Bottleneck(planes * 4, planes) for _ in range(n - 1)]
    return nn.Sequential(*blocks)


class _BasicBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.c1 = ConvBiasAct(c, c, 3, 1, "relu")
        self.c2 = ConvBiasAct(c, c, 3, 1, "relu")                           # ReLU applied after the residual add

    def forward(self, x):
        return self.c2(self.c1(x), residual=x)


class _HRModule(nn.Module):
    """One HRNet exchange unit: four basic blocks per branch, then every branch receives the sum of all branches brought to its resolution
    (higher resolution: 1x1 conv + nearest up-sampling; lower: a chain of stride-2 3x3 convs), ReLU."""

    def __init__(self, chans):
        super().__init__()
        n = len(chans)
        self.branches = nn.ModuleList([nn.Sequential(*[_BasicBlock(c) for _ in range(4)]) for c in chans])
        self.fuse = nn.ModuleList()
        for i in range(n):
            row = nn.ModuleList()
            for j in range(n):
                if j == i:
                    row.append(nn.Identity())
                elif j > i:
                    row.append(ConvBiasAct(chans[j], chans[i], 1, 1, None))
                else:
                    steps = [ConvBiasAct(chans[j], chans[j] if k < i - j - 1 else chans[i], 3, 2, "relu" if k < i - j - 1 else None) for k in range(i - j)]
                    row.append(nn.Sequential(*steps))
            self.fuse.append(row)

    def forward(self, xs):
        xs = [b(x) for b, x in zip(self.branches, xs)]
        if isinstance(xs[0], SplitAct):
            return self._exchange_split(xs)
        if all(_fuse32_ok(t) for t in xs):
            # exact-fp32 (and f16) route, r06: ((t0 + t1) + t2) + t3 over the up-sampled terms, then ReLU -- the loop below in ONE pass per receiving branch, same bits
            from .. import _lib
            out = []
            for i, row in enumerate(self.fuse):
                terms = [xs[j] if j == i else f(xs[j]) for j, f in enumerate(row)]
                if all(_fuse32_ok(t) and t.dtype == terms[0].dtype for t in terms):
                    out.append(_fuse_plain(terms, relu=True))
                else:
                    y = None
                    for j, t in enumerate(terms):
                        if j > i:
                            t = nn.functional.interpolate(t, size=xs[i].shape[-2:], mode="nearest")
                        y = t if y is None else y + t
                    out.append(torch.relu(y))
            return out
        out = []
        for i, row in enumerate(self.fuse):
            y = None
            for j, f in enumerate(row):
                t = f(xs[j])
                if j > i:
                    t = nn.functional.interpolate(t, size=xs[i].shape[-2:], mode="nearest")
                y = t if y is None else y + t
            out.append(torch.relu_(y) if y is not xs[i] else torch.relu(y))
        return out


    def _exchange_split(self, xs):
        """split-precision route (r06): the exchange convolutions hand plain fp32 to ONE pass per receiving branch (tlk_split_fuse_sum: the branch's
        own planes + the other branches' contributions, nearest up-sampling by index, ReLU, scaled planes out) -- the terms are summed in the
        order of the fp32 route above"""
        from .. import _lib
        states = getattr(self, "_fuse_states", None)
        out = []
        for i, row in enumerate(self.fuse):
            terms = [(xs[j].hi, xs[j].lo, xs[j].state) if j == i else f(xs[j]) for j, f in enumerate(row)]
            st = states[i] if states is not None else None
            out.append(SplitAct(*_lib.split_fuse_sum(terms, relu=True, out_state=st, dynamic_batch=True), state=st))
        return out

    def exchange_outputs_f32(self):
        """mark the LAST convolution of every exchange path as writing plain fp32 in the split route (its output is a term of the fused sum)"""
        for i, row in enumerate(self.fuse):
            for j, f in enumerate(row):
                if j != i:
                    (f if isinstance(f, ConvBiasAct) else f[-1]).out_f32 = True


class HRNetW32(nn.Module):
    """HRNet-W32 (Sun et al., CVPR 2019; the `backbone: "hrnet32"` of tracklab/configs/modules/reid/bpbreid.yaml:53, built by the third-party
    torchreid fork -- not vendored, so the definition follows the paper: stem to 1/4 resolution, a bottleneck stage, then 1 + 4 + 3 exchange
    modules over 2 / 3 / 4 branches of 32 / 64 / 128 / 256 channels). Output: all branches up-sampled to 1/4 resolution and concatenated,
    480 channels -- the high-resolution representation the part-based head pools over."""
    out_channels = 480

    def __init__(self):
        super().__init__()
        self.stem = nn.Sequential(ConvBiasAct(3, 64, 3, 2, "relu"), ConvBiasAct(64, 64, 3, 2, "relu"))
        self.layer1 = nn.Sequential(_Bottleneck(64, 64, 1, True), *[_Bottleneck(256, 64) for _ in range(3)])
        c = (32, 64, 128, 256)
        self.t1 = nn.ModuleList([ConvBiasAct(256, c[0], 3, 1, "relu"), ConvBiasAct(256, c[1], 3, 2, "relu")])
        self.stage2 = nn.ModuleList([_HRModule(c[:2])])
        self.t2 = ConvBiasAct(c[1], c[2], 3, 2, "relu")
        self.stage3 = nn.ModuleList([_HRModule(c[:3]) for _ in range(4)])
        self.t3 = ConvBiasAct(c[2], c[3], 3, 2, "relu")
        self.stage4 = nn.ModuleList([_HRModule(c) for _ in range(3)])

    def hr_modules(self):
        return list(self.stage2) + list(self.stage3) + list(self.stage4)

    def forward(self, x, split=False, concat=True):
        """concat=False (split route): the list of branches is returned as it is -- PartBasedReID reduces them branch by branch"""
        if split:
            # split-precision route (r06): the RGB convolution in EXACT fp32 (direct stem kernel), (hi, lo) planes from there on
            x = SplitAct.from_f32(self.stem[0](x), state=getattr(self, "_entry_state", None), dynamic_batch=True)
            x = self.layer1(self.stem[1](x))
        else:
            x = self.layer1(self.stem(x))
        xs = [t(x) for t in self.t1]
        for m in self.stage2:
            xs = m(xs)
        xs = xs + [self.t2(xs[-1])]
        for m in self.stage3:
            xs = m(xs)
        xs = xs + [self.t3(xs[-1])]
        for m in self.stage4:
            xs = m(xs)
        if not concat:
            return xs
        if isinstance(xs[0], SplitAct):
            # the concatenation brings four tensors with four plane scales onto ONE: each branch is re-split into its channel slice (one pass each)
            from .. import _lib
            n, _, h, w = xs[0].shape
            hi = torch.empty((n, self.out_channels, h, w), dtype=torch.float16, device=xs[0].hi.device, memory_format=torch.channels_last)
            lo = torch.empty_like(hi)
            st, off = getattr(self, "_cat_state", None), 0
            for t in xs:
                c = t.shape[1]
                _lib.split_fuse_sum([(t.hi, t.lo, t.state)], out=(hi[:, off:off + c], lo[:, off:off + c]), out_state=st, dynamic_batch=True)
                off += c
            return SplitAct(hi, lo, st)
        if all(_fuse32_ok(t) for t in xs):
            # exact-fp32 route, r06: up-sampling + concatenation in one pass per branch (a copy: the bits of torch.cat over the interpolated tensors)
            from .. import _lib
            n, _, h, w = xs[0].shape
            y = torch.empty((n, self.out_channels, h, w), dtype=xs[0].dtype, device=xs[0].device, memory_format=torch.channels_last)
            off = 0
            for t in xs:
                _fuse_plain([t], out=y[:, off:off + t.shape[1]])
                off += t.shape[1]
            return y
        size = xs[0].shape[-2:]
        return torch.cat([xs[0]] + [nn.functional.interpolate(t, size=size, mode="nearest") for t in xs[1:]], dim=1)


class _ResNet50(nn.Module):
    out_channels = 2048

    def __init__(self):
        super().__init__()
        self.conv1 = ConvBiasAct(3, 64, 7, 2, "relu")
        self.pool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = _layer(64, 64, 3, 1)
        self.layer2 = _layer(256, 128, 4, 2)
        self.layer3 = _layer(512, 256, 6, 2)
        self.layer4 = _layer(1024, 512, 3, 1)          # last_stride = 1 (BPBReID)

    def forward(self, x, split=False):
        if split and USE_TLK_MAXPOOL and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last):
            # split-precision route, r05: the RGB stem and its pool in EXACT fp32 (direct stem kernel 5.4 ms + one pooling pass; the split-mode
            # stem -- K = 7 * 7 * 8 padded channels on the r04 kernel, merge, torch's pool, split again -- took ~15 ms and was less exact),
            # the (hi, lo) planes start behind the pool
            from .. import _lib
            y = SplitAct.from_f32(_lib.maxpool2d_nhwc(self.conv1(x), 3, 2, 1), state=getattr(self, "_entry_state", None), dynamic_batch=True)
            return self.layer4(self.layer3(self.layer2(self.layer1(y))))
        if split:
            x = SplitAct.from_f32(x, 8)
        if isinstance(x, torch.Tensor) and x.dtype == torch.float16 and USE_TLK_MAXPOOL:
            y = self.conv1.stem16(x, pool=True)              # r05: stem + bias + ReLU + max-pool in one kernel (the 192 x 64 map is never written)
            if y is not None:
                return self.layer4(self.layer3(self.layer2(self.layer1(y))))
        x = self.conv1(x)
        if isinstance(x, SplitAct):            # split-precision route: the max-pool runs on the merged fp32 tensor (exact: max commutes with the split)
            x = SplitAct.from_f32(self.pool(x.merge()))
        elif USE_TLK_MAXPOOL and x.is_cuda and x.dtype in (torch.float32, torch.float16) and x.is_contiguous(memory_format=torch.channels_last):
            from .. import _lib
            x = _lib.maxpool2d_nhwc(x, 3, 2, 1)            # r05: one hand-written pass (torch's nhwc max-pool took 2.4-2.6 ms of a step)
        else:
            x = self.pool(x)
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


class PartBasedReID(nn.Module):
    """arch "resnet50" (default: what the benches are quoted on) or "hrnet32" (the backbone bpbreid.yaml names)."""

    def __init__(self, parts=6, dim=256, vis_threshold=0.5, arch="resnet50"):
        super().__init__()
        if arch not in ("resnet50", "hrnet32"):
            raise ValueError(f"PartBasedReID: unknown backbone {arch!r} (resnet50, hrnet32)")
        self.backbone = _ResNet50() if arch == "resnet50" else HRNetW32()
        self.reduce = ConvBiasAct(self.backbone.out_channels, dim, 1, 1, None)
        self.part_cls = nn.Conv2d(dim, parts, 1, bias=True)   # foreground + (parts-1) body parts
        self.parts, self.dim, self.vis_threshold, self.arch = parts, dim, vis_threshold, arch

    def features(self, x):
        """backbone + dimension reduction: (N, D, h, w) feature map the head pools over"""
        if getattr(self, "split_precision", False) and x.is_cuda and x.dtype == torch.float32 \
                and (self.arch == "resnet50" or x.is_contiguous(memory_format=torch.channels_last)):
            # fp32 weights, fp32-class arithmetic on the 16-bit MFMA: every convolution of the backbone behind the stem in split mode
            # (csrc/tlk_conv16x.hip; the stem + pool in exact fp32), `reduce` hands fp32 back to the head below
            self.reduce.out_f32 = True
            # r06: scaled planes -- every layer's largest |output| is recorded while it runs and the power-of-two scale of its planes follows it
            # (common.SplitScales), so activations beyond float16's 65504 no longer saturate.  First forward outside a capture: calibration
            # (run, update, repeat while a scale grew); afterwards one run + the update as the forward's last launch
            sc = getattr(self, "_split_scales", None)
            if sc is None and USE_SPLIT_SCALES:
                sc = SplitScales(x.device)
                first = self.backbone.conv1 if self.arch == "resnet50" else self.backbone.stem[0]
                layers = [m for m in self.backbone.modules() if isinstance(m, ConvBiasAct) and m is not first]
                hrm = self.backbone.hr_modules() if self.arch == "hrnet32" else []
                extra = sc.attach(layers, extra=2 + sum(len(m.fuse) for m in hrm))
                self.backbone._entry_state = extra[0]
                self.backbone._cat_state = extra[1]
                k = 2
                for m in hrm:           # HRNet: one state per receiving branch of every exchange unit
                    m._fuse_states = extra[k:k + len(m.fuse)]
                    k += len(m.fuse)
                self._split_scales = sc
            if self.arch == "hrnet32" and not getattr(self, "_hr_split_ready", False):
                for m in self.backbone.hr_modules():
                    m.exchange_outputs_f32()
                self._hr_split_ready = True
            if self.arch == "hrnet32" and USE_BRANCH_REDUCE:
                run = lambda: self._reduce_branches(self.backbone(x, split=True, concat=False))      # noqa: E731
            else:
                run = lambda: self.reduce(self.backbone(x, split=True))      # noqa: E731
            if sc is None:
                return run()
            if not sc.calibrated and not torch.cuda.is_current_stream_capturing():
                return sc.calibrate(run)
            f = run()
            sc.update()
            return f
        return self.reduce(self.backbone(x))                 # (N, D, h, w)

    def _reduce_branches(self, xs):
        """HRNet, split route (r06): the 1 x 1 dimension reduction over the concatenated branches WITHOUT the concatenation.  A 1 x 1 convolution
        commutes with nearest up-sampling and is linear in its input channels: reduce(cat(up(x_j))) = W_0 x_0 + sum_j up(W_j x_j) + b.  The
        low-resolution branches are reduced at THEIR resolution (1/4, 1/16, 1/64 of the pixels), summed onto the high-resolution grid as one plane
        pair (tlk_split_fuse_sum), and ride into the high-resolution branch's convolution as its residual: the 480-channel tensor (13 GB of planes
        at 2400 crops, written once and read once) never exists.  The sum is taken in a different order than the concatenated convolution's --
        fp32 round-off, inside the split route's tolerance (tests/test_gpu_split_hrnet.py)."""
        from .. import _lib
        from .common import param_key
        conv, bias = self.reduce.conv, self.reduce.bias
        c = getattr(self, "_reduce_parts", None)
        key = param_key(conv.weight, bias) + tuple(t.shape[1] for t in xs)
        if c is None or c[0] != key:
            parts, off = [], 0
            for t in xs:
                w = conv.weight.detach().float()[:, off:off + t.shape[1]].contiguous(memory_format=torch.channels_last)
                parts.append(_lib.split_planes(w))
                off += t.shape[1]
            c = (key, parts, bias.detach().float())
            self._reduce_parts = c
        parts, b32 = c[1], c[2]
        low = [_lib.conv2d_nhwc_16(t.hi, parts[j][0], None, None, x_lo=t.lo, weight_lo=parts[j][1], out_f32=True, in_scale=t.state)
               for j, t in enumerate(xs) if j > 0]
        n, _, h, w = xs[0].shape
        st = getattr(self.backbone, "_cat_state", None)
        rh = torch.empty((n, conv.out_channels, h, w), dtype=torch.float16, device=low[0].device, memory_format=torch.channels_last)
        rl = torch.empty_like(rh)
        _lib.split_fuse_sum(low, out=(rh, rl), out_state=st, dynamic_batch=True)
        return _lib.conv2d_nhwc_16(xs[0].hi, parts[0][0], b32, self.reduce.act, rh, x_lo=xs[0].lo, weight_lo=parts[0][1], residual_lo=rl, out_f32=True,
                                   in_scale=xs[0].state, res_scale=st)

    def fused_head_ok(self, f):
        return USE_TLK_HEADS and f.is_cuda and f.dtype in (torch.float32, torch.float16) and f.shape[1] % 8 == 0 and f.shape[1] <= 512 \
            and self.parts <= 8 and f.is_contiguous(memory_format=torch.channels_last)

    def head(self, f, counts=None, slot_base=None, max_dets=0, out_emb=None, out_vis=None, flag=None):
        """(N, D, h, w) feature map -> (emb (rows, K, D) float32, vis (rows, K) bool).  On the GPU one libtlk launch; with counts (+ slot_base:
        dense batch) the rows come out in the tracker's (frame, slot) layout, padding rows zero, `flag` set when a live embedding is not finite."""
        if self.fused_head_ok(f):
            from .. import _lib
            from .common import param_key
            c = getattr(self, "_head_w", None)
            key = param_key(self.part_cls.weight, self.part_cls.bias)
            if c is None or c[0] != key:
                c = (key, self.part_cls.weight.detach().reshape(self.parts, -1).float().contiguous(), self.part_cls.bias.detach().float().contiguous())
                self._head_w = c
            emb, vis = _lib.reid_part_head(f, c[1], c[2], self.vis_threshold / self.parts, counts, slot_base, max_dets, out_emb, out_vis, flag)
            rows = emb.numel() // (self.parts * self.dim)
            return emb.view(rows, self.parts, self.dim), (vis.view(torch.bool) if vis.dtype == torch.uint8 else vis).view(rows, self.parts)
        assert counts is None and slot_base is None, "the (frame, slot) hand-off layout is written by the libtlk head only"
        att = torch.softmax(self.part_cls(f).float(), dim=1)  # (N, K, h, w) pixel-wise part attention
        ff = f.float().flatten(2)                            # (N, D, hw)
        a = att.flatten(2)                                   # (N, K, hw)
        emb = torch.bmm(a, ff.transpose(1, 2)) / a.sum(-1, keepdim=True).clamp_min(1e-6)   # (N, K, D)
        vis = a.amax(-1) > self.vis_threshold / self.parts
        vis[:, 0] = True
        return emb.contiguous(), vis

    def forward(self, x):
        return self.head(self.features(x))


def part_based_reid(parts=6, dim=256, device="cuda", dtype=torch.float16, channels_last=True, seed=0, arch="resnet50", split_precision=False):
    """split_precision (with dtype float32; ResNet-50, and HRNet-W32 since r06): the backbone's convolutions run in split mode -- fp32 values as (hi, lo) float16 pairs,
    three f16 MFMAs per product pair, fp32 accumulation: fp32-class results at ~5x the fp32 MFMA rate (csrc/tlk_conv16.hip)."""
    m = finalize(random_init_(PartBasedReID(parts, dim, arch=arch), seed), device, dtype, channels_last)
    m.split_precision = bool(split_precision)
    return m
