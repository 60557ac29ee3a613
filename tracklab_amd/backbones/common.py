"""Shared pieces of the backbones: convolution (no bias) + ONE fused libtlk epilogue pass."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


import os as _os

USE_GEMM_1X1 = _os.environ.get("TLK_CONV1X1_GEMM", "1") != "0"
USE_FUSED_GEMM = _os.environ.get("TLK_FUSED_GEMM", "1") != "0"
# fp32 (the reference's precision): every convolution + its epilogue is ONE launch of libtlk's hand-written fp32 MFMA kernel
# (tlk_conv2d_nhwc_f32, csrc/tlk_conv.hip); TLK_CONV_F32=0 restores the library route (MIOpen + separate torch epilogue passes) for A/B runs.
USE_TLK_CONV_F32 = _os.environ.get("TLK_CONV_F32", "1") != "0"
# f16 convolutions with the epilogue inside on libtlk's 16-bit MFMA kernel (tlk_conv2d_nhwc_16, csrc/tlk_conv16.hip) instead of MIOpen / CK /
# hipBLASLt + a separate epilogue pass; TLK_CONV_F16=0 restores the library route for A/B runs.
# r05: ON by default -- the direct-to-LDS kernels of csrc/tlk_conv16x.hip win the A/B at 24 frames per step (417 vs 356 frames/s) and at 1 (219 vs
# 200): profiles/r05_f16_route_ab.txt
USE_TLK_CONV_F16 = _os.environ.get("TLK_CONV_F16", "1") != "0"
# ... except the narrow 1 x 1 convolutions (Cin, Cout <= 96: CSPNeXt / CSPDarknet stage 1-2 pointwise layers, HBM-bound), where the kernel with
# the epilogue inside beats GEMM + library epilogue (RTMPose-m, 2400 crops: 0.68 vs 1.02 ms at 48 -> 48); TLK_CONV_F16_NARROW=0 opts out
USE_TLK_CONV_F16_NARROW = _os.environ.get("TLK_CONV_F16_NARROW", "1") != "0"
# bench.py's roofline pass: a list here makes every fp32 convolution record (start event, end event, algorithmic flops) around its launch
# TLK_STEM=0: the RGB stems through the implicit-GEMM kernel on a 4-channel-padded image (the r04 route) instead of the direct stem kernel
USE_TLK_STEM = _os.environ.get("TLK_STEM", "1") != "0"
CONV_TIMER = None


def param_key(*tensors):
    """Identity + content stamp of the parameters a derived tensor was built from: storage address, in-place version counter (load_state_dict /
    copy_ bump it), device, dtype.  Weight-derived caches (padded stem weight, split planes, fp32 bias, depthwise taps, Toeplitz form of
    RTMPose's last layer) store it and rebuild when it changes -- ADVICE r04: a checkpoint loaded after a first forward used to leave them stale."""
    return tuple((t.data_ptr(), t._version, str(t.device), t.dtype) for t in tensors if t is not None)
# (capacity, live) while a dynamic batch is set (tlk_conv_set_dynamic_batch): a convolution whose batch is `capacity` images computes `live` of
# them -- CONV_TIMER counts the algorithmic flops of those only
LIVE_BATCH = None


def epilogue_(x: torch.Tensor, bias: torch.Tensor, act: str | None, residual: torch.Tensor | None = None) -> torch.Tensor:
    """x = act(x + bias[c] (+ residual)). On the GPU (fp16/bf16, channels-last) this is one in-place
    ``tlk_bias_act_nhwc`` launch instead of MIOpen's bias op-tensor + activation + add passes; elsewhere
    (the CPU forward of bench.py's cpu_baseline) plain torch ops."""
    if x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and x.shape[1] % 8 == 0 \
            and x.is_contiguous(memory_format=torch.channels_last):
        from .. import _lib
        if residual is not None and not residual.is_contiguous(memory_format=torch.channels_last):
            residual = residual.contiguous(memory_format=torch.channels_last)
        return _lib.bias_act_(x, bias, act, residual)
    y = x + bias.view(1, -1, 1, 1)
    if residual is not None:
        y = y + residual
    if act == "relu":
        return F.relu(y, inplace=True)
    if act == "silu":
        return F.silu(y, inplace=True)
    return y


# the pooling half of an SPPBottleneck (5 / 9 / 13 max pools + concatenation) as ONE libtlk pass (tlk_spp_maxpool_nhwc); TLK_SPP=0 restores
# the library route (three max_pool2d launches + cat) for A/B runs
USE_TLK_SPP = _os.environ.get("TLK_SPP", "1") != "0"


def spp_concat(x: torch.Tensor, ks=(5, 9, 13)) -> torch.Tensor:
    """[x | maxpool_k(x) for k in ks] along the channels (stride 1, same size): what the second 1 x 1 convolution of an SPPBottleneck reads"""
    if USE_TLK_SPP and tuple(ks) == (5, 9, 13) and x.is_cuda and x.dtype in (torch.float16, torch.float32) \
            and x.shape[1] % (8 if x.dtype == torch.float16 else 4) == 0 and x.is_contiguous(memory_format=torch.channels_last):
        from .. import _lib
        return _lib.spp_maxpool_nhwc(x)
    return torch.cat([x] + [F.max_pool2d(x, k, 1, k // 2) for k in ks], 1)


class SplitAct:
    """An fp32 activation travelling as two float16 planes: value = scale * (hi + lo * 2**-11) (relative 2**-22), both (N, C, H, W) channels_last.
    What the split-precision convolutions (tlk_conv2d_nhwc_16, split mode: three f16 MFMAs per product pair, fp32 accumulation) read and
    write; `merge()` gives the fp32 tensor back.  `state` (r06; None = scale 1): the 2-element float32 device tensor {scale, recorded maximum}
    of the layer that wrote the planes -- a power of two >= 1 chosen so that the planes stay inside float16's range
    (SplitScales below; include/tlk.h, tlk_conv2d_nhwc_16s)."""
    __slots__ = ("hi", "lo", "state")

    def __init__(self, hi, lo, state=None):
        self.hi, self.lo, self.state = hi, lo, state

    @staticmethod
    def from_f32(x, c_out=None, state=None, dynamic_batch=False):
        from .. import _lib
        return SplitAct(*_lib.split_planes(x, c_out, state=state, dynamic_batch=dynamic_batch), state=state)

    def merge(self):
        from .. import _lib
        return _lib.merge_planes(self.hi, self.lo, scale=self.state)

    @property
    def shape(self):
        return self.hi.shape


class SplitScales:
    """The plane scales of ONE split-precision network (r06): a contiguous (n, 2) float32 device buffer of layer states {scale, recorded maximum},
    handed out to the network's ConvBiasAct layers (`_sstate`) and to its fp32 -> planes entry point.
      * every forward records each layer's largest |output| (an atomic max in the convolution's epilogue) and ends with `update()`
        (tlk_split_scale_update): the scales the NEXT forward uses -- inside a hipGraph capture it is the graph's last node;
      * `calibrate(run)`: run the network, update, and repeat while a scale grew or a maximum was not finite (an overflow upstream hides the
        layers behind it: at most one pass per saturating layer, bounded) -- once, on the first forward outside a capture.
    Scales are powers of two >= 1: a network whose activations fit float16 keeps every scale at 1 and computes the r05 planes bit for bit."""

    def __init__(self, device):
        import torch
        self.device = device
        self.n = 0
        self.buf = None
        self.changed = torch.zeros(1, dtype=torch.int32, device=device)
        self.calibrated = False
        self._mods = []

    def attach(self, modules, extra=1):
        """give every module of `modules` (ConvBiasAct layers that write planes) a state row; `extra` more rows for entry points"""
        import torch
        self._mods = list(modules)
        self.n = len(self._mods) + extra
        self.buf = torch.zeros((self.n, 2), dtype=torch.float32, device=self.device)
        self.buf[:, 0] = 1.0
        for i, m in enumerate(self._mods):
            m._sstate = self.buf[i]
        return [self.buf[len(self._mods) + j] for j in range(extra)]

    def update(self, count_changes=False):
        from .. import _lib
        _lib.split_scale_update(self.buf, self.changed if count_changes else None)

    def calibrate(self, run, max_passes=None):
        """max_passes: default = one per state + 2 (a chain in which EVERY layer saturates behind the previous one -- random-init HRNet-W32 on 0..255
        pixels needs ~40; a network that fits float16 leaves after the first)"""
        out = None
        for _ in range(max_passes if max_passes is not None else max(12, self.n + 2)):
            self.changed.zero_()
            out = run()
            self.update(count_changes=True)
            if int(self.changed.item()) == 0:
                break
        self.calibrated = True
        return out

    def scales(self):
        return self.buf[:, 0].clone()


class ConvBiasAct(nn.Module):
    """Conv2d (BatchNorm folded into weight + bias) followed by the fused epilogue."""

    def __init__(self, cin, cout, k=1, s=1, act="silu"):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, s, (k - 1) // 2, bias=False)
        self.bias = nn.Parameter(torch.zeros(cout))
        self.act = act

    def _split_weights(self, cin):
        """(hi, lo) float16 planes of the fp32 weight, input channels zero-padded to `cin` (the RGB stem arrives with 8); cached"""
        c = getattr(self, "_w_split", None)
        key = param_key(self.conv.weight) + (cin,)
        if c is None or c[0] != key:
            from .. import _lib
            w = self.conv.weight.detach().float()
            if w.shape[1] != cin:
                w = F.pad(w, (0, 0, 0, 0, 0, cin - w.shape[1]))
            c = (key, _lib.split_planes(w.contiguous(memory_format=torch.channels_last)))
            self._w_split = c
        return c[1]

    def stem16(self, x, pool=False):
        """r05: the RGB stem in f16 as one direct kernel (csrc/tlk_conv_stem16.hip), with ResNet's max-pool fused behind it when `pool`; None when
        this layer / tensor is not one it takes (the caller goes on as before)"""
        ks, st = self.conv.kernel_size, self.conv.stride
        if not (USE_TLK_STEM and USE_TLK_CONV_F16 and isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float16 and x.shape[1] == 3
                and ks in ((7, 7), (3, 3)) and st == (2, 2) and self.conv.padding == (ks[0] // 2, ks[0] // 2) and self.conv.out_channels <= 64
                and self.conv.out_channels % 8 == 0 and self.conv.weight.dtype == torch.float16
                and x.is_contiguous(memory_format=torch.channels_last)):
            return None
        if pool and (x.shape[3] + 2 * (ks[0] // 2) - ks[0]) // 2 + 1 > 64:
            return None
        from .. import _lib
        c = getattr(self, "_w_stem16", None)
        key = param_key(self.conv.weight, self.bias)
        if c is None or c[0] != key:
            c = (key, _lib.conv_stem16_pack(self.conv.weight), self.bias.detach().float())
            self._w_stem16 = c
        return _lib.conv_stem16(x, c[1], self.conv.out_channels, ks[0], c[2], self.act, pool=pool)

    def writes_slices(self, x):
        """True when forward(x, out=...) writes straight into `out` (a channel slice of a wider tensor): the libtlk convolution routes"""
        if isinstance(x, SplitAct):
            return True
        if not x.is_cuda or not x.is_contiguous(memory_format=torch.channels_last):
            return False
        if x.dtype == torch.float32:
            return USE_TLK_CONV_F32 and x.shape[1] % 4 == 0
        narrow = USE_TLK_CONV_F16_NARROW and self.conv.kernel_size == (1, 1) and self.conv.in_channels <= 96 and self.conv.out_channels <= 96
        return x.dtype == torch.float16 and (USE_TLK_CONV_F16 or narrow) and x.shape[1] % 8 == 0 and self.conv.out_channels % 8 == 0 \
            and self.conv.weight.dtype == torch.float16

    def forward(self, x, residual=None, residual_after_act=False, out=None):
        """residual_after_act: y = act(conv + bias) + residual (CSPNeXt's identity add) instead of act(conv + bias + residual) (ResNet).
        out: a channels_last tensor or channel slice of one to write into (see writes_slices; other routes compute, then copy)"""
        if out is not None and not isinstance(x, SplitAct) and not (isinstance(x, torch.Tensor) and self.writes_slices(x)):
            out.copy_(self.forward(x, residual, residual_after_act))
            return out
        if isinstance(x, SplitAct):
            # split-precision route (fp32-class results on the 16-bit MFMA): input, residual and output are (hi, lo) plane pairs
            from .. import _lib
            wh, wl = self._split_weights(x.hi.shape[1])
            of32 = getattr(self, "out_f32", False)
            st = None if of32 else getattr(self, "_sstate", None)       # r06: this layer's plane state (SplitScales), None = unscaled planes
            if CONV_TIMER is not None:
                n0, n1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n0.record(); n1.record()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            res = _lib.conv2d_nhwc_16(x.hi, wh, self.bias.float() if self.bias.dtype != torch.float32 else self.bias, self.act,
                                      residual.hi if residual is not None else None, self.conv.stride[0], self.conv.padding[0],
                                      x_lo=x.lo, weight_lo=wl, residual_lo=residual.lo if residual is not None else None,
                                      out_f32=of32, residual_after_act=residual_after_act, out=(out.hi, out.lo) if out is not None else None,
                                      in_scale=x.state, res_scale=residual.state if residual is not None else None, out_state=st)
            if CONV_TIMER is not None:
                e1.record()
                y_ = res if of32 else res[0]
                cout, cin, kh, kw = self.conv.weight.shape
                nb = LIVE_BATCH[1] if (LIVE_BATCH is not None and y_.shape[0] == LIVE_BATCH[0]) else y_.shape[0]
                # algorithmic count = the fp32 convolution this stands for (the 16-bit MFMA does three products per operand pair: bench.py's
                # roofline_split prices that); bytes: two f16 planes per tensor = 4 bytes per element, like fp32
                nbytes = 4.0 * (nb * x.hi.shape[2] * x.hi.shape[3] * cin + nb * y_.shape[2] * y_.shape[3] * cout * (2 if residual is not None else 1) + cout * cin * kh * kw)
                CONV_TIMER.append((e0, e1, n0, n1, 2.0 * nb * y_.shape[2] * y_.shape[3] * cout * cin * kh * kw, ("split", _lib.ACT[self.act], residual is not None), nbytes))
            return res if of32 else SplitAct(*res, state=st)
        if x.shape[1] == 3 and residual is None and x.dtype == torch.float16:
            y = self.stem16(x)
            if y is not None:
                return y
        narrow = USE_TLK_CONV_F16_NARROW and self.conv.kernel_size == (1, 1) and self.conv.in_channels <= 96 and self.conv.out_channels <= 96
        if (USE_TLK_CONV_F16 or narrow) and x.is_cuda and x.dtype == torch.float16 and x.shape[1] % 8 == 0 and self.conv.out_channels % 8 == 0 \
                and x.is_contiguous(memory_format=torch.channels_last) and self.conv.weight.dtype == torch.float16:
            from .. import _lib
            c = getattr(self, "_bias32", None)
            key = param_key(self.bias)
            if c is None or c[0] != key:
                c = (key, self.bias.detach().float())
                self._bias32 = c
            b32 = c[1]
            if residual is not None and not residual.is_contiguous(memory_format=torch.channels_last):
                residual = residual.contiguous(memory_format=torch.channels_last)
            weight = self.conv.weight
            if weight.shape[1] != x.shape[1]:         # input channels zero-padded by the caller (the Focus stem: 12 -> 16): the weight follows (cached)
                c = getattr(self, "_w_cpad", None)
                key = param_key(weight) + (x.shape[1],)
                if c is None or c[0] != key:
                    c = (key, F.pad(weight.detach(), (0, 0, 0, 0, 0, x.shape[1] - weight.shape[1])).contiguous(memory_format=torch.channels_last))
                    self._w_cpad = c
                weight = c[1]
            return _lib.conv2d_nhwc_16(x, weight, b32, self.act, residual, self.conv.stride[0], self.conv.padding[0],
                                       residual_after_act=residual_after_act, out=out)
        if USE_TLK_CONV_F32 and x.is_cuda and x.dtype == torch.float32 and (x.shape[1] % 4 == 0 or x.shape[1] == 3) \
                and x.is_contiguous(memory_format=torch.channels_last):
            from .. import _lib
            ks, st = self.conv.kernel_size, self.conv.stride
            if x.shape[1] == 3 and residual is None and ks in ((7, 7), (3, 3)) and st == (2, 2) and self.conv.out_channels <= 64 and USE_TLK_STEM:
                # RGB stem, direct kernel (r05, csrc/tlk_conv_stem.hip): reads the 3-channel image as it is; bit-identical to the padded route below
                weight = self.conv.weight
            elif x.shape[1] == 3:
                # RGB stem: the kernel gathers 16 B per tap, so the image gets a zero 4th channel (one small pass) and the weight a zero 4th
                # input channel (cached): zero terms in the fmaf chain, the sum is unchanged
                c = getattr(self, "_w4", None)
                key = param_key(self.conv.weight)
                if c is None or c[0] != key:
                    c = (key, F.pad(self.conv.weight.detach(), (0, 0, 0, 0, 0, 1)).contiguous(memory_format=torch.channels_last))
                    self._w4 = c
                w4 = c[1]
                x4 = F.pad(x, (0, 0, 0, 0, 0, 1)).contiguous(memory_format=torch.channels_last)
                x, weight = x4, w4
            else:
                weight = self.conv.weight
            if residual is not None and not residual.is_contiguous(memory_format=torch.channels_last):
                residual = residual.contiguous(memory_format=torch.channels_last)
            if CONV_TIMER is not None:
                n0, n1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n0.record(); n1.record()                   # an empty pair first: what the pair itself costs on this stream
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            y = _lib.conv2d_nhwc_f32(x, weight, self.bias, self.act, residual, self.conv.stride[0], self.conv.padding[0],
                                     residual_after_act=residual_after_act, out=out)
            if CONV_TIMER is not None:
                e1.record()
                cout, cin, kh, kw = self.conv.weight.shape       # the algorithmic count: 3 input channels for the RGB stem, not the padded 4
                nb = LIVE_BATCH[1] if (LIVE_BATCH is not None and y.shape[0] == LIVE_BATCH[0]) else y.shape[0]
                # algorithmic bytes: input, output (and residual) once, weights once -- fp32, live images only
                nbytes = 4.0 * (nb * x.shape[2] * x.shape[3] * cin + nb * y.shape[2] * y.shape[3] * cout * (2 if residual is not None else 1)
                                + cout * cin * kh * kw)
                CONV_TIMER.append((e0, e1, n0, n1, 2.0 * nb * y.shape[2] * y.shape[3] * cout * cin * kh * kw,
                                   (_lib.lib().tlk_conv2d_last_config(), _lib.ACT[self.act], residual is not None), nbytes))
            return y
        if residual_after_act and residual is not None:
            return self.forward(x) + residual                 # library routes fuse the residual only ahead of the activation
        if USE_GEMM_1X1 and self.conv.kernel_size == (1, 1) and self.conv.stride == (1, 1) and x.is_cuda \
                and x.is_contiguous(memory_format=torch.channels_last):
            # a channels-last 1x1 convolution IS a plain GEMM (rows = N*H*W): hand it to hipBLASLt
            n, c, h, w = x.shape
            x2 = x.permute(0, 2, 3, 1).reshape(-1, c)
            w2 = self.conv.weight.reshape(self.conv.out_channels, c)
            if USE_FUSED_GEMM and x.dtype in (torch.float16, torch.bfloat16):
                # one hipBLASLt call with bias + activation (+ residual as beta*C) inside, algorithm tuned per shape by libtlk
                # (tlk_gemm_bias_act): 1.4-2.3x faster than GEMM + epilogue pass on the ReID bottleneck shapes
                from .. import _lib
                r2 = None
                if residual is not None:
                    if not residual.is_contiguous(memory_format=torch.channels_last):
                        residual = residual.contiguous(memory_format=torch.channels_last)
                    r2 = residual.permute(0, 2, 3, 1).reshape(-1, self.conv.out_channels)
                y2 = _lib.gemm_bias_act(x2, w2, self.bias, self.act, r2)
                if y2 is not None:
                    return y2.view(n, h, w, -1).permute(0, 3, 1, 2)
            # (measured: torch's own routes to the fused epilogue -- riding the residual in as the GEMM's beta*C, or
            #  torch._addmm_activation -- take the heuristic's first algorithm, which is slower here: 206 -> 192 frames/s on config3.)
            y = F.linear(x2, w2)
            y = y.view(n, h, w, -1).permute(0, 3, 1, 2)
        else:
            y = self.conv(x)
        return epilogue_(y, self.bias, self.act, residual)


def random_init_(module: nn.Module, seed: int = 0) -> nn.Module:
    """Variance-preserving random weights (no checkpoints offline); biases zero."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            if p.dim() > 1:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / fan_in) ** 0.5)
            else:
                p.zero_()
    return module


def finalize(module: nn.Module, device, dtype, channels_last: bool) -> nn.Module:
    module = module.eval().to(device=device, dtype=dtype)
    if channels_last:
        module = module.to(memory_format=torch.channels_last)
    for p in module.parameters():
        p.requires_grad_(False)
    return module
