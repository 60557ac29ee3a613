"""Part-based ReID network in the shape of BPBReID (ResNet-50, last stride 1, 1x1 dim-reduce,
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

from .common import ConvBiasAct, finalize, random_init_


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


class PartBasedReID(nn.Module):
    def __init__(self, parts=6, dim=256, vis_threshold=0.5):
        super().__init__()
        self.conv1 = ConvBiasAct(3, 64, 7, 2, "relu")
        self.pool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = _layer(64, 64, 3, 1)
        self.layer2 = _layer(256, 128, 4, 2)
        self.layer3 = _layer(512, 256, 6, 2)
        self.layer4 = _layer(1024, 512, 3, 1)          # last_stride = 1 (BPBReID)
        self.reduce = ConvBiasAct(2048, dim, 1, 1, None)
        self.part_cls = nn.Conv2d(dim, parts, 1, bias=True)   # foreground + (parts-1) body parts
        self.parts, self.dim, self.vis_threshold = parts, dim, vis_threshold

    def forward(self, x):
        x = self.pool(self.conv1(x))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        f = self.reduce(x)                                   # (N, D, h, w)
        att = torch.softmax(self.part_cls(f).float(), dim=1)  # (N, K, h, w) pixel-wise part attention
        ff = f.float().flatten(2)                            # (N, D, hw)
        a = att.flatten(2)                                   # (N, K, hw)
        emb = torch.bmm(a, ff.transpose(1, 2)) / a.sum(-1, keepdim=True).clamp_min(1e-6)   # (N, K, D)
        vis = a.amax(-1) > self.vis_threshold / self.parts
        vis[:, 0] = True
        return emb.contiguous(), vis


def part_based_reid(parts=6, dim=256, device="cuda", dtype=torch.float16, channels_last=True, seed=0):
    return finalize(random_init_(PartBasedReID(parts, dim), seed), device, dtype, channels_last)
