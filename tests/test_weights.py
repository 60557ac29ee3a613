"""CPU: weight import for the drop-in (tracklab_amd/weights.py). The reference's own artefacts (rtmlib ONNX files, torchreid checkpoints) are not
available offline, so the check is a round trip: a module of this repo is exported to ONNX here, the file is read back by the dependency-free
protobuf reader, matched STRUCTURALLY against a differently initialised copy and imported -- forwards must be bit-identical. A variant whose
CSP layers evaluate their branches in the other order (as mmdet's CSPLayer does: short branch first), which permutes the ONNX nodes and would
defeat name- or position-based matching, must import identically."""
import numpy as np
import pytest
import torch

import importlib

from tracklab_amd import weights as W

Y = importlib.import_module("tracklab_amd.backbones.yolox")          # (the package re-exports a function of the same name)


def _yolox(seed, cls=None):
    m = Y.YOLOX("s", 1) if cls is None else cls("s", 1)
    Y.random_init_(m, seed)
    with torch.no_grad():                                    # non-zero biases: a swapped pair of equal-shape convolutions must show
        g = torch.Generator().manual_seed(100 + seed)
        for p in m.parameters():
            if p.dim() == 1:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    return m.eval()


@pytest.fixture(scope="module")
def exported():
    src = _yolox(0)
    x = torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(1))
    return src, x, W.export_onnx_bytes(src, (x,))


def test_protobuf_reader_returns_every_initializer_bit_exact(exported):
    src, x, blob = exported
    g = W.read_onnx(blob)
    sd = src.state_dict()
    named = {k: v for k, v in g.tensors.items() if k in sd}
    assert len(named) == len(sd)                             # do_constant_folding=False keeps the state_dict names
    for k, v in named.items():
        np.testing.assert_array_equal(v, sd[k].numpy())
    ops = {n.op for n in g.nodes}
    assert {"Conv", "Concat", "MaxPool", "Sigmoid", "Mul"} <= ops and len(g.inputs) == 1 and len(g.outputs) == 1
    conv = next(n for n in g.nodes if n.op == "Conv")
    assert conv.attrs["kernel_shape"] == [3, 3] and conv.attrs["strides"] == [1, 1]


def test_structural_import_round_trip_is_bit_identical(exported):
    src, x, blob = exported
    dst = _yolox(7)
    with torch.no_grad():
        assert not torch.equal(dst(x), src(x))
    n = W.import_onnx_weights(dst, (x,), blob)
    assert n == len(src.state_dict())
    with torch.no_grad():
        assert torch.equal(dst(x), src(x))
    for k, v in src.state_dict().items():
        assert torch.equal(dst.state_dict()[k], v), k


class _ShortFirstCSP(Y.CSPLayer):
    def forward(self, x):                                    # mmdet's CSPLayer order: short branch, then main branch + blocks
        short = self.conv2(x)
        return self.conv3(torch.cat((self.m(self.conv1(x)), short), dim=1))


def test_import_does_not_depend_on_node_order(exported, monkeypatch):
    src, x, blob = exported
    monkeypatch.setattr(Y, "CSPLayer", _ShortFirstCSP)
    variant = _yolox(0)                                      # same weights as `src`, other evaluation order -> permuted ONNX nodes
    monkeypatch.undo()
    blob2 = W.export_onnx_bytes(variant, (x,))
    order = lambda b: [n.inputs[1] for n in W.read_onnx(b).nodes if n.op == "Conv"]       # noqa: E731
    assert order(blob2) != order(blob) and sorted(order(blob2)) == sorted(order(blob))
    dst = _yolox(9)
    W.import_onnx_weights(dst, (x,), blob2)
    with torch.no_grad():
        assert torch.equal(dst(x), src(x))


def test_anonymous_initializer_names_do_not_matter(exported):
    """BatchNorm folding in mmdeploy leaves names like onnx::Conv_1035: rename every initializer of the reference graph."""
    src, x, blob = exported
    g = W.read_onnx(blob)
    ren = {k: f"onnx::Conv_{i}" for i, k in enumerate(sorted(g.tensors))}
    g.tensors = {ren[k]: v for k, v in g.tensors.items()}
    for n in g.nodes:
        n.inputs = [ren.get(i, i) for i in n.inputs]
    dst = _yolox(11)
    W.import_onnx_weights(dst, (x,), g)
    with torch.no_grad():
        assert torch.equal(dst(x), src(x))


def test_a_different_architecture_is_refused(exported):
    src, x, blob = exported
    other = Y.YOLOX("tiny", 1).eval()
    with pytest.raises(ValueError, match="structural partners|weighted nodes"):
        W.import_onnx_weights(other, (x,), blob)


def test_rtmpose_round_trip():
    """The pose network has more than convolutions: depthwise convs, channel attention, ScaleNorm, a gated attention unit (MatMul linears without
    bias, elementwise gains and per-head scale / offset): every parameter is imported through the structural match."""
    from tracklab_amd.backbones.rtmpose import rtmpose
    a = rtmpose("t", device="cpu", dtype=torch.float32, channels_last=False, seed=0)
    b = rtmpose("t", device="cpu", dtype=torch.float32, channels_last=False, seed=5)      # biases all zero: the exporter would merge them
    with torch.no_grad():
        g = torch.Generator().manual_seed(3)
        for p in a.parameters():
            if p.dim() <= 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
    x = torch.randn(1, 3, 256, 192, generator=torch.Generator().manual_seed(2))
    blob = W.export_onnx_bytes(a, (x,))
    n = W.import_onnx_weights(b, (x,), blob)
    assert n == len(a.state_dict()) and W.import_onnx_weights.unmatched == []
    for k, v in a.state_dict().items():
        assert torch.equal(v, b.state_dict()[k]), k
    with torch.no_grad():
        ya, yb = a(x), b(x)
    for p, q in zip(ya, yb):
        assert torch.equal(p, q)


def test_resnet50_batchnorm_checkpoint_folds_into_the_reid_backbone():
    """A ResNet-50 checkpoint in torchvision / torchreid naming (conv + BatchNorm with running statistics) -> backbones.reid._ResNet50."""
    import torch.nn as nn
    from tracklab_amd.backbones.reid import _ResNet50

    class Bott(nn.Module):
        def __init__(self, cin, planes, stride, down):
            super().__init__()
            self.conv1, self.bn1 = nn.Conv2d(cin, planes, 1, bias=False), nn.BatchNorm2d(planes)
            self.conv2, self.bn2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False), nn.BatchNorm2d(planes)
            self.conv3, self.bn3 = nn.Conv2d(planes, planes * 4, 1, bias=False), nn.BatchNorm2d(planes * 4)
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4)) if down else None

        def forward(self, x):
            idt = x if self.downsample is None else self.downsample(x)
            y = torch.relu(self.bn1(self.conv1(x)))
            y = torch.relu(self.bn2(self.conv2(y)))
            return torch.relu(self.bn3(self.conv3(y)) + idt)

    class Ref(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1, self.bn1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64)
            self.maxpool = nn.MaxPool2d(3, 2, 1)
            cin = 64
            for li, (planes, nb, st) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 1)), start=1):
                blocks = [Bott(cin, planes, st, True)] + [Bott(planes * 4, planes, 1, False) for _ in range(nb - 1)]
                setattr(self, f"layer{li}", nn.Sequential(*blocks))
                cin = planes * 4

        def forward(self, x):
            x = self.maxpool(torch.relu(self.bn1(self.conv1(x))))
            return self.layer4(self.layer3(self.layer2(self.layer1(x))))

    torch.manual_seed(3)
    ref = Ref().eval()
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.1); m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
    folded = W.fold_batchnorm_state_dict(ref.state_dict(), W.resnet50_bn_pairs())
    own = _ResNet50().eval()
    assert set(folded) == set(own.state_dict())
    own.load_state_dict({k: torch.from_numpy(v) for k, v in folded.items()})
    x = torch.randn(2, 3, 96, 48)
    with torch.no_grad():
        a, b = ref(x), own(x)
    assert a.shape == b.shape
    np.testing.assert_allclose(b.numpy(), a.numpy(), rtol=2e-4, atol=2e-4)


def test_load_checkpoint_entry_point(tmp_path):
    """What cfg.checkpoint of the Hip* modules accepts: an ONNX file, this repo's state_dict, a BatchNorm ResNet-50 checkpoint; a missing file is loud."""
    from tracklab_amd.backbones.reid import PartBasedReID
    src, dst = _yolox(0), _yolox(3)
    x = torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(1))
    onnx_path = tmp_path / "det.onnx"
    onnx_path.write_bytes(W.export_onnx_bytes(src, (x,)))
    rep = W.load_checkpoint(dst, onnx_path, (x,))
    assert rep["format"] == "onnx" and rep["tensors"] == len(src.state_dict())
    with torch.no_grad():
        assert torch.equal(dst(x), src(x))
    pt = tmp_path / "det.pt"
    torch.save({"state_dict": src.state_dict()}, pt)
    dst2 = _yolox(4)
    assert W.load_checkpoint(dst2, pt)["format"] == "state_dict"
    with torch.no_grad():
        assert torch.equal(dst2(x), src(x))
    with pytest.raises(FileNotFoundError):
        W.load_checkpoint(dst, tmp_path / "nope.onnx", (x,))
    # torchreid-style file: BatchNorm ResNet-50 under a prefix + head keys this network does not have
    import torch.nn as nn
    reid = PartBasedReID(6, 64).eval()
    sd = {}
    g = torch.Generator().manual_seed(0)
    for conv, bn, tw, tb in W.resnet50_bn_pairs("module.backbone."):
        w = reid.backbone.state_dict()[tw]
        sd[conv + ".weight"] = torch.randn(w.shape, generator=g) * 0.05
        for k, v in (("weight", 1.0), ("bias", 0.0), ("running_mean", 0.0), ("running_var", 1.0)):
            sd[f"{bn}.{k}"] = torch.full((w.shape[0],), v) + torch.rand(w.shape[0], generator=g) * 0.1
    sd["module.classifier.weight"] = torch.zeros(10, 2048)
    ck = tmp_path / "reid.pth.tar"
    torch.save({"state_dict": sd, "epoch": 3}, ck)
    with pytest.warns(RuntimeWarning, match="RANDOM initialisation"):                  # the part-based head is NOT in such a file: said loudly
        rep = W.load_checkpoint(reid, ck)
    assert rep["format"] == "resnet50+batchnorm" and rep["unmapped_keys"] == ["module.classifier.weight"]
    assert rep["uninitialised_parameters"] == sorted(k for k in reid.state_dict() if not k.startswith("backbone."))
    assert "reduce.conv.weight" in rep["uninitialised_parameters"] and "part_cls.weight" in rep["uninitialised_parameters"]
    with pytest.raises(ValueError, match="RANDOM initialisation"):
        W.load_checkpoint(PartBasedReID(6, 64).eval(), ck, strict_heads=True)
    exp_w, exp_b = W.fold_batchnorm(sd["module.backbone.conv1.weight"], sd["module.backbone.bn1.weight"], sd["module.backbone.bn1.bias"],
                                    sd["module.backbone.bn1.running_mean"], sd["module.backbone.bn1.running_var"])
    np.testing.assert_array_equal(reid.backbone.conv1.conv.weight.detach().numpy(), exp_w)
    np.testing.assert_array_equal(reid.backbone.conv1.bias.detach().numpy(), exp_b)


# ---------------------------------------------------------------------------------------------------------------------------------------
# an ONNX file NOT written by torch's exporter.  The model below is authored node by node in the form mmdeploy gives
# rtmlib's YOLOX files (tracklab/configs/modules/bbox_detector/yolox_rtmlib.yaml:1-7): BatchNorm folded INTO the convolutions (weight and bias
# as anonymous initializers `onnx::Conv_NNNN`, the bias as the node's third input), SiLU as Sigmoid + Mul, Focus as eight strided Slice nodes
# whose `ends` are the concrete extents (torch writes INT64_MAX), mmdet's CSPLayer order (short branch first), Resize for the up-sampling,
# the head flattened by Reshape + Concat + Transpose, prediction convolutions under mmdet's parameter names.  It is serialised by a protobuf
# WRITER that exists only in this test; the product's dependency-free reader and structural matcher have never seen its output before.
def _pb_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _pb(fno, payload):
    if isinstance(payload, int):
        return _pb_varint(fno << 3) + _pb_varint(payload)
    if isinstance(payload, str):
        payload = payload.encode()
    return _pb_varint((fno << 3) | 2) + _pb_varint(len(payload)) + bytes(payload)


def _pb_tensor(name, a):
    a = np.ascontiguousarray(a)
    dt = {np.dtype("float32"): 1, np.dtype("int64"): 7}[a.dtype]
    return b"".join(_pb(1, int(d)) for d in a.shape) + _pb(2, dt) + _pb(8, name) + _pb(9, a.tobytes())


def _pb_attr(name, v):
    if isinstance(v, (list, tuple)):
        return _pb(1, name) + b"".join(_pb(8, int(x)) for x in v) + _pb(20, 7)
    if isinstance(v, float):
        import struct
        return _pb(1, name) + _pb_varint((2 << 3) | 5) + struct.pack("<f", v) + _pb(20, 1)
    if isinstance(v, str):
        return _pb(1, name) + _pb(4, v) + _pb(20, 3)
    return _pb(1, name) + _pb(3, int(v)) + _pb(20, 2)


class _HandGraph:
    def __init__(self):
        self.nodes, self.inits, self.n, self.k = [], [], 0, 1000

    def val(self, op):
        self.n += 1
        return f"/{op}_{self.n}_output_0"

    def init(self, a, stem="onnx::Conv"):
        self.k += 1
        name = f"{stem}_{self.k}"
        self.inits.append(_pb_tensor(name, a))
        return name

    def named(self, name, a):
        self.inits.append(_pb_tensor(name, a))
        return name

    def node(self, op, ins, **attrs):
        out = self.val(op)
        self.nodes.append(_pb(1, b"") [:0] + b"".join(_pb(1, i) for i in ins) + _pb(2, out) + _pb(3, f"/{op}_{self.n}") + _pb(4, op) +
                          b"".join(_pb(5, _pb_attr(k, v)) for k, v in attrs.items()))
        return out

    def serialize(self, inp, out):
        g = b"".join(_pb(1, n) for n in self.nodes) + _pb(2, "torch_jit") + b"".join(_pb(5, t) for t in self.inits) + _pb(11, _pb(1, inp)) + _pb(12, _pb(1, out))
        return _pb(1, 7) + _pb(2, "pytorch") + _pb(8, _pb(2, 11)) + _pb(7, g)


def _hand_written_rtmlib_style_yolox(m, size):
    """YOLOX module `m` (this repo's definition, holding the weights to ship) -> bytes of an ONNX model in mmdeploy's style"""
    G = _HandGraph()
    i64 = lambda *v: np.asarray(v, np.int64)      # noqa: E731

    def conv(c, x, act=True):                     # c: backbones.common.ConvBiasAct -> Conv (BN folded: bias inside) + Sigmoid + Mul
        w, b = c.conv.weight.detach().numpy(), c.bias.detach().numpy()
        k, s = c.conv.kernel_size[0], c.conv.stride[0]
        y = G.node("Conv", [x, G.init(w), G.init(b)], dilations=[1, 1], group=1, kernel_shape=[k, k], pads=[k // 2] * 4, strides=[s, s])
        return G.node("Mul", [y, G.node("Sigmoid", [y])]) if act else y

    def csp(layer, x):                            # mmdet CSPLayer.forward: short branch FIRST, then main + blocks; cat(main, short)
        short = conv(layer.conv2, x)
        t = conv(layer.conv1, x)
        for blk in layer.m:
            u = conv(blk.conv2, conv(blk.conv1, t))
            t = G.node("Add", [u, t]) if blk.add else u
        return conv(layer.conv3, G.node("Concat", [t, short], axis=1))

    # Focus: patch_top_left = x[..., ::2, ::2] etc. -- two Slice nodes per phase (H then W), ends = the concrete extent
    def phase(dy, dx):
        a = G.node("Slice", ["input", G.init(i64(dy), "onnx::Slice"), G.init(i64(size), "onnx::Slice"), G.init(i64(2), "onnx::Slice"), G.init(i64(2), "onnx::Slice")])
        return G.node("Slice", [a, G.init(i64(dx), "onnx::Slice"), G.init(i64(size), "onnx::Slice"), G.init(i64(3), "onnx::Slice"), G.init(i64(2), "onnx::Slice")])
    bb, neck, head = m.backbone, m.neck, m.head
    x = conv(bb.stem.conv, G.node("Concat", [phase(0, 0), phase(1, 0), phase(0, 1), phase(1, 1)], axis=1))      # tl, bl, tr, br
    x = csp(bb.dark2[1], conv(bb.dark2[0], x))
    c3 = csp(bb.dark3[1], conv(bb.dark3[0], x))
    c4 = csp(bb.dark4[1], conv(bb.dark4[0], c3))
    t = conv(bb.dark5[1].conv1, conv(bb.dark5[0], c4))
    pools = [G.node("MaxPool", [t], ceil_mode=0, kernel_shape=[k, k], pads=[k // 2] * 4, strides=[1, 1]) for k in (5, 9, 13)]
    c5 = csp(bb.dark5[2], conv(bb.dark5[1].conv2, G.node("Concat", [t] + pools, axis=1)))
    up = lambda v: G.node("Resize", [v, "", G.init(np.asarray([1, 1, 2, 2], np.float32), "onnx::Resize")], coordinate_transformation_mode="asymmetric",   # noqa: E731
                          mode="nearest", nearest_mode="floor")
    fpn0 = conv(neck.lateral_conv0, c5)
    f1 = csp(neck.C3_p4, G.node("Concat", [up(fpn0), c4], axis=1))
    fpn1 = conv(neck.reduce_conv1, f1)
    p3 = csp(neck.C3_p3, G.node("Concat", [up(fpn1), c3], axis=1))
    p4 = csp(neck.C3_n3, G.node("Concat", [conv(neck.bu_conv2, p3), fpn1], axis=1))
    p5 = csp(neck.C3_n4, G.node("Concat", [conv(neck.bu_conv1, p4), fpn0], axis=1))
    outs = []
    for k, f in enumerate((p3, p4, p5)):
        s = conv(head.stems[k], f)
        cf, rf = s, s
        for c in head.cls_convs[k]:
            cf = conv(c, cf)
        for c in head.reg_convs[k]:
            rf = conv(c, rf)

        def pred(mod, v, nm):
            return G.node("Conv", [v, G.named(f"bbox_head.multi_level_conv_{nm}.{k}.weight", mod.weight.detach().numpy()),
                                   G.named(f"bbox_head.multi_level_conv_{nm}.{k}.bias", mod.bias.detach().numpy())],
                          dilations=[1, 1], group=1, kernel_shape=[1, 1], pads=[0, 0, 0, 0], strides=[1, 1])
        cls, reg, obj = pred(head.cls_preds[k], cf, "cls"), pred(head.reg_preds[k], rf, "reg"), pred(head.obj_preds[k], rf, "obj")
        o = G.node("Concat", [reg, G.node("Sigmoid", [obj]), G.node("Sigmoid", [cls])], axis=1)
        outs.append(G.node("Reshape", [o, G.init(i64(0, 0, -1), "onnx::Reshape")]))
    dets = G.node("Transpose", [G.node("Concat", outs, axis=2)], perm=[0, 2, 1])
    return G.serialize("input", dets)


def test_import_from_a_hand_written_rtmlib_style_onnx_file(tmp_path):
    src = _yolox(3)
    x = torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(5))
    blob = _hand_written_rtmlib_style_yolox(src, 64)
    path = tmp_path / "yolox_s_hand_written.onnx"
    path.write_bytes(blob)
    g = W.read_onnx(str(path))
    names = [k for k in g.tensors if k.startswith("onnx::Conv_")]
    convs = [n for n in g.nodes if n.op == "Conv"]
    assert len(convs) == len([mm for mm in src.modules() if isinstance(mm, torch.nn.Conv2d)]) and len(names) == 2 * (len(convs) - 9)
    assert all(len(n.inputs) == 3 for n in convs) and sum(n.op == "Slice" for n in g.nodes) == 8 and not set(g.tensors) & set(src.state_dict())
    dst = _yolox(13)
    with torch.no_grad():
        assert not torch.equal(dst(x), src(x))
    n = W.import_onnx_weights(dst, (x,), str(path))
    assert n == len(src.state_dict())
    with torch.no_grad():
        assert torch.equal(dst(x), src(x))
    for k, v in src.state_dict().items():
        assert torch.equal(dst.state_dict()[k], v), k
    # the entry point the wrappers use takes the same file
    dst2 = _yolox(17)
    rep = W.load_checkpoint(dst2, str(path), (x,))
    with torch.no_grad():
        assert torch.equal(dst2(x), src(x)), rep
