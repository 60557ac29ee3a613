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
