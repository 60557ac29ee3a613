"""Weight import for the drop-in: the reference's artefacts -> state_dicts of this repo's backbones.

The reference loads its detector / pose networks as ONNX files through third-party rtmlib
(tracklab/configs/modules/bbox_detector/yolox_rtmlib.yaml:1-7, configs/modules/pose_estimator/rtmlib.yaml) and its ReID network as a torchreid
checkpoint (tracklab/wrappers/reid/kpreid_api.py:133-161).  Neither `onnx` nor torchreid is needed here:

* `read_onnx(path)` parses the protobuf wire format itself (ModelProto -> GraphProto: nodes, initializers, Constant tensors) -- 150 lines, no
  dependency.
* `match_convolutions(ref_graph, own_graph)` pairs the weighted nodes (Conv / Gemm / MatMul) of two graphs of the SAME architecture by
  STRUCTURE, not by name or position: every weighted node gets a signature built from its own attributes (kernel, stride, groups, channels) and,
  recursively, from the signatures of the weighted nodes that feed it, seen through the unweighted operators in between (activations however they
  are decomposed, ordered Concat slots, Add, pooling, resize, slicing).  Export order, initializer naming (`onnx::Conv_1035` after BatchNorm
  folding) and the order in which a CSP layer evaluates its two branches do not matter.
* `import_onnx_weights(module, example_inputs, path)` exports `module` itself to ONNX in memory (TorchScript exporter; its initializers are named
  like the state_dict), matches the two graphs and copies every matched tensor into the module.
* `fold_batchnorm_state_dict` + `load_torchreid_resnet50` turn a ResNet-50 checkpoint with BatchNorm layers (torchvision / torchreid naming)
  into the folded conv + bias form of `backbones.reid`.

Offline check (tests/test_weights.py): a module exported here, re-imported into a differently initialised copy -- also from a variant whose CSP
layers run their branches in the other order, which permutes the ONNX nodes -- gives bit-identical forwards.
"""
from __future__ import annotations

import io
import struct
from dataclasses import dataclass, field

import numpy as np

# ------------------------------------------------------------------------------------------------ protobuf wire format
_WT_VARINT, _WT_I64, _WT_LEN, _WT_I32 = 0, 1, 2, 5


def _varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _fields(buf):
    """(field number, wire type, value) of one message; LEN values are memoryviews (no copy of 100 MB weight blobs)."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == _WT_VARINT:
            v, pos = _varint(buf, pos)
        elif wt == _WT_I64:
            v = bytes(buf[pos:pos + 8]); pos += 8
        elif wt == _WT_LEN:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]; pos += ln
        elif wt == _WT_I32:
            v = bytes(buf[pos:pos + 4]); pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fno, wt, v


def _signed(v):
    return v - (1 << 64) if v >= 1 << 63 else v


def _packed_varints(v):
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(_signed(x))
    return out


_DTYPES = {1: np.float32, 2: np.uint8, 3: np.int8, 5: np.int16, 6: np.int32, 7: np.int64, 9: np.bool_, 10: np.float16, 11: np.float64, 12: np.uint32, 13: np.uint64}


def _tensor(buf):
    """TensorProto -> (name, ndarray)."""
    dims, dtype, name, raw, floats, int32s, int64s, doubles = [], 1, "", None, [], [], [], []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            dims += _packed_varints(v) if wt == _WT_LEN else [_signed(v)]
        elif fno == 2:
            dtype = v
        elif fno == 8:
            name = bytes(v).decode()
        elif fno == 9:
            raw = v
        elif fno == 4:
            floats.append(np.frombuffer(v, "<f4") if wt == _WT_LEN else np.frombuffer(v, "<f4"))
        elif fno == 5:
            int32s += _packed_varints(v) if wt == _WT_LEN else [_signed(v)]
        elif fno == 7:
            int64s += _packed_varints(v) if wt == _WT_LEN else [_signed(v)]
        elif fno == 10:
            doubles.append(np.frombuffer(v, "<f8"))
        elif fno == 14 and v == 1:
            raise ValueError(f"tensor {name!r} uses external data: not supported")
    if dtype == 16:                                          # bfloat16: widen to float32
        a = (np.frombuffer(raw, "<u2").astype(np.uint32) << 16).view(np.float32)
    elif raw is not None:
        a = np.frombuffer(raw, np.dtype(_DTYPES[dtype]).newbyteorder("<"))
    elif floats:
        a = np.concatenate(floats)
    elif doubles:
        a = np.concatenate(doubles)
    elif int64s:
        a = np.asarray(int64s, np.int64)
    elif dtype == 10:                                        # float16 in int32_data: the bit patterns
        a = np.asarray(int32s, np.uint16).view(np.float16)
    else:
        a = np.asarray(int32s, _DTYPES.get(dtype, np.int32))
    return name, np.array(a).reshape(dims) if dims else np.array(a).reshape(())


@dataclass
class OnnxNode:
    op: str
    inputs: list
    outputs: list
    name: str = ""
    attrs: dict = field(default_factory=dict)


@dataclass
class OnnxGraph:
    nodes: list
    tensors: dict                        # initializers + Constant node values, by value name
    inputs: list
    outputs: list


def _attribute(buf):
    name, val = "", None
    ints, floats = [], []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            name = bytes(v).decode()
        elif fno == 2:
            val = struct.unpack("<f", v)[0]
        elif fno == 3:
            val = _signed(v)
        elif fno == 4:
            val = bytes(v)
        elif fno == 5:
            val = _tensor(v)[1]
        elif fno == 8:
            ints += _packed_varints(v) if wt == _WT_LEN else [_signed(v)]
        elif fno == 7:
            floats += list(np.frombuffer(v, "<f4")) if wt == _WT_LEN else [struct.unpack("<f", v)[0]]
    if ints:
        val = ints
    elif floats and val is None:
        val = floats
    return name, val


def _node(buf):
    n = OnnxNode("", [], [])
    for fno, wt, v in _fields(buf):
        if fno == 1:
            n.inputs.append(bytes(v).decode())
        elif fno == 2:
            n.outputs.append(bytes(v).decode())
        elif fno == 3:
            n.name = bytes(v).decode()
        elif fno == 4:
            n.op = bytes(v).decode()
        elif fno == 5:
            k, val = _attribute(v)
            n.attrs[k] = val
    return n


def _value_name(buf):
    for fno, wt, v in _fields(buf):
        if fno == 1:
            return bytes(v).decode()
    return ""


def read_onnx(src) -> OnnxGraph:
    """src: path, bytes or a file object with an ONNX ModelProto."""
    if isinstance(src, (bytes, bytearray, memoryview)):
        data = src
    elif hasattr(src, "read"):
        data = src.read()
    else:
        with open(src, "rb") as f:
            data = f.read()
    graph = None
    for fno, wt, v in _fields(memoryview(data)):
        if fno == 7:
            graph = v
    if graph is None:
        raise ValueError("no GraphProto in the file: not an ONNX model")
    g = OnnxGraph([], {}, [], [])
    for fno, wt, v in _fields(graph):
        if fno == 1:
            g.nodes.append(_node(v))
        elif fno == 5:
            name, a = _tensor(v)
            g.tensors[name] = a
        elif fno == 11:
            g.inputs.append(_value_name(v))
        elif fno == 12:
            g.outputs.append(_value_name(v))
    for n in g.nodes:
        if n.op == "Constant" and "value" in n.attrs and n.outputs:
            g.tensors[n.outputs[0]] = n.attrs["value"]
    g.inputs = [i for i in g.inputs if i not in g.tensors]           # old exporters list the initializers as inputs too
    return g


# ------------------------------------------------------------------------------------------------ structural matching
_WEIGHTED = ("Conv", "Gemm", "MatMul", "ConvTranspose")
_SHAPE_ONLY = ("Shape", "ConstantOfShape", "Range", "Constant")


def _stored_source(g: OnnxGraph, v, producer):
    """name of the stored tensor that value v is a reshape of (Reshape / Unsqueeze / Identity / Cast chain), or None"""
    for _ in range(6):
        if v in g.tensors:
            return v if g.tensors[v].ndim >= 1 and g.tensors[v].size > 1 or g.tensors[v].ndim == 1 else None
        i = producer.get(v)
        if i is None or g.nodes[i].op not in ("Reshape", "Unsqueeze", "Identity", "Cast", "Squeeze", "Expand"):
            return None
        v = g.nodes[i].inputs[0]
    return None


def _graph_maps(g: OnnxGraph):
    if not hasattr(g, "_maps"):
        cons, prod = {}, {}
        for i, n in enumerate(g.nodes):
            for o in n.outputs:
                prod[o] = i
            for k, x in enumerate(n.inputs):
                cons.setdefault(x, []).append((i, k))
        g._maps = (cons, prod)
    return g._maps


def _weighted_param_inputs(g: OnnxGraph, n: OnnxNode):
    """(weight name, bias name or None, weight stored transposed?) of a weighted node whose weight is a stored tensor; None for activation x
    activation MatMuls. A Linear exported without constant folding is Transpose(stored (out, in)) -> MatMul: the stored tensor is found
    through the Transpose and flagged. The bias is the node's own third input (an exported BatchNorm-folded convolution) or -- modules that keep
    the bias outside the convolution, like backbones.common.ConvBiasAct, and MatMul + Add linears -- the stored per-channel tensor added to the
    node's output."""
    consumers, producer = _graph_maps(g)
    transposed = False
    if n.op == "MatMul":
        w = None
        for k in (1, 0):
            if k >= len(n.inputs):
                continue
            v = n.inputs[k]
            if v in g.tensors and g.tensors[v].ndim >= 2:
                w = v
                break
            i = producer.get(v)
            if i is not None and g.nodes[i].op == "Transpose" and g.nodes[i].inputs[0] in g.tensors and g.tensors[g.nodes[i].inputs[0]].ndim == 2:
                w, transposed = g.nodes[i].inputs[0], True
                break
        if w is None:
            return None
        bias = None
        cout = g.tensors[w].shape[0] if transposed else g.tensors[w].shape[-1]
    elif len(n.inputs) > 1 and n.inputs[1] in g.tensors:
        w, bias = n.inputs[1], (n.inputs[2] if len(n.inputs) > 2 and n.inputs[2] in g.tensors else None)
        cout = g.tensors[w].shape[0]
    else:
        return None
    if bias is None:
        for ci, slot in consumers.get(n.outputs[0], []):
            c = g.nodes[ci]
            if c.op == "Add" and len(c.inputs) == 2:
                src = _stored_source(g, c.inputs[1 - slot], producer)
                if src is not None and g.tensors[src].size == cout:
                    bias = src
                    break
    return w, bias, transposed


_TRANSPARENT = ("Mul", "Sigmoid", "Relu", "Clip", "HardSigmoid", "HardSwish", "LeakyRelu", "Identity", "Add", "Div", "Sub", "Pow", "Sqrt", "Erf", "Tanh",
                "Softmax", "Reshape", "Transpose", "Flatten", "Expand", "Tile", "Cast", "Unsqueeze", "Squeeze", "Gather")


class _Interner:
    """Hash-consing of nested signatures: every distinct tuple gets a small integer, children are referred to by their integers (a signature
    written out in full grows exponentially with the depth of the network)."""

    def __init__(self):
        self.ids = {}

    def __call__(self, *key):
        return self.ids.setdefault(key, len(self.ids))


def _signatures(g: OnnxGraph, intern: _Interner):
    """{node index: label} for the weighted nodes of g, equal labels <=> same structural position.
    Upstream part: the label of a VALUE is, for the output of a weighted node, (its attributes, the label of its data input); for an
    unweighted operator with several distinct data inputs (Concat: ordered; others: sorted) the tuple of their labels; unary operators --
    and n-ary ones whose inputs collapse to one producer, e.g. SiLU = Mul(x, Sigmoid(x)) -- are transparent.
    Downstream part (two convolutions that read the same tensor with the same shape, like the two branches of a CSP layer, have equal upstream
    labels): a few rounds of refinement with the multiset of (path through the unweighted operators incl. the Concat slot, label of the first
    weighted consumer) -- what the node's output is USED for."""
    producer, consumers = {}, {}
    for i, n in enumerate(g.nodes):
        for o in n.outputs:
            producer[o] = i
        for k, x in enumerate(n.inputs):
            consumers.setdefault(x, []).append((i, k))
    memo, up, elem = {}, {}, {}

    def own(n, wp):
        shape = tuple(int(d) for d in g.tensors[wp[0]].shape)
        if wp[2]:
            shape = shape[::-1]                               # a Linear weight stored (out, in) behind a Transpose: compare as (in, out)
        return (n.op, shape, tuple(n.attrs.get("strides", []) or []), int(n.attrs.get("group", 1) or 1), tuple(n.attrs.get("dilations", []) or []))

    def weighted(n):
        return _weighted_param_inputs(g, n) if n.op in _WEIGHTED else None

    def value_sig(v):
        if v in memo:
            return memo[v]
        memo[v] = intern("cycle")
        if v in g.tensors:
            s = None                                          # a stored tensor is not a data path
        elif v not in producer:
            s = intern("in", g.inputs.index(v) if v in g.inputs else -1)
        else:
            i = producer[v]
            n = g.nodes[i]
            wp = weighted(n)
            if wp is not None:
                data = [x for x in n.inputs if x not in g.tensors]
                data = [x for x in data if producer.get(x) is None or g.nodes[producer[x]].op != "Transpose" or g.nodes[producer[x]].inputs[0] not in g.tensors]
                s = intern("W", own(n, wp), value_sig(data[0]) if data else None)
                up[i] = s
            elif n.op in _SHAPE_ONLY:
                s = None                                      # shape arithmetic feeding Reshape / Resize / Slice: not a data path
            else:
                ins = [value_sig(x) for x in n.inputs if x]
                ins = [x for x in ins if x is not None]
                if not ins:
                    s = None                                  # computed from shapes / constants only
                elif n.op == "Concat":
                    s = intern("cat", tuple(ins))
                elif n.op == "Slice":                       # Focus: the four phase slices differ by their constant starts
                    # canonical form (axes, starts, steps): the END of a slice is written differently by different exporters (INT64_MAX, the
                    # concrete extent, a Shape-derived value) and does not say which phase it is; opset < 10 carries starts / axes as attributes
                    def consts(k, attr):
                        v_ = n.inputs[k] if k < len(n.inputs) else None
                        for _ in range(6):                   # exporters without constant folding wrap the constants: Unsqueeze(Constant, axes)
                            if v_ is None or v_ in g.tensors:
                                break
                            pi = producer.get(v_)
                            v_ = g.nodes[pi].inputs[0] if pi is not None and g.nodes[pi].op in ("Unsqueeze", "Squeeze", "Identity", "Cast", "Reshape") else None
                        if v_ is not None and v_ in g.tensors:
                            return tuple(int(t) for t in np.asarray(g.tensors[v_]).reshape(-1)[:4])
                        return tuple(int(t) for t in (n.attrs.get(attr) or []))[:4]
                    s = intern("slice", consts(3, "axes"), consts(1, "starts"), consts(4, "steps"), tuple(ins[:1]))
                else:
                    uniq = tuple(sorted(set(ins)))
                    if len(uniq) == 1:
                        s = uniq[0] if n.op in _TRANSPARENT else intern(n.op, tuple(n.attrs.get("kernel_shape", []) or []), uniq[0])
                        if n.op in ("Mul", "Add", "Sub", "Div") and len(n.inputs) == 2:
                            # an elementwise PARAMETER (norm gain, per-head scale / offset): stored tensor (x) data path
                            for k in (0, 1):
                                src = _stored_source(g, n.inputs[k], producer)
                                if src is not None and value_sig(n.inputs[1 - k]) is not None:
                                    shape = tuple(int(d) for d in g.tensors[src].shape if d != 1)
                                    elem[i] = (intern("E", n.op, shape, uniq[0]), src)
                    else:
                        s = intern(n.op, uniq)
        memo[v] = s
        return s

    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 20000))
    for n in g.nodes:
        for o in n.outputs:
            value_sig(o)

    def first_weighted_consumers(i):
        """[(path tokens, consumer node index | -1 for a graph output)] reached from node i's output through unweighted operators."""
        out, seen = [], set()
        stack = [(o, ()) for o in g.nodes[i].outputs]
        while stack:
            v, path = stack.pop()
            if v in g.outputs:
                out.append((path, -1))
            for ci, slot in consumers.get(v, []):
                c = g.nodes[ci]
                if weighted(c) is not None:
                    out.append((path, ci))
                    continue
                if c.op in _SHAPE_ONLY:
                    continue
                # path tokens: Concat slots and the operators that change what flows (pooling, resizing, slicing).  The element-wise
                # operators are NOT part of the path: how an exporter spells bias + activation differs between files of one architecture --
                # Conv with the BatchNorm-folded bias inside, then Sigmoid + Mul (mmdeploy / rtmlib) against Conv, Add(bias), Sigmoid, Mul
                # (this repo's modules keep the bias outside the convolution) -- r06, found by the hand-written rtmlib-style file of
                # tests/test_weights.py
                tok = (("cat", slot),) if c.op == "Concat" else () if c.op in _TRANSPARENT else ((c.op,),)
                key = (ci, slot)
                if key in seen or len(path) > 24:
                    continue
                seen.add(key)
                for o in c.outputs:
                    stack.append((o, path + tok))
        return out

    down = {i: first_weighted_consumers(i) for i in up}
    label = dict(up)
    for _ in range(6):
        if len(set(label.values())) == len(label):
            break
        label = {i: intern("R", label[i], tuple(sorted((path, label.get(ci, -1)) for path, ci in down[i]))) for i in label}
    # elementwise parameters: upstream label + what their result feeds (final labels of the first weighted consumers)
    claimed = {wp[1] for wp in (weighted(g.nodes[i]) for i in up) if wp[1] is not None}
    elabel = {}
    for i, (lab, src) in elem.items():
        if src in claimed:
            continue
        elabel[i] = (intern("ER", lab, tuple(sorted((path, label.get(ci, -1)) for path, ci in first_weighted_consumers(i)))), src)
    return label, elabel


def match_convolutions(ref: OnnxGraph, own: OnnxGraph, own_params=None):
    """[(own weight name, own bias name | None, ref weight array, ref bias array | None)] for every weighted node of `own` -- raises when the two
    graphs are not the same architecture (a node without a partner, or an ambiguous label) -- followed by (name, None, array, None) for the
    elementwise parameters of `own` (norm gains, per-head scales; `own_params` = names that are parameters) that have exactly one structural
    partner in `ref`; those without one are returned in the second list, not guessed."""
    intern = _Interner()                                     # shared: equal structures get equal integers in both graphs
    (sr, er), (so, eo) = _signatures(ref, intern), _signatures(own, intern)
    by_sig = {}
    for i, s in sr.items():
        by_sig.setdefault(s, []).append(i)
    out = []
    for i, s in sorted(so.items()):
        cands = by_sig.get(s, [])
        n = own.nodes[i]
        if len(cands) != 1:
            raise ValueError(f"weight import: node {n.name or n.op} (weight {tuple(own.tensors[_weighted_param_inputs(own, n)[0]].shape)}) has "
                             f"{len(cands)} structural partners in the reference graph: not the same architecture")
        rn = ref.nodes[cands[0]]
        ow, ob, ot = _weighted_param_inputs(own, n)
        rw, rb, rt = _weighted_param_inputs(ref, rn)
        arr = ref.tensors[rw]
        out.append((ow, ob, arr.T if ot != rt else arr, ref.tensors[rb] if rb is not None else None))
    if len(out) != len(sr):
        raise ValueError(f"weight import: the reference graph has {len(sr)} weighted nodes, this module {len(out)}")
    by_e = {}
    for i, (lab, src) in er.items():
        by_e.setdefault(lab, []).append(src)
    unmatched = []
    for i, (lab, src) in sorted(eo.items()):
        if own_params is not None and src not in own_params:
            continue
        cands = sorted(set(by_e.get(lab, [])))
        if len(cands) == 1:
            out.append((src, None, ref.tensors[cands[0]], None))
        else:
            unmatched.append(src)
    return out, unmatched


# ------------------------------------------------------------------------------------------------ torch side
def export_onnx_bytes(module, example_inputs) -> bytes:
    """TorchScript ONNX export of `module` into memory without the `onnx` package (the exporter only needs it for a post-processing step that
    does not apply here). Initializers keep their state_dict names."""
    import torch
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    import warnings
    saved = onnx_proto_utils._add_onnxscript_fn
    onnx_proto_utils._add_onnxscript_fn = lambda proto, custom_opsets: proto
    try:
        f = io.BytesIO()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.onnx.export(module, tuple(example_inputs), f, opset_version=13, dynamo=False, do_constant_folding=False)
        return f.getvalue()
    finally:
        onnx_proto_utils._add_onnxscript_fn = saved


def import_onnx_weights(module, example_inputs, ref, strict: bool = True):
    """Copy the weights of the ONNX model `ref` (path / bytes / OnnxGraph) into `module` (same architecture, e.g. backbones.yolox.YOLOX for the
    rtmlib YOLOX file). `module` must be on the CPU in float32 for the export; returns the number of tensors written."""
    import torch
    ref_g = ref if isinstance(ref, OnnxGraph) else read_onnx(ref)
    sd = module.state_dict()
    # the exporter merges initializers with identical VALUES (all-zero biases of equal length become one tensor): export the structure with
    # every parameter filled with its own noise, then put the module's values back
    saved = {k: v.clone() for k, v in sd.items()}
    gen = torch.Generator().manual_seed(1234)
    with torch.no_grad():
        for v in sd.values():
            if v.is_floating_point():
                v.copy_(torch.randn(v.shape, generator=gen) * 0.05 + (1.0 if v.dim() <= 1 else 0.0))
    try:
        own_g = read_onnx(export_onnx_bytes(module, example_inputs))
    finally:
        with torch.no_grad():
            for k, v in sd.items():
                v.copy_(saved[k])
    pairs, unmatched = match_convolutions(ref_g, own_g, set(sd))
    if strict and unmatched:
        raise ValueError(f"weight import: elementwise parameters without a structural partner in the reference graph: {unmatched}")
    written = 0
    with torch.no_grad():
        for ow, ob, rw, rb in pairs:
            for key, arr in ((ow, rw), (ob, rb)):
                if key is None:
                    continue
                if key not in sd:
                    raise KeyError(f"exported initializer {key!r} is not a state_dict entry (constant folding?)")
                if arr is None:
                    if strict:
                        raise ValueError(f"{key}: the reference node has no bias")
                    continue
                t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))
                if t.numel() != sd[key].numel():
                    raise ValueError(f"{key}: reference tensor {tuple(t.shape)} does not fit {tuple(sd[key].shape)}")
                sd[key].copy_(t.reshape(sd[key].shape))
                written += 1
    import_onnx_weights.unmatched = unmatched
    return written


def fold_batchnorm(conv_w, gamma, beta, mean, var, eps=1e-5, conv_b=None):
    """Conv + BatchNorm(eval) -> (weight, bias) of the equivalent convolution, in float64 then float32."""
    w = np.asarray(conv_w, np.float64)
    s = np.asarray(gamma, np.float64) / np.sqrt(np.asarray(var, np.float64) + eps)
    b = np.asarray(beta, np.float64) - np.asarray(mean, np.float64) * s
    if conv_b is not None:
        b = b + np.asarray(conv_b, np.float64) * s
    return (w * s.reshape(-1, *([1] * (w.ndim - 1)))).astype(np.float32), b.astype(np.float32)


def fold_batchnorm_state_dict(sd, pairs, eps=1e-5):
    """sd: {name: array-like}; pairs: [(conv prefix, bn prefix, target conv key, target bias key)] -> {target key: float32 array}."""
    out = {}
    g = lambda k: np.asarray(sd[k].detach().cpu().numpy() if hasattr(sd[k], "detach") else sd[k])      # noqa: E731
    for conv, bn, tw, tb in pairs:
        w, b = fold_batchnorm(g(conv + ".weight"), g(bn + ".weight"), g(bn + ".bias"), g(bn + ".running_mean"), g(bn + ".running_var"), eps,
                              g(conv + ".bias") if conv + ".bias" in sd else None)
        out[tw], out[tb] = w, b
    return out


def resnet50_bn_pairs(prefix=""):
    """(conv, bn, target weight, target bias) of a ResNet-50 in torchvision / torchreid naming -> backbones.reid._ResNet50 naming."""
    pairs = [(prefix + "conv1", prefix + "bn1", "conv1.conv.weight", "conv1.bias")]
    for li, nblocks in enumerate((3, 4, 6, 3), start=1):
        for b in range(nblocks):
            src, dst = f"{prefix}layer{li}.{b}", f"layer{li}.{b}"
            for c in (1, 2, 3):
                pairs.append((f"{src}.conv{c}", f"{src}.bn{c}", f"{dst}.c{c}.conv.weight", f"{dst}.c{c}.bias"))
            if b == 0:
                pairs.append((f"{src}.downsample.0", f"{src}.downsample.1", f"{dst}.down.conv.weight", f"{dst}.down.bias"))
    return pairs


# ------------------------------------------------------------------------------------------------ one entry point for the plugin modules
def load_checkpoint(module, path, example_inputs=None, backbone_attr="backbone", strict_heads=False):
    """What `cfg.checkpoint` / `cfg.model_weights` of the Hip* modules accepts:

    * ``*.onnx`` -- the artefact the reference itself loads for its detector / pose estimator (rtmlib model zoo,
      configs/modules/bbox_detector/yolox_rtmlib.yaml:1-7): imported structurally (`import_onnx_weights`) through a CPU fp32 twin of `module`;
      `example_inputs` = a tuple with one input of the network's shape.
    * a torch checkpoint whose keys are `module`'s own -> `load_state_dict`.
    * a torch checkpoint of a ResNet-50 WITH BatchNorm layers in torchvision / torchreid naming (optionally under a prefix, optionally wrapped in
      {"state_dict": ...}; tracklab/wrappers/reid/kpreid_api.py:133-161 loads such files through torchreid): the backbone is folded
      (`fold_batchnorm_state_dict`) into ``getattr(module, backbone_attr)``; keys outside the backbone that do not exist here are returned,
      the module's own tensors outside the backbone (the part-based head) are reported as ``uninitialised_parameters`` with a WARNING
      (``strict_heads=True``: an error instead).

    A path that does not exist raises FileNotFoundError -- never a silent fall-back to random weights. Returns a dict with what was done."""
    import copy
    import os
    import torch
    path = str(path)
    if not os.path.exists(path):
        raise FileNotFoundError(f"checkpoint {path!r} does not exist; use null for random-init weights (throughput only)")
    if path.lower().endswith(".onnx"):
        if example_inputs is None:
            raise ValueError("load_checkpoint: an ONNX file needs example_inputs (one input tensor of the network's shape)")
        twin = copy.deepcopy(module).to(device="cpu", dtype=torch.float32).to(memory_format=torch.contiguous_format)
        n = import_onnx_weights(twin, tuple(t.detach().to("cpu", torch.float32) for t in example_inputs), path)
        module.load_state_dict(twin.state_dict())
        return {"format": "onnx", "tensors": n}
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    own = module.state_dict()
    if set(sd) == set(own):
        module.load_state_dict(sd)
        return {"format": "state_dict", "tensors": len(sd)}
    bn_key = next((k for k in sd if k.endswith("layer1.0.bn1.running_mean")), None)
    if bn_key is not None and hasattr(module, backbone_attr):
        prefix = bn_key[:-len("layer1.0.bn1.running_mean")]
        folded = fold_batchnorm_state_dict(sd, resnet50_bn_pairs(prefix))
        bb = getattr(module, backbone_attr)
        bb.load_state_dict({k: torch.from_numpy(v) for k, v in folded.items()})
        used = {p[0] + s for p in resnet50_bn_pairs(prefix) for s in (".weight", ".bias")} | \
               {p[1] + s for p in resnet50_bn_pairs(prefix) for s in (".weight", ".bias", ".running_mean", ".running_var", ".num_batches_tracked")}
        left = sorted(k for k in sd if k not in used)
        # Only the backbone can be mapped: the part-based head of the reference lives in the third-party torchreid fork (not vendored, its key
        # names are not in the reference tree), so `module`'s own head stays as it was initialised.  That must never pass silently
        # (ADVICE r03): the report lists it, a WARNING is logged and warned, and strict_heads=True turns it into an error.
        uninit = sorted(k for k in own if not k.startswith(backbone_attr + "."))
        report = {"format": "resnet50+batchnorm", "tensors": len(folded), "unmapped_keys": left, "uninitialised_parameters": uninit}
        if uninit:
            msg = (f"checkpoint {path!r}: the ResNet-50 backbone was loaded ({len(folded)} tensors, BatchNorm folded) but {len(uninit)} tensors of "
                   f"{type(module).__name__} outside `{backbone_attr}` keep their RANDOM initialisation: {uninit[:6]}{' ...' if len(uninit) > 6 else ''}; "
                   f"{len(left)} checkpoint tensors were not used: {left[:6]}{' ...' if len(left) > 6 else ''}. Embeddings of this module are NOT the "
                   "reference model's until the head is loaded too (save this module's own state_dict, or an ONNX export).")
            if strict_heads:
                raise ValueError(msg)
            import logging
            import warnings
            logging.getLogger("tracklab_amd.weights").warning(msg)
            warnings.warn(msg, RuntimeWarning, stacklevel=2)
        return report
    missing = sorted(set(own) - set(sd))[:5]
    raise ValueError(f"checkpoint {path!r}: neither this module's state_dict (missing e.g. {missing}) nor a ResNet-50 with BatchNorm layers")
