"""TensorFlow-free reader of frozen TF1 graphs (``*.pb`` = serialized ``GraphDef``), SURVEY.md row f1.

The reference loads its nets with ``tf.GraphDef().ParseFromString`` + ``tf.import_graph_def``
(``net_post_processing_helper.py:36-53``, ``gnn/io.py:12-25``).  The frozen graphs were produced by
``convert_variables_to_constants`` (``gnn/model/model_base.py:473-476``), which turns every variable into a
``Const`` node *with the variable's name*.  This module decodes the protobuf wire format directly, extracts those
constants and maps them onto the engine's named weight set (``weights.py``) by the reference's variable-scope names
(any graph prefix such as ``graph/`` is ignored), and derives the model hyper-parameters from tensor shapes and the
op list.

Two ways onto the engine's weight set:

  * by NAME (``aru_from_constants`` / ``gnn_from_nodes``): graphs frozen by this repository's own exporter keep the
    variable scopes of ``ARU_v1.py`` / ``graph_relation.py`` (``aru_net/featMapG/unet_down_0/conv1/weights`` ...);
  * by TOPOLOGY (``aru_from_topology``): the shipped separator / heading nets were exported by another project
    (``region_net_post_processor_base.py:278-290``), only ``inImg`` / ``output`` are known
    (``net_post_processing_helper.py:69-70``).  The op graph between the two is walked: every Conv2D /
    Conv2DBackpropInput with a constant filter is a layer, its bias (BiasAdd / Add with a constant), an optional
    inference-mode batch normalisation (FusedBatchNorm*, or the Mul / Add pair it is usually folded into) and its
    activation are read off its consumers, layers are ordered by their conv depth below ``inImg`` (the ARU-Net is one
    chain with skip connections, so the depth order IS the creation order), the hyper-parameters follow from the
    filter shapes and the weights are assigned by position.  Batch normalisation is folded into weights and bias.
    Anything that does not fit the ARU_v1 family raises ``IOError`` with the reason -- nothing is guessed.

Message fields decoded (field numbers from tensorflow/core/framework/*.proto); everything else is skipped:
    GraphDef.node = 1 ; NodeDef{name=1, op=2, input=3, attr=5(map<string,AttrValue>)}
    AttrValue{list=1{s=2,i=3,f=4,b=5,type=6,shape=7}, s=2, i=3, f=4, b=5, type=6, shape=7, tensor=8}
    TensorProto{dtype=1, tensor_shape=2, tensor_content=4, float_val=5, double_val=6, int_val=7, int64_val=10}
    TensorShapeProto{dim=2{size=1}}
The reader is cross-checked against GraphDefs serialised by ``google.protobuf`` (tests/test_pb_import_protobuf.py).
"""
import struct
from collections import OrderedDict

import numpy as np

from .config import AruConfig, GnnConfig
from .weights import aru_tensor_shapes, gnn_tensor_shapes

DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64 = 1, 2, 3, 9


# ----------------------------------------------------------------------------------------------
# protobuf wire format
# ----------------------------------------------------------------------------------------------
def _varint(buf, pos):
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise IOError("malformed varint")


def _fields(buf):
    """Yields (field_number, wire_type, value) for one message; value is int or memoryview."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = buf[pos:pos + 4]
            pos += 4
        else:
            raise IOError(f"unsupported protobuf wire type {wt}")
        if pos > n:
            raise IOError("truncated protobuf message")
        yield fno, wt, val


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _parse_shape(buf):
    dims = []
    for fno, wt, val in _fields(buf):
        if fno == 2 and wt == 2:
            size = 0
            for f2, w2, v2 in _fields(val):
                if f2 == 1 and w2 == 0:
                    size = _signed(v2)
            dims.append(size)
    return dims


def _parse_tensor(buf):
    dtype, shape, content = 0, [], None
    floats, doubles, ints, int64s = [], [], [], []
    for fno, wt, val in _fields(buf):
        if fno == 1 and wt == 0:
            dtype = val
        elif fno == 2 and wt == 2:
            shape = _parse_shape(val)
        elif fno == 4 and wt == 2:
            content = bytes(val)
        elif fno == 5:
            if wt == 2:
                floats.extend(struct.unpack(f"<{len(val) // 4}f", bytes(val)))
            else:
                floats.append(struct.unpack("<f", bytes(val))[0])
        elif fno == 6:
            if wt == 2:
                doubles.extend(struct.unpack(f"<{len(val) // 8}d", bytes(val)))
            else:
                doubles.append(struct.unpack("<d", bytes(val))[0])
        elif fno == 7:
            if wt == 2:
                p = 0
                while p < len(val):
                    v, p = _varint(val, p)
                    ints.append(_signed(v))
            else:
                ints.append(_signed(val))
        elif fno == 10:
            if wt == 2:
                p = 0
                while p < len(val):
                    v, p = _varint(val, p)
                    int64s.append(_signed(v))
            else:
                int64s.append(_signed(val))
    np_dtype = {DT_FLOAT: "<f4", DT_DOUBLE: "<f8", DT_INT32: "<i4", DT_INT64: "<i8"}.get(dtype)
    if np_dtype is None:
        return None
    count = int(np.prod(shape)) if shape else 1
    if content is not None and len(content):
        arr = np.frombuffer(content, dtype=np_dtype).copy()
    else:
        vals = {DT_FLOAT: floats, DT_DOUBLE: doubles, DT_INT32: ints, DT_INT64: int64s}[dtype]
        arr = np.asarray(vals, dtype=np_dtype)
        if arr.size == 1 and count > 1:                      # TF stores a splat value once
            arr = np.full(count, arr[0], dtype=np_dtype)
        elif arr.size == 0 and count:
            arr = np.zeros(count, dtype=np_dtype)
    if arr.size != count:
        raise IOError(f"tensor has {arr.size} elements, shape {shape} needs {count}")
    return arr.reshape(shape)


def _scalars(val, wt, fmt, size):
    """repeated scalar field: one element (wire type 0 / 1 / 5) or a packed run (wire type 2)"""
    if wt == 2:
        if fmt is None:
            out, p = [], 0
            while p < len(val):
                v, p = _varint(val, p)
                out.append(_signed(v))
            return out
        return list(struct.unpack(f"<{len(val) // size}{fmt}", bytes(val)))
    if fmt is None:
        return [_signed(val)]
    return [struct.unpack("<" + fmt, bytes(val))[0]]


def _parse_attr(buf, want_tensor):
    """AttrValue -> python value: bytes -> str, int, float, bool, ('type', enum), ('shape', dims), list of those,
    ndarray for a tensor (only decoded when ``want_tensor``)"""
    out = None
    for fno, wt, val in _fields(buf):
        if fno == 1 and wt == 2:                                   # ListValue
            items = []
            for f2, w2, v2 in _fields(val):
                if f2 == 2 and w2 == 2:
                    items.append(bytes(v2).decode("utf-8", "replace"))
                elif f2 == 3:
                    items.extend(_scalars(v2, w2, None, 0))
                elif f2 == 4:
                    items.extend(_scalars(v2, w2, "f", 4))
                elif f2 == 5:
                    items.extend(bool(v) for v in _scalars(v2, w2, None, 0))
                elif f2 == 6:
                    items.extend(("type", v) for v in _scalars(v2, w2, None, 0))
                elif f2 == 7 and w2 == 2:
                    items.append(("shape", _parse_shape(v2)))
            out = items
        elif fno == 2 and wt == 2:
            out = bytes(val).decode("utf-8", "replace")
        elif fno == 3 and wt == 0:
            out = _signed(val)
        elif fno == 4 and wt == 5:
            out = struct.unpack("<f", bytes(val))[0]
        elif fno == 5 and wt == 0:
            out = bool(val)
        elif fno == 6 and wt == 0:
            out = ("type", val)
        elif fno == 7 and wt == 2:
            out = ("shape", _parse_shape(val))
        elif fno == 8 and wt == 2 and want_tensor:
            out = _parse_tensor(val)
    return out


def parse_graphdef(data: bytes):
    """-> list of nodes: {'name', 'op', 'input': [...], 'attr': {key: value}, 'value': ndarray or None (Const only)}"""
    nodes = []
    for fno, wt, val in _fields(memoryview(data)):
        if fno != 1 or wt != 2:
            continue
        node = {"name": "", "op": "", "input": [], "attr": {}, "value": None}
        raw_attrs = []
        for f2, w2, v2 in _fields(val):
            if f2 == 1 and w2 == 2:
                node["name"] = bytes(v2).decode("utf-8", "replace")
            elif f2 == 2 and w2 == 2:
                node["op"] = bytes(v2).decode("utf-8", "replace")
            elif f2 == 3 and w2 == 2:
                node["input"].append(bytes(v2).decode("utf-8", "replace"))
            elif f2 == 5 and w2 == 2:                        # map entry {key=1, value=2}
                key, attr = None, None
                for f3, w3, v3 in _fields(v2):
                    if f3 == 1 and w3 == 2:
                        key = bytes(v3).decode("utf-8", "replace")
                    elif f3 == 2 and w3 == 2:
                        attr = v3
                if key is not None and attr is not None:
                    raw_attrs.append((key, attr))
        for key, attr in raw_attrs:                          # the op may come after the attrs on the wire
            is_value = key == "value" and node["op"] == "Const"
            parsed = _parse_attr(attr, want_tensor=is_value)
            if is_value:
                node["value"] = parsed
            else:
                node["attr"][key] = parsed
        nodes.append(node)
    if not nodes:
        raise IOError("no NodeDef found: not a GraphDef")
    return nodes


def read_graph(path):
    with open(path, "rb") as f:
        return parse_graphdef(f.read())


def const_tensors(nodes):
    return OrderedDict((n["name"], n["value"]) for n in nodes if n["op"] == "Const" and n["value"] is not None)


# ----------------------------------------------------------------------------------------------
# mapping onto the engine's weight sets
# ----------------------------------------------------------------------------------------------
def _find(consts, suffix):
    """Constant whose name equals `suffix` up to a graph prefix ('graph/aru_net/...' matches 'aru_net/...')."""
    hits = [k for k in consts if k == suffix or k.endswith("/" + suffix)]
    if len(hits) > 1:
        hits.sort(key=len)
    return consts[hits[0]] if hits else None


def _hint_num_scales(nodes, use_att):
    """constants-only containers carry one AvgPool node per extra pyramid scale (ARU_v1.py:106-109) as a hint"""
    if not use_att:
        return 1
    n = sum(1 for n in nodes if n["op"] == "AvgPool")
    return 1 + n if n else None


def _hint_softmax(nodes):
    by_name = {n["name"]: n for n in nodes}
    node = by_name.get("output")
    hops = 0
    while node is not None and node["op"] == "Identity" and node["input"] and hops < 64:
        node = by_name.get(node["input"][0].lstrip("^").split(":")[0])
        hops += 1
    if node is None:
        return None
    return node["op"] == "Softmax"


def aru_from_constants(nodes, num_scales_att=None, apply_softmax=None):
    """Mapping by variable NAME (graphs frozen with the scopes of ARU_v1.py; also the engine's own export, which holds
    constants only).  What the constants cannot tell -- the number of pyramid scales and whether ``output`` is behind a
    class softmax -- must be passed in or be readable from the op list; it is never defaulted."""
    consts = const_tensors(nodes)
    if _find(consts, "aru_net/featMapG/unet_down_0/conv1/weights") is None:
        raise IOError("no ARU-Net variables (aru_net/featMapG/...) among the graph constants; "
                      "constants found: " + ", ".join(list(consts)[:8]) + " ...")
    levels = 0
    while _find(consts, f"aru_net/featMapG/unet_down_{levels}/conv1/weights") is not None:
        levels += 1
    res_depth = 0
    while _find(consts, f"aru_net/featMapG/unet_down_0/convR_{res_depth}/weights") is not None:
        res_depth += 1
    plain_u = res_depth == 0 and _find(consts, "aru_net/featMapG/unet_down_0/conv2/weights") is not None   # graph 'U'
    act = _find(consts, "asep_meta/activation")             # written by weights_to_graphdef(meta={"activation": code}); absent: relu,
    try:                                                    # ARU_v1.py:43's default (the constants cannot tell)
        activation_name = ("relu", "elu", "leaky")[int(np.asarray(act).reshape(-1)[0])] if act is not None else "relu"
    except IndexError:
        raise IOError("asep_meta/activation must be 0 (relu), 1 (elu) or 2 (leaky)")
    w0 = _find(consts, "aru_net/featMapG/unet_down_0/conv1/weights")
    wl = _find(consts, "aru_net/logit/class/weights")
    if wl is None:
        raise IOError("aru_net/logit/class/weights missing")
    use_att = _find(consts, "aru_net/attMapG/attPart/conv1/weights") is not None
    if num_scales_att is None:
        num_scales_att = _hint_num_scales(nodes, use_att)
        if num_scales_att is None:
            raise IOError("the graph has attention variables but no AvgPool op tells the number of pyramid scales: "
                          "pass num_scales_att explicitly")
    if apply_softmax is None:
        apply_softmax = _hint_softmax(nodes)
        if apply_softmax is None:
            raise IOError("the graph has no 'output' node: pass apply_softmax explicitly")
    cfg = AruConfig(graph="ARU" if use_att else ("U" if plain_u else "RU"), channels=int(w0.shape[2]), n_classes=int(wl.shape[3]),
                    feat_root=int(w0.shape[3]), scale_space_num=levels, res_depth=3 if plain_u else res_depth, activation_name=activation_name,
                    num_scales_att=int(num_scales_att), filter_size=int(w0.shape[0]),
                    mvn=any("aru_net/mvn" in n["name"] for n in nodes), apply_softmax=bool(apply_softmax))
    tensors = OrderedDict()
    for name, shape in aru_tensor_shapes(cfg).items():
        t = _find(consts, name)
        if t is None:
            raise IOError(f"frozen graph lacks the constant {name}")
        if tuple(t.shape) != tuple(shape):
            raise IOError(f"{name}: shape {tuple(t.shape)} in the graph, {tuple(shape)} expected")
        tensors[name] = np.ascontiguousarray(t, dtype=np.float32)
    return tensors, cfg


# ----------------------------------------------------------------------------------------------
# mapping by topology (graphs whose variable scopes are unknown)
# ----------------------------------------------------------------------------------------------
_PASS_OPS = ("Identity", "StopGradient", "PlaceholderWithDefault", "CheckNumerics", "Snapshot", "PreventGradient")
_SHAPE_OPS = ("Shape", "ShapeN", "Size", "Rank")
_BN_OPS = ("FusedBatchNorm", "FusedBatchNormV2", "FusedBatchNormV3")
_ADD_OPS = ("Add", "AddV2", "BiasAdd")
# every op type that may sit on the data path of an ARU_v1 graph (ARU_v1.py:62-294, layers.py); anything else is refused
_KNOWN_OPS = set(_PASS_OPS + _SHAPE_OPS + _BN_OPS + _ADD_OPS + (
    "Placeholder", "Const", "Fill", "Conv2D", "Conv2DBackpropInput", "Relu", "MaxPool", "AvgPool", "ConcatV2", "Softmax",
    "Split", "SplitV", "Mul", "AddN", "StridedSlice", "Pack"))
# the activations of the ARU_v1 variants (ARU_v1.py:70-75): tf.nn.elu, and layers.leaky_relu's composite
# maximum(0, x) + leak * minimum(0, x) (layers.py:10-30; tf.nn.leaky_relu's fused op is accepted too)
_ACT_VARIANT_OPS = {"elu": {"Elu"}, "leaky": {"Maximum", "Minimum", "LeakyRelu"}}
_MVN_OPS = {"Mean", "Square", "Sqrt", "Rsqrt", "Sub", "RealDiv", "Maximum", "Enter", "Exit", "Merge", "Switch",
            "NextIteration", "LoopCond", "Less", "LogicalAnd", "Range", "TensorArrayV3", "TensorArrayReadV3",
            "TensorArrayWriteV3", "TensorArrayScatterV3", "TensorArrayGatherV3", "TensorArraySizeV3",
            "TensorArrayUnstack", "Cast", "Reshape", "Squeeze", "ExpandDims"}
_UNSUPPORTED_ACT = {"Selu": "selu", "Relu6": "relu6", "Tanh": "tanh", "Sigmoid": "sigmoid"}
LEAK = 0.1                                                  # layers.py:10: the engine's act_kernel implements this slope only


def _activation_of(graph, path, ops_on_path, mvn):
    """'relu', 'elu' or 'leaky' from the activation ops between inImg and output; mixtures other than ARU_v1's own (the ReLU
    behind conv1 of every residual block, ARU_v1.py:214) and leaks other than 0.1 are refused"""
    has_elu = "Elu" in ops_on_path
    has_leaky = "Minimum" in ops_on_path or "LeakyRelu" in ops_on_path
    if has_elu and has_leaky:
        raise IOError("Elu and leaky-ReLU ops on the same path: not an ARU_v1 graph (ARU_v1.py:70-75 picks one activation)")
    if has_elu:
        return "elu"
    if not has_leaky:
        if "Maximum" in ops_on_path and not mvn:
            raise IOError("Maximum ops on the path without the Minimum half of layers.leaky_relu (layers.py:10-30)")
        return "relu"
    for n in path:
        if n["op"] == "LeakyRelu":
            alpha = n["attr"].get("alpha")
            alpha = 0.2 if alpha is None else float(alpha)  # the op's registered default
            if abs(alpha - LEAK) > 1e-6:
                raise IOError(f"{n['name']}: LeakyRelu alpha {alpha}; the engine implements leak {LEAK} (layers.py:10)")
        elif n["op"] == "Minimum":
            # leak * minimum(0, x): the Minimum compares with the constant 0 and its only consumer multiplies by the leak
            consts = [graph.const_of(r)[1] for r in n["input"] if not r.startswith("^")]
            if not any(c is not None and c.size == 1 and float(c.reshape(-1)[0]) == 0.0 for c in consts):
                raise IOError(f"{n['name']}: Minimum without a constant 0 operand is not layers.leaky_relu (layers.py:10-30)")
            cons = graph.data_consumers(n)
            leak = None
            if len(cons) == 1 and cons[0]["op"] == "Mul":
                for r in cons[0]["input"]:
                    if not r.startswith("^") and graph.base(r) != n["name"]:
                        _, v = graph.const_of(r)
                        if v is not None and v.size == 1:
                            leak = float(v.reshape(-1)[0])
            if leak is None:
                raise IOError(f"{n['name']}: no constant leak factor behind the Minimum (layers.py:10-30)")
            if abs(leak - LEAK) > 1e-6:
                raise IOError(f"{n['name']}: leak {leak}; the engine implements leak {LEAK} (layers.py:10)")
    return "leaky"


class _Layer:
    __slots__ = ("filter_name", "kind", "w", "b", "bias_name", "scale", "shift", "depth", "preds", "ops", "tail")

    def __init__(self, filter_name, kind, w):
        self.filter_name, self.kind, self.w = filter_name, kind, w
        self.b = self.bias_name = self.scale = self.shift = None
        self.depth, self.preds, self.ops, self.tail = 0, set(), [], None

    @property
    def cout(self):
        return int(self.w.shape[3] if self.kind == "conv" else self.w.shape[2])


class _Graph:
    def __init__(self, nodes):
        self.nodes = {}
        for n in nodes:
            if n["name"] in self.nodes:
                raise IOError(f"duplicate node name {n['name']} in the GraphDef")
            self.nodes[n["name"]] = n
        self.consumers = {}
        for n in nodes:
            for ref in n["input"]:
                if not ref.startswith("^"):
                    self.consumers.setdefault(self.base(ref), []).append(n)

    @staticmethod
    def base(ref):
        return ref.lstrip("^").split(":")[0]

    def src(self, ref):
        n = self.nodes.get(self.base(ref))
        if n is None:
            raise IOError(f"input {ref} refers to a node that is not in the graph")
        return n

    def skip_pass(self, node):
        hops = 0
        while node["op"] in _PASS_OPS and node["input"] and hops < 256:
            node = self.src(node["input"][0])
            hops += 1
        return node

    def const_of(self, ref):
        """(name, ndarray) of the constant ``ref`` resolves to through pass-through ops / Fill, else (None, None)"""
        n = self.skip_pass(self.src(ref))
        if n["op"] == "Const" and n["value"] is not None:
            return n["name"], n["value"]
        if n["op"] == "Fill" and len(n["input"]) == 2:
            _, dims = self.const_of(n["input"][0])
            _, val = self.const_of(n["input"][1])
            if dims is not None and val is not None and val.size == 1:
                return n["name"], np.full([int(d) for d in dims.reshape(-1)], val.reshape(-1)[0], dtype=val.dtype)
        return None, None

    def data_inputs(self, node):
        """references that carry image data into ``node`` (no control edges, shape operands or constants)"""
        op, ins = node["op"], [r for r in node["input"] if not r.startswith("^")]
        if op in _SHAPE_OPS or op in ("Const", "Placeholder", "Fill"):
            return []
        if op == "Conv2D":
            return ins[:1]
        if op == "Conv2DBackpropInput":
            return ins[2:3]
        if op == "ConcatV2":
            return ins[:-1]
        if op == "Split":
            return ins[1:2]
        if op in ("SplitV", "StridedSlice", "Mean", "Reshape", "ExpandDims", "Squeeze"):
            return ins[:1]
        if op in _BN_OPS:
            return ins[:1]
        return ins

    def data_consumers(self, node):
        out = []
        for c in self.consumers.get(node["name"], []):
            if c["op"] in _SHAPE_OPS:
                continue
            if any(self.base(r) == node["name"] for r in self.data_inputs(c)):
                out.append(c)
        return out


def _attr_list(node, key, default):
    v = node["attr"].get(key)
    return list(v) if isinstance(v, list) and v else list(default)


def _check_conv_attrs(node, kind):
    a = node["attr"]
    fmt = a.get("data_format") or "NHWC"
    if fmt != "NHWC":
        raise IOError(f"{node['name']}: data_format {fmt}; the engine's layout is NHWC")
    if (a.get("padding") or "SAME") != "SAME":
        raise IOError(f"{node['name']}: padding {a.get('padding')}; every ARU_v1 layer uses SAME")
    want = [1, 1, 1, 1] if kind == "conv" else [1, 2, 2, 1]
    if _attr_list(node, "strides", want) != want:
        raise IOError(f"{node['name']}: strides {a.get('strides')} (expected {want})")
    if _attr_list(node, "dilations", [1, 1, 1, 1]) != [1, 1, 1, 1]:
        raise IOError(f"{node['name']}: dilated convolutions are not part of ARU_v1")


def _vec(graph, ref, n):
    """constant vector of length n (or a scalar) behind ``ref`` -> float64 [n], else None"""
    _, v = graph.const_of(ref)
    if v is None or v.dtype.kind != "f" or v.size not in (1, n) or (v.ndim > 1 and v.size != max(v.shape)):
        return None
    return np.broadcast_to(v.reshape(-1).astype(np.float64), (n,)).copy()


def _read_layer_tail(graph, layer, op):
    """bias / batch-norm constants behind one use of the layer; returns the last node of the conv -> bias -> bn run"""
    n = layer.cout
    cur, bias, scale, shift, bias_name = op, None, None, None, None

    def sole(node, ops):
        cs = [c for c in graph.data_consumers(node) if c["op"] in ops]
        return cs[0] if len(cs) == 1 and len(graph.data_consumers(node)) == 1 else None

    c = sole(cur, _ADD_OPS)
    if c is not None:
        other = [r for r in c["input"] if not r.startswith("^") and graph.base(r) != cur["name"]]
        v = _vec(graph, other[0], n) if len(other) == 1 else None
        if v is not None:
            bias, bias_name, cur = v, graph.const_of(other[0])[0], c
    c = sole(cur, _BN_OPS)
    if c is not None:
        if c["attr"].get("is_training") is True:
            raise IOError(f"{c['name']}: batch normalisation in training mode; freeze the graph for inference")
        parts = [_vec(graph, r, n) for r in c["input"][1:5]]
        if any(p is None for p in parts):
            raise IOError(f"{c['name']}: batch-norm statistics are not constants")
        gamma, beta, mean, var = parts
        eps = c["attr"].get("epsilon")
        eps = 1e-4 if eps is None else float(eps)             # the op's registered default (a graph written with
                                                              # strip_default_attrs omits it; 1e-3 is the Keras LAYER default and
                                                              # is always written out because it differs)
        scale = gamma / np.sqrt(var + eps)
        shift = beta - mean * scale
        cur = c
    else:
        c = sole(cur, ("Mul",))
        if c is not None:
            other = [r for r in c["input"] if graph.base(r) != cur["name"]]
            v = _vec(graph, other[0], n) if len(other) == 1 else None
            c2 = sole(c, _ADD_OPS) if v is not None else None
            if c2 is not None:
                other2 = [r for r in c2["input"] if graph.base(r) != c["name"]]
                v2 = _vec(graph, other2[0], n) if len(other2) == 1 else None
                if v2 is not None:
                    scale, shift, cur = v, v2, c2
    return cur, bias, bias_name, scale, shift


def _collect_layers(graph, on_path):
    layers, op_layer = OrderedDict(), {}
    for node in on_path:
        if node["op"] not in ("Conv2D", "Conv2DBackpropInput"):
            continue
        kind = "conv" if node["op"] == "Conv2D" else "deconv"
        fname, w = graph.const_of(node["input"][1])
        if w is None or w.ndim != 4:
            raise IOError(f"{node['name']}: the filter is not a 4-D constant (is this graph frozen?)")
        if kind == "deconv" and np.all(w == 1.0) and w.shape[0] == w.shape[1]:
            continue                                        # upsample_simple (layers.py:716-720): not a parameter
        _check_conv_attrs(node, kind)
        layer = layers.get(fname)
        if layer is None:
            layer = layers[fname] = _Layer(fname, kind, w)
        elif layer.kind != kind:
            raise IOError(f"constant {fname} is used by a Conv2D and by a Conv2DBackpropInput")
        tail, bias, bias_name, scale, shift = _read_layer_tail(graph, layer, node)
        if layer.ops:
            if bias_name != layer.bias_name or (scale is None) != (layer.scale is None):
                raise IOError(f"the uses of filter {fname} do not share one bias / batch norm")
        else:
            layer.b, layer.bias_name, layer.scale, layer.shift = bias, bias_name, scale, shift
        layer.ops.append(node)
        op_layer[node["name"]] = layer
    return layers, op_layer


def _path_nodes(graph, start, end):
    """nodes on a data path start -> end (both included), in an order where producers come first"""
    fwd, stack = set(), [start]
    while stack:
        n = stack.pop()
        if n["name"] in fwd:
            continue
        fwd.add(n["name"])
        stack.extend(graph.data_consumers(n))
    order, seen = [], set()
    stack = [(end, False)]
    while stack:
        n, done = stack.pop()
        if done:
            order.append(n)
            continue
        if n["name"] in seen:
            continue
        seen.add(n["name"])
        stack.append((n, True))
        for r in graph.data_inputs(n):
            p = graph.src(r)
            if p["name"] in fwd and p["name"] not in seen:
                stack.append((p, False))
    if start["name"] not in seen:
        raise IOError(f"no data path from {start['name']} to {end['name']}")
    return order


def aru_from_topology(nodes, input_name="inImg", output_name="output"):
    """ARU_v1 weights + hyper-parameters from the op graph between ``inImg`` and ``output``, whatever the scopes are
    called (module docstring).  Returns (tensors under the engine's names, AruConfig)."""
    graph = _Graph(nodes)
    if input_name not in graph.nodes or output_name not in graph.nodes:
        raise IOError(f"the graph has no '{input_name}' / '{output_name}' node (net_post_processing_helper.py:69-70)")
    start, end = graph.nodes[input_name], graph.nodes[output_name]
    path = _path_nodes(graph, start, end)
    ops_on_path = {n["op"] for n in path}
    bad_act = sorted(_UNSUPPORTED_ACT[o] for o in ops_on_path if o in _UNSUPPORTED_ACT)
    if bad_act:
        raise IOError(f"activation {bad_act} on the path: ARU_v1 knows relu, elu and leaky (ARU_v1.py:70-75)")
    mvn = bool(ops_on_path & {"Enter", "Mean", "Sqrt", "Rsqrt", "RealDiv"})
    activation_name = _activation_of(graph, path, ops_on_path, mvn)
    unknown = sorted(ops_on_path - _KNOWN_OPS - (_MVN_OPS if mvn else set()) - _ACT_VARIANT_OPS.get(activation_name, set()))
    if unknown:
        raise IOError(f"ops {unknown} between {input_name} and {output_name} are not part of the ARU_v1 family")
    for n in path:
        if n["op"] in ("MaxPool", "AvgPool"):
            if _attr_list(n, "ksize", [1, 2, 2, 1]) != [1, 2, 2, 1] or _attr_list(n, "strides", [1, 2, 2, 1]) != [1, 2, 2, 1] \
                    or (n["attr"].get("padding") or "SAME") != "SAME":
                raise IOError(f"{n['name']}: only 2x2 / stride 2 / SAME pooling is supported")
    layers, op_layer = _collect_layers(graph, path)
    if not layers:
        raise IOError("no convolution with a constant filter between inImg and output")
    # conv depth of every node on the path and, per layer, the layers whose output reaches it without another layer
    depth, reach = {}, {}
    for n in path:                                          # producers first
        d, rs = 0, set()
        for r in graph.data_inputs(n):
            p = graph.src(r)
            if p["name"] not in depth:
                continue
            d = max(d, depth[p["name"]])
            rs |= reach[p["name"]]
        lay = op_layer.get(n["name"])
        if lay is not None:
            if lay.depth and lay.depth != d + 1:
                raise IOError(f"filter {lay.filter_name} is used at conv depths {lay.depth} and {d + 1}")
            lay.depth = d + 1
            lay.preds |= rs
            d, rs = d + 1, {lay.filter_name}
        depth[n["name"]], reach[n["name"]] = d, rs
    by_depth = {}
    for lay in layers.values():
        by_depth.setdefault(lay.depth, []).append(lay)

    def follow(root):
        chain = [root]
        while True:
            nxt = [c for c in by_depth.get(chain[-1].depth + 1, []) if chain[-1].filter_name in c.preds]
            if len(nxt) != 1:
                if len(nxt) > 1:
                    raise IOError(f"two layers at conv depth {chain[-1].depth + 1} consume {chain[-1].filter_name}")
                return chain
            chain.append(nxt[0])

    roots = by_depth.get(1, [])
    if not 1 <= len(roots) <= 2:
        raise IOError(f"{len(roots)} different filters are applied to the image; ARU_v1 has one (RU) or two (ARU)")
    chains = sorted((follow(r) for r in roots), key=len)
    last = max(layers.values(), key=lambda l: l.depth)
    if sum(1 for l in layers.values() if l.depth == last.depth) != 1 or chains[-1][-1] is not last:
        raise IOError("the graph does not end in one classification convolution")
    det = chains[-1][:-1]
    att = []
    if len(chains) == 2:
        att = chains[0][:-1] if chains[0][-1] is last else chains[0]
        if len(att) != 4 or att[-1].cout != 1 or any(l.kind != "conv" for l in att):
            raise IOError("the second branch on the image is not the 4-layer attention CNN of ARU_v1.py:165-184")
    if len(det) + len(att) + 1 != len(layers):
        stray = [l.filter_name for l in layers.values() if l not in det and l not in att and l is not last]
        raise IOError(f"layers outside the U-Net chain / attention chain: {stray[:4]}")
    # hyper-parameters from the chain's filter shapes
    k, _, channels, feat_root = (int(v) for v in det[0].w.shape)
    res_depth = 0
    while 1 + res_depth < len(det) and det[1 + res_depth].kind == "conv" and \
            tuple(det[1 + res_depth].w.shape) == (k, k, feat_root, feat_root):
        res_depth += 1
    n_dec = sum(1 for l in det if l.kind == "deconv")
    n_levels = n_dec + 1
    if res_depth < 1 or len(det) != n_levels * (1 + res_depth) + n_dec * (2 + res_depth):
        raise IOError(f"{len(det)} U-Net layers do not form {n_levels} levels of 1 + {res_depth} residual convolutions")
    n_scales = len(det[0].ops)
    if att and len(att[0].ops) != n_scales:
        raise IOError(f"attention CNN applied {len(att[0].ops)} times, U-Net {n_scales} times")
    if not att and n_scales != 1:
        raise IOError("a graph without attention branch applies the U-Net once")
    # class softmax between the last layer and `output`?
    node, softmax = graph.skip_pass(end), False
    if node["op"] == "Softmax":
        softmax, node = True, graph.skip_pass(graph.src(node["input"][0]))
    if node is not last.ops[0] and node is not _read_layer_tail(graph, last, last.ops[0])[0]:
        raise IOError(f"'{output_name}' is not the (softmax of the) classification layer but {node['op']} {node['name']}")
    # graph 'U' (ARU_v1.py:228-233: conv1 + conv2 per block, no residual add) has the layer list of a residual graph with
    # res_depth 1; the wiring tells them apart: in 'U' the block's output is conv2 alone, in 'RU' the sum conv1 + convR_0
    plain_u = False
    if res_depth == 1 and not att and len(det) > 1:
        nxt = det[2] if len(det) > 2 else last               # the layer that consumes the first block's output
        plain_u = nxt.preds == {det[1].filter_name}
    cfg = AruConfig(graph="ARU" if att else ("U" if plain_u else "RU"), channels=channels, n_classes=last.cout, feat_root=feat_root,
                    scale_space_num=n_levels, res_depth=3 if plain_u else res_depth, num_scales_att=n_scales, filter_size=k, mvn=mvn,
                    apply_softmax=softmax, activation_name=activation_name)
    if activation_name != "relu" and not plain_u:
        # ARU_v1.py:214,268: the activation behind conv1 of a residual block is layers.relu in every variant -- one Relu op per
        # block and pyramid scale, and no other
        want_relu = (n_levels + n_dec) * n_scales
        n_relu = sum(1 for n in path if n["op"] == "Relu")
        if n_relu != want_relu:
            raise IOError(f"{n_relu} Relu ops beside the {activation_name} activations; ARU_v1 has exactly one per residual block "
                          f"and scale ({want_relu}: the ReLU behind conv1, ARU_v1.py:214) -- the activations are not placed like ARU_v1's")
    elif activation_name != "relu" and any(n["op"] == "Relu" for n in path):
        raise IOError(f"Relu ops in a 'U' graph with {activation_name} activations (ARU_v1.py:228-233 uses the graph's activation only)")
    # assign by position and verify the skip / residual wiring against the template
    names = list(aru_tensor_shapes(cfg))
    wnames = [n for n in names if n.endswith("/weights")]
    ordered = att + det + [last]
    if len(wnames) != len(ordered):
        raise IOError(f"{len(ordered)} layers in the graph, {len(wnames)} in ARU_v1 with {cfg}")
    shapes = aru_tensor_shapes(cfg)
    tensors = OrderedDict()
    name_of = {}
    for wname, lay in zip(wnames, ordered):
        scope = wname[:-len("/weights")]
        if ("/deconv" in scope) != (lay.kind == "deconv"):
            raise IOError(f"{scope}: expected a {'transposed ' if '/deconv' in scope else ''}convolution at conv depth "
                          f"{lay.depth}, found {lay.ops[0]['op']} ({lay.filter_name})")
        if tuple(lay.w.shape) != tuple(shapes[wname]):
            raise IOError(f"{scope}: filter {lay.filter_name} has shape {tuple(lay.w.shape)}, ARU_v1 expects "
                          f"{tuple(shapes[wname])}")
        name_of[lay.filter_name] = scope
        w = lay.w.astype(np.float64)
        b = lay.b if lay.b is not None else np.zeros(lay.cout)
        if lay.scale is not None:                          # y = s * (conv + b) + t, folded
            w = w * (lay.scale[None, None, None, :] if lay.kind == "conv" else lay.scale[None, None, :, None])
            b = b * lay.scale + lay.shift
        tensors[wname] = np.ascontiguousarray(w, dtype=np.float32)
        bname = scope + ("/bias" if lay.kind == "deconv" else "/biases")
        tensors[bname] = np.ascontiguousarray(b, dtype=np.float32)
    tensors = OrderedDict((n, tensors[n]) for n in names)
    _check_wiring(cfg, ordered, name_of)
    return tensors, cfg


def _check_wiring(cfg, ordered, name_of):
    """which layers feed which (through bias / relu / pool / add / concat only) must be ARU_v1's residual U-Net"""
    got = {name_of[l.filter_name]: {name_of[p] for p in l.preds} for l in ordered}
    n, R = cfg.scale_space_num, cfg.res_depth
    want = {}
    block_out = None                                        # layers whose sum is the previous block's output
    det = "aru_net/featMapG/"
    plain_u = not cfg.use_residual                          # ARU_v1.py:228-233,283-288: conv1 -> conv2, the block's output is conv2
    for l in range(n):
        s = f"{det}unet_down_{l}"
        want[s + "/conv1"] = set(block_out or ())
        if plain_u:
            want[s + "/conv2"] = {s + "/conv1"}
            block_out = {s + "/conv2"}
            continue
        for r in range(R):
            want[s + f"/convR_{r}"] = {s + ("/conv1" if r == 0 else f"/convR_{r - 1}")}
        block_out = {s + "/conv1", s + f"/convR_{R - 1}"}
    for l in range(n - 2, -1, -1):
        s, skip = f"{det}unet_up_{l}", f"{det}unet_down_{l}"
        want[s + "/deconv"] = set(block_out)
        if plain_u:
            want[s + "/conv1"] = {s + "/deconv", skip + "/conv2"}
            want[s + "/conv2"] = {s + "/conv1"}
            block_out = {s + "/conv2"}
            continue
        want[s + "/conv1"] = {s + "/deconv", skip + "/conv1", skip + f"/convR_{R - 1}"}
        for r in range(R):
            want[s + f"/convR_{r}"] = {s + ("/conv1" if r == 0 else f"/convR_{r - 1}")}
        block_out = {s + "/conv1", s + f"/convR_{R - 1}"}
    logit_in = set(block_out)
    if cfg.use_attention:
        a = "aru_net/attMapG/attPart/conv"
        want[a + "1"] = set()
        for i in (2, 3, 4):
            want[a + str(i)] = {a + str(i - 1)}
        logit_in.add(a + "4")
    want["aru_net/logit/class"] = logit_in
    for scope, preds in want.items():
        if got.get(scope) != preds:
            raise IOError(f"{scope} is fed by {sorted(got.get(scope, ()))} in the graph, ARU_v1 ('{cfg.graph}') wires it to {sorted(preds)}")


def aru_from_nodes(nodes, num_scales_att=None, apply_softmax=None):
    """Frozen ARU-Net -> (tensors, AruConfig).  A graph with its op structure is mapped by topology; when it also
    carries ARU_v1's variable names both mappings must agree.  A constants-only container is mapped by name."""
    has_ops = any(n["op"] == "Conv2D" for n in nodes)
    named = any(n["op"] == "Const" and n["name"].endswith("aru_net/featMapG/unet_down_0/conv1/weights") for n in nodes)
    if not has_ops:
        return aru_from_constants(nodes, num_scales_att, apply_softmax)
    tensors, cfg = aru_from_topology(nodes)
    if num_scales_att is not None and int(num_scales_att) != cfg.num_scales_att:
        raise IOError(f"num_scales_att={num_scales_att} given, the graph applies the nets to {cfg.num_scales_att} scales")
    if apply_softmax is not None and bool(apply_softmax) != cfg.apply_softmax:
        raise IOError(f"apply_softmax={apply_softmax} given, the graph says {cfg.apply_softmax}")
    if named:
        by_name, _ = aru_from_constants(nodes, cfg.num_scales_att, cfg.apply_softmax)
        folded = any(n["op"] in _BN_OPS for n in nodes)
        for k in tensors:
            if not folded and not np.array_equal(by_name[k], tensors[k]):
                raise IOError(f"{k}: the constant of that name is not the one the op graph uses at this position")
    return tensors, cfg


_PASS_THROUGH = ("Identity", "Enter", "Switch", "Merge", "NextIteration", "Exit", "StopGradient", "PlaceholderWithDefault")


def _find_key(consts, suffix):
    hits = [k for k in consts if k == suffix or k.endswith("/" + suffix)]
    hits.sort(key=len)
    return hits[0] if hits else None


def _count_matmul_users(nodes, const_name):
    """MatMul ops that read the constant ``const_name`` (through /read identities and the Enter ops of a while loop): the edge
    MLP of the message function is built once per transition step (graph_gnn.py:134-157 unrolls the steps in Python, the
    variables are shared with AUTO_REUSE, message_fn_chunk.py:356-363), so this is ``num_transition_steps``."""
    users = {}
    for n in nodes:
        for r in n.get("input", []):
            base = r.lstrip("^").split(":")[0]
            users.setdefault(base, []).append(n)
    seen, todo, hits = {const_name}, [const_name], set()
    while todo:
        cur = todo.pop()
        for n in users.get(cur, []):
            if n["op"] in ("MatMul", "BatchMatMul", "BatchMatMulV2"):
                hits.add(n["name"])
            elif n["op"] in _PASS_THROUGH and n["name"] not in seen:
                seen.add(n["name"])
                todo.append(n["name"])
    return len(hits)


def gnn_from_nodes(nodes, undirected_graph=True, visual_layers=None, num_transition_steps=None):
    """GraphDef of a relation net (model_relation.py / graph_relation.py / graph_gnn.py) -> (tensors, GnnConfig).

    Widths come from the constants' shapes; the options an exporter can bake into the graph are read from its STRUCTURE and
    either served or refused with the reason (never silently assumed):
      num_transition_steps (graph_gnn.py:19)        = number of MatMul ops on the edge MLP's first layer; a constants-only
                                                      container carries it as ``asep_meta/num_transition_steps`` or takes the argument
      compress_node_feature_dim (graph_gnn.py:20)   = GraphLSTM1/compress_input/ff_compress_input/weights present -> served
      output_type add / concat (graph_gnn.py:23)    = GraphLSTM1/dense/weights present / a classifier input of 2 x (hidden + fed width) -> served
      use_attention / heads / merge (message_fn_chunk.py:35-41) = .../head_<k>/calculation_unnormalized_attention_values/... variables,
                                                      AddN (average) or ConcatV2 (concat) of the heads -> served (graphs of <= 316 nodes)
      aggregation_type (message_fn_chunk.py:16,57-62) = the op that reduces the sparse [from, to] tensor of attenuated features:
                                                      SparseReduceMax -> 'max', else 'sum' -> both served
      num_hidden_units_interaction_fct / _attention_fct / classifier num_hidden_units (lists) = the fully_connected_layer_h<i> variables
                                                      of each MLP -> up to four hidden layers served, more refused
      incorporate_*_in_update (update_fn_lstm.py:13-16,43-50) = number of tensors the gates' ConcatV2 joins, checked against the gate
                                                      weights' input width x + [h] + [u] -> served; an ambiguous layout is refused
      assign_visual_features_to_edges (graph_relation.py:141-172) = visual_edge_feature_compression_fm_<i> variables -> served
    """
    consts = const_tensors(nodes)
    pref = ("GraphLSTM1/message_fn_default/head_0/calculation_interaction_features/concat_u_and_h/"
            "interaction_features")
    w1 = _find(consts, pref + "/fully_connected_layer_h1/weights")
    wu = _find(consts, "GraphLSTM1/update_function_LSTM/ingate_activation/dense/weights")
    if w1 is None or wu is None:
        raise IOError("no GNN variables (GraphLSTM1/...) among the graph constants")
    # ---- attention (message_fn_chunk.py:35-41,167-245): heads = number of head_<k> scopes, use_attention = the attention MLP's
    #      variables exist, merge type = how the op graph combines the heads (AddN / RealDiv = 'average', ConcatV2 = 'concat';
    #      a constants-only container says it in asep_meta/attention_merge_average)
    heads = 0
    while any(f"/message_fn_default/head_{heads}/" in k for k in consts):
        heads += 1
    att_pref = ("GraphLSTM1/message_fn_default/head_0/calculation_unnormalized_attention_values/calculation_interaction_features/"
                "concat_u_and_h/interaction_features")
    wa1 = _find(consts, att_pref + "/fully_connected_layer_h1/weights")
    use_attention = wa1 is not None
    if not use_attention and any("calculation_unnormalized_attention_values" in k for k in consts):
        raise IOError("attention variables (calculation_unnormalized_attention_values) under an unexpected scope; expected "
                      + att_pref)
    if heads > 1 and not use_attention:
        raise IOError("several head_<k> scopes without attention variables: message_fn_chunk.py:167-169 builds one head then")
    merge = "concat"
    if use_attention:
        meta_avg = _find(consts, "asep_meta/attention_merge_average")
        if meta_avg is not None:
            merge = "average" if int(np.asarray(meta_avg).reshape(-1)[0]) else "concat"
        elif any(n["op"] == "AddN" and "message_fn_default" in n["name"] for n in nodes):
            merge = "average"                               # message_fn_chunk.py:229-233 tf.add_n(...) / num_attention_heads

    def hidden_widths(scope, what):
        """widths of the fully_connected_layer_h<i> variables under ``scope`` (layers.py:477-480)"""
        out = []
        while _find(consts, f"{scope}/fully_connected_layer_h{len(out) + 1}/weights") is not None:
            out.append(int(_find(consts, f"{scope}/fully_connected_layer_h{len(out) + 1}/weights").shape[1]))
        if len(out) > 4:
            raise IOError(f"{what} has {len(out)} hidden layers: the engine serves up to four")
        return out
    int_hidden = hidden_widths(pref, "the interaction MLP (num_hidden_units_interaction_fct)")
    att_hidden = hidden_widths(att_pref, "the attention MLP (num_hidden_units_attention_fct)") if use_attention else [16]
    # ---- aggregation (message_fn_chunk.py:57-62,398-417): tf.sparse.reduce_max / reduce_sum of the attenuated features (the degree
    #      count of the balanced weighting is always a SparseReduceSum); a constants-only container says it in asep_meta/aggregation_max
    meta_agg = _find(consts, "asep_meta/aggregation_max")
    if meta_agg is not None:
        aggregation = "max" if int(np.asarray(meta_agg).reshape(-1)[0]) else "sum"
    else:
        aggregation = "max" if any(n["op"] == "SparseReduceMax" and "message_fn_default" in n["name"] for n in nodes) else "sum"
    w_add = _find(consts, "GraphLSTM1/dense/weights")       # graph_gnn.py:160-163 output_type='add_final_hidden_and_input'
    hidden = int(wu.shape[1])
    w2 = _find(consts, pref + "/fully_connected_logit_layer_out/weights")
    inter = int(w2.shape[1]) if w2 is not None else hidden
    if use_attention and merge == "concat":
        inter *= heads                                      # message_fn_chunk.py:69-72,234-237: x = concat of heads x x_dim columns
    wcmp = _find(consts, "GraphLSTM1/compress_input/ff_compress_input/weights")
    # ---- what the LSTM gates read (update_fn_lstm.py:41-50): v = [x] + [h] + [u].  The gates' ConcatV2 says how many tensors are
    #      joined; which ones follows from the widths.  u is known independently when the graph carries the node_features
    #      placeholder's static shape or a compress_input layer; a layout that cannot be told apart is refused.
    v_dim = int(wu.shape[0])
    fed_ph = next((n for n in nodes if n["op"] == "Placeholder" and n["name"].split("/")[-1] == "node_features"), None)
    ph_dim = None
    if fed_ph is not None and isinstance(fed_ph["attr"].get("shape"), tuple) and fed_ph["attr"]["shape"][1]:
        last = fed_ph["attr"]["shape"][1][-1]
        ph_dim = int(last) if last is not None and int(last) > 0 else None
    vis_total = 0
    k = 0
    while _find(consts, f"visual_node_feature_compression_fm_{k}/dense/weights") is not None:
        vis_total += int(_find(consts, f"visual_node_feature_compression_fm_{k}/dense/weights").shape[1])
        k += 1
    u_known = int(wcmp.shape[1]) if wcmp is not None else (ph_dim + vis_total if ph_dim is not None else None)
    gate_cat = next((n for n in nodes if n["op"] == "ConcatV2" and "update_function_LSTM" in n["name"] and "ingate_activation" in n["name"]), None)
    n_join = int(gate_cat["attr"].get("N", 3)) if gate_cat is not None else None
    meta_h, meta_u = _find(consts, "asep_meta/lstm_use_hidden"), _find(consts, "asep_meta/lstm_use_input")
    if meta_h is not None or meta_u is not None:
        use_h = bool(int(np.asarray(meta_h).reshape(-1)[0])) if meta_h is not None else True
        use_u = bool(int(np.asarray(meta_u).reshape(-1)[0])) if meta_u is not None else True
    elif n_join is None or n_join == 3:
        use_h = use_u = True
    elif n_join == 1:
        use_h = use_u = False
    elif n_join == 2:
        rest = v_dim - inter
        if u_known is None:
            raise IOError("the LSTM gates join two tensors (one of incorporate_hidden_features_in_update / "
                          "incorporate_node_input_features_in_update is off, update_fn_lstm.py:43-50) and the graph does not say how wide "
                          "the node features are: cannot tell which")
        if rest == hidden and rest != u_known:
            use_h, use_u = True, False
        elif rest == u_known and rest != hidden:
            use_h, use_u = False, True
        else:
            raise IOError(f"the LSTM gates read {v_dim} = x ({inter}) + {rest} values: neither clearly h ({hidden}) nor u ({u_known})")
    else:
        raise IOError(f"the LSTM gates join {n_join} tensors; update_fn_lstm.py:41-50 joins x, [h], [u]")
    if use_u:
        u_dim = v_dim - inter - (hidden if use_h else 0)
    elif u_known is not None:
        u_dim = u_known
    else:
        raise IOError("the gates do not read the node features and the graph does not say how wide they are (no node_features shape, no "
                      "compress_input layer)")
    if v_dim != inter + (hidden if use_h else 0) + (u_dim if use_u else 0) or u_dim < 0:
        raise IOError(f"LSTM gate input width {v_dim} is not x ({inter})" + (f" + h ({hidden})" if use_h else "") + (f" + u ({u_dim})" if use_u else ""))
    if u_known is not None and u_known != u_dim:
        raise IOError(f"the gates' input width says {u_dim} node features, the graph feeds {u_known} "
                      "(update_fn_lstm.py:41-50: v = [x, h, u])")
    e_dim = int(w1.shape[0]) - 4 * u_dim - 4 * hidden
    if u_dim < 0 or e_dim < 0:
        raise IOError(f"inconsistent GNN constant shapes: update input {wu.shape[0]}, edge MLP input {w1.shape[0]}, hidden {hidden}")
    compress = 0
    u_in = u_dim
    if wcmp is not None:
        if int(wcmp.shape[1]) != u_dim:
            raise IOError(f"compress_input layer maps to {wcmp.shape[1]} features, the GNN reads {u_dim}")
        compress, u_in = u_dim, int(wcmp.shape[0])
    wc1 = _find(consts, "Classification/logits/fully_connected_layer_h1/weights")
    wo = _find(consts, "Classification/logits/fully_connected_logit_layer_out/weights")
    first = wc1 if wc1 is not None else wo
    if first is None:
        raise IOError("no classifier (Classification/logits/...) among the graph constants")
    output_type = "hidden"
    if w_add is not None:
        if tuple(w_add.shape) != (u_in, hidden):
            raise IOError(f"GraphLSTM1/dense/weights is {tuple(w_add.shape)}, output_type='add_final_hidden_and_input' "
                          f"(graph_gnn.py:160-163) projects the {u_in} fed features to {hidden}")
        output_type = "add_final_hidden_and_input"
    if int(first.shape[0]) == 2 * (hidden + u_in) and u_in > 0 and w_add is None:
        output_type = "concat_final_hidden_and_input"       # graph_gnn.py:164-166: the classifier pairs up [h | x]
    elif int(first.shape[0]) != 2 * hidden:
        raise IOError(f"the pair classifier reads {first.shape[0]} features per pair; 2 x hidden = {2 * hidden} or, with "
                      f"output_type='concat_final_hidden_and_input', 2 x (hidden + {u_in}) expected")
    # ---- number of transition steps: from the op graph, the container's metadata or the caller
    steps_graph = _count_matmul_users(nodes, _find_key(consts, pref + "/fully_connected_layer_h1/weights"))
    meta = _find(consts, "asep_meta/num_transition_steps")
    if steps_graph > 0:
        steps = steps_graph
        if num_transition_steps is not None and int(num_transition_steps) != steps:
            raise IOError(f"num_transition_steps={num_transition_steps} given, the graph unrolls {steps} transition steps")
    elif meta is not None:
        steps = int(np.asarray(meta).reshape(-1)[0])
    elif num_transition_steps is not None:
        steps = int(num_transition_steps)
    else:
        raise IOError("the container holds the GNN constants but no op graph: the number of transition steps (graph_gnn.py:19) is "
                      "not derivable from shared weights; pass num_transition_steps (or export with asep_meta/num_transition_steps)")
    cls_hidden = hidden_widths("Classification/logits", "the pair classifier (num_hidden_units)")
    if not cls_hidden:
        raise IOError("the pair classifier has no hidden layer (graph_relation.py:196 num_hidden_units): the engine evaluates the first "
                      "hidden layer per node and needs one")
    vis_kw = {}
    if any("visual_node_feature_compression" in k for k in consts):
        # graph exported with --image_input (graph_relation.py:17-37): backbone + one compression layer per map
        # the backbone's logits are not used by the relation graph (only its end points): no class softmax to look for
        _, bcfg = aru_from_constants(nodes, apply_softmax=False)
        dims, chans = [], []
        i = 0
        while _find(consts, f"visual_node_feature_compression_fm_{i}/dense/weights") is not None:
            wv = _find(consts, f"visual_node_feature_compression_fm_{i}/dense/weights")
            chans.append(int(wv.shape[0]))
            dims.append(int(wv.shape[1]))
            i += 1
        if visual_layers is None:
            # the from_layer names are not stored with the constants; assume the up-path block outputs of scale 0
            visual_layers = []
            for c in chans:
                lvl = int(round(np.log2(c / bcfg.feat_root)))
                visual_layers.append(f"scale_0_unet_up_{lvl}_conv")
        if len(visual_layers) != len(dims):
            raise IOError(f"{len(dims)} compression layers in the graph but {len(visual_layers)} visual_layers given")
        mvn = any("per_image_standardization" in n.get("name", "") and "aru_net" not in n.get("name", "")
                  for n in nodes)
        backbone = {k: v for k, v in bcfg.to_dict().items() if k not in ("apply_softmax", "mvn")}
        backbone["mvn"] = bool(bcfg.mvn)
        vis_kw = dict(visual_dims=dims, visual_layers=list(visual_layers), mvn=mvn, backbone=backbone)
        u_in -= sum(dims)
        if u_in < 0:
            raise IOError(f"the visual compression layers produce {sum(dims)} features, the GNN is fed {u_in + sum(dims)}")
        if any("visual_edge_feature_compression" in k for k in consts):
            # graph_relation.py:141-172 assign_visual_features_to_edges: the same feature maps and compressed widths (misc.py:384-470)
            edims = []
            while _find(consts, f"visual_edge_feature_compression_fm_{len(edims)}/dense/weights") is not None:
                edims.append(int(_find(consts, f"visual_edge_feature_compression_fm_{len(edims)}/dense/weights").shape[1]))
            if edims != dims:
                raise IOError(f"visual edge compression layers {edims} differ from the node ones {dims}: both come from layer_compressed_dim")
            vis_kw["visual_edges"] = True
            e_dim -= sum(edims)
            if e_dim < 0:
                raise IOError(f"the visual edge compression layers produce {sum(edims)} features, the edge MLP reads {e_dim + sum(edims)} edge features")
    elif any("visual_edge_feature_compression" in k for k in consts):
        raise IOError("visual EDGE compression layers without the node ones: graph_relation.py builds both from the same feature maps")
    cfg = GnnConfig(node_feature_dim=u_in, edge_feature_dim=e_dim, num_transition_steps=steps, hidden_dim=hidden,
                    interaction_dim=inter,
                    interaction_hidden=int_hidden, classifier_hidden=cls_hidden, aggregation_type=aggregation,
                    incorporate_hidden_features_in_update=use_h, incorporate_node_input_features_in_update=use_u,
                    num_classes=int(wo.shape[1]), undirected_graph=undirected_graph, compress_node_feature_dim=compress,
                    output_type=output_type, use_attention=use_attention, num_attention_heads=max(heads, 1) if use_attention else 1,
                    multihead_attention_merge_type=merge,
                    attention_hidden=att_hidden, **vis_kw)
    if vis_kw and cfg.visual_channels() != chans:
        raise IOError(f"visual_layers {cfg.visual_layers} have {cfg.visual_channels()} channels, the compression "
                      f"layers expect {chans}")
    tensors = OrderedDict()
    for name, shape in gnn_tensor_shapes(cfg).items():
        t = _find(consts, name)
        if t is None:
            raise IOError(f"frozen graph lacks the constant {name}")
        if tuple(t.shape) != tuple(shape):
            raise IOError(f"{name}: shape {tuple(t.shape)} in the graph, {tuple(shape)} expected")
        tensors[name] = np.ascontiguousarray(t, dtype=np.float32)
    return tensors, cfg


# ----------------------------------------------------------------------------------------------
# minimal encoder (used by the tests to synthesise frozen graphs; also handy to export engine weights as .pb)
# ----------------------------------------------------------------------------------------------
def _enc_varint(v):
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _enc_field(fno, payload: bytes):
    return _enc_varint((fno << 3) | 2) + _enc_varint(len(payload)) + payload


def _enc_tensor(arr, use_content=True):
    arr = np.asarray(arr)
    dtype = {np.dtype("float32"): DT_FLOAT, np.dtype("int32"): DT_INT32}[arr.dtype]
    shape = b"".join(_enc_field(2, _enc_varint((1 << 3) | 0) + _enc_varint(int(d))) for d in arr.shape)
    msg = _enc_varint((1 << 3) | 0) + _enc_varint(dtype) + _enc_field(2, shape)
    if use_content:
        msg += _enc_field(4, np.ascontiguousarray(arr).tobytes())
    elif dtype == DT_FLOAT:
        msg += _enc_field(5, np.ascontiguousarray(arr, dtype="<f4").tobytes())          # packed float_val
    else:
        msg += _enc_field(7, b"".join(_enc_varint(int(v)) for v in arr.reshape(-1)))
    return msg


def encode_graphdef(nodes):
    """nodes: iterable of dicts {'name', 'op', 'input': [...], 'value': ndarray (Const only), 'packed': bool}."""
    out = bytearray()
    for n in nodes:
        msg = _enc_field(1, n["name"].encode()) + _enc_field(2, n["op"].encode())
        for i in n.get("input", []):
            msg += _enc_field(3, i.encode())
        if n.get("value") is not None:
            attr = _enc_field(8, _enc_tensor(n["value"], use_content=not n.get("packed", False)))
            msg += _enc_field(5, _enc_field(1, b"value") + _enc_field(2, attr))
        out += _enc_field(1, msg)
    return bytes(out)


def weights_to_graphdef(tensors, prefix="graph/", extra_nodes=(), meta=None):
    """constants-only container; ``meta`` ({"num_transition_steps": 3, ...}) becomes int32 constants ``asep_meta/<key>`` (what the op
    graph of a real export says structurally and shared weights cannot)"""
    nodes = [{"name": "inImg", "op": "Placeholder"}]
    for k, v in (meta or {}).items():
        nodes.append({"name": prefix + "asep_meta/" + k, "op": "Const", "value": np.asarray([int(v)], np.int32)})
    for i, (name, arr) in enumerate(tensors.items()):
        nodes.append({"name": prefix + name, "op": "Const", "value": np.asarray(arr, np.float32), "packed": i % 2 == 1})
        nodes.append({"name": prefix + name + "/read", "op": "Identity", "input": [prefix + name]})
    nodes.extend(extra_nodes)
    return encode_graphdef(nodes)
