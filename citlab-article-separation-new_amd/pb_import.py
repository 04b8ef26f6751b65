"""TensorFlow-free reader of frozen TF1 graphs (``*.pb`` = serialized ``GraphDef``), SURVEY.md row f1.

The reference loads its nets with ``tf.GraphDef().ParseFromString`` + ``tf.import_graph_def``
(``net_post_processing_helper.py:36-53``, ``gnn/io.py:12-25``).  The frozen graphs were produced by
``convert_variables_to_constants`` (``gnn/model/model_base.py:473-476``), which turns every variable into a
``Const`` node *with the variable's name*.  This module decodes the protobuf wire format directly, extracts those
constants and maps them onto the engine's named weight set (``weights.py``) by the reference's variable-scope names
(any graph prefix such as ``graph/`` is ignored), and derives the model hyper-parameters from tensor shapes and the
op list.

Only the message fields needed for that are decoded (field numbers from tensorflow/core/framework/*.proto):
    GraphDef.node = 1 ; NodeDef{name=1, op=2, input=3, attr=5(map<string,AttrValue>)}
    AttrValue{tensor=8, type=6, shape=7, s=2, i=3, f=4, b=5}
    TensorProto{dtype=1, tensor_shape=2, tensor_content=4, float_val=5, double_val=6, int_val=7, int64_val=10}
    TensorShapeProto{dim=2{size=1}}
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


def parse_graphdef(data: bytes):
    """-> list of nodes: {'name', 'op', 'input': [...], 'value': ndarray or None}"""
    nodes = []
    for fno, wt, val in _fields(memoryview(data)):
        if fno != 1 or wt != 2:
            continue
        node = {"name": "", "op": "", "input": [], "value": None}
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
                if key == "value" and attr is not None and node["op"] in ("", "Const"):
                    for f4, w4, v4 in _fields(attr):
                        if f4 == 8 and w4 == 2:
                            node["value"] = _parse_tensor(v4)
        if node["op"] != "Const":
            node["value"] = None
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


def aru_from_nodes(nodes, num_scales_att=None, apply_softmax=None):
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
    w0 = _find(consts, "aru_net/featMapG/unet_down_0/conv1/weights")
    wl = _find(consts, "aru_net/logit/class/weights")
    if wl is None:
        raise IOError("aru_net/logit/class/weights missing")
    use_att = _find(consts, "aru_net/attMapG/attPart/conv1/weights") is not None
    if num_scales_att is None:
        # one AvgPool per extra scale of the image pyramid (ARU_v1.py:106-109)
        num_scales_att = 1 + sum(1 for n in nodes if n["op"] == "AvgPool" and "attMapG" in n["name"]) if use_att else 1
        if use_att and num_scales_att == 1:
            num_scales_att = 3
    if apply_softmax is None:
        out = [n for n in nodes if n["name"] == "output"]
        apply_softmax = True
        if out and out[0]["op"] not in ("Softmax", "Identity"):
            apply_softmax = False
    cfg = AruConfig(graph="ARU" if use_att else "RU", channels=int(w0.shape[2]), n_classes=int(wl.shape[3]),
                    feat_root=int(w0.shape[3]), scale_space_num=levels, res_depth=res_depth,
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


def gnn_from_nodes(nodes, undirected_graph=True, visual_layers=None):
    consts = const_tensors(nodes)
    pref = ("GraphLSTM1/message_fn_default/head_0/calculation_interaction_features/concat_u_and_h/"
            "interaction_features")
    w1 = _find(consts, pref + "/fully_connected_layer_h1/weights")
    wu = _find(consts, "GraphLSTM1/update_function_LSTM/ingate_activation/dense/weights")
    if w1 is None or wu is None:
        raise IOError("no GNN variables (GraphLSTM1/...) among the graph constants")
    hidden = int(wu.shape[1])
    u_dim = int(wu.shape[0]) - 2 * hidden                       # v = [x, h, u]
    e_dim = int(w1.shape[0]) - 4 * u_dim - 4 * hidden
    cls_hidden = []
    i = 1
    while _find(consts, f"Classification/logits/fully_connected_layer_h{i}/weights") is not None:
        cls_hidden.append(int(_find(consts, f"Classification/logits/fully_connected_layer_h{i}/weights").shape[1]))
        i += 1
    wo = _find(consts, "Classification/logits/fully_connected_logit_layer_out/weights")
    vis_kw = {}
    if any("visual_node_feature_compression" in k for k in consts):
        # graph exported with --image_input (graph_relation.py:17-37): backbone + one compression layer per map
        _, bcfg = aru_from_nodes(nodes)
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
        u_dim -= sum(dims)
    cfg = GnnConfig(node_feature_dim=u_dim, edge_feature_dim=e_dim, hidden_dim=hidden, interaction_dim=hidden,
                    interaction_hidden=[int(w1.shape[1])], classifier_hidden=cls_hidden,
                    num_classes=int(wo.shape[1]), undirected_graph=undirected_graph, **vis_kw)
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


def weights_to_graphdef(tensors, prefix="graph/", extra_nodes=()):
    nodes = [{"name": "inImg", "op": "Placeholder"}]
    for i, (name, arr) in enumerate(tensors.items()):
        nodes.append({"name": prefix + name, "op": "Const", "value": np.asarray(arr, np.float32), "packed": i % 2 == 1})
        nodes.append({"name": prefix + name + "/read", "op": "Identity", "input": [prefix + name]})
    nodes.extend(extra_nodes)
    return encode_graphdef(nodes)
