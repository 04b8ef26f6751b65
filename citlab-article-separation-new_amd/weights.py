"""Named fp32 weight sets for the two nets: seeded initialisation, (de)serialisation.

Tensor names follow the reference's TF variable scopes so that a frozen-graph importer can
fill the same dictionary ("Variable naming"):
  * ARU-Net: ``ARU_v1.py:77,105,119,140,172,210,253`` + ``layers.py:234-238,356-358``
    (conv: ``weights [kh,kw,Cin,Cout]`` / ``biases [Cout]``; deconv: ``weights [kh,kw,Cout,Cin]`` /
    ``bias [Cout]``).
  * GNN: ``graph_relation.py:181,194``, ``message_fn_chunk.py:167,175,253,356``,
    ``update_fn_lstm.py:32,55-66,93-97``, ``layers.py:475-486,91,104``.

Blob format ("ASEPW001"): little endian
    8s magic | u32 n_tensors | n x { u16 name_len | name | u8 ndim | ndim x u32 dims | f32 data }
"""
import struct
from collections import OrderedDict
from typing import Dict

import numpy as np

from .config import AruConfig, GnnConfig

MAGIC = b"ASEPW001"


# ----------------------------------------------------------------------------------------------
# tensor inventories
# ----------------------------------------------------------------------------------------------
def aru_tensor_shapes(cfg: AruConfig) -> "OrderedDict[str, tuple]":
    """All variables of the ARU-Net graph in creation order (ARU_v1.py:62-294)."""
    k = cfg.filter_size
    shapes = OrderedDict()
    if cfg.use_attention:
        cin = cfg.channels
        for i, f in enumerate((12, 16, 32, 1), start=1):          # ARU_v1.py:172-183
            p = f"aru_net/attMapG/attPart/conv{i}"
            shapes[p + "/weights"] = (4, 4, cin, f)
            shapes[p + "/biases"] = (f,)
            cin = f
    last = cfg.channels
    for l in range(cfg.scale_space_num):                           # ARU_v1.py:208-245
        f = cfg.feat(l)
        p = f"aru_net/featMapG/unet_down_{l}"
        shapes[p + "/conv1/weights"] = (k, k, last, f)
        shapes[p + "/conv1/biases"] = (f,)
        if cfg.use_residual:
            for r in range(cfg.res_depth):
                shapes[p + f"/convR_{r}/weights"] = (k, k, f, f)
                shapes[p + f"/convR_{r}/biases"] = (f,)
        else:                                                      # graph 'U': ARU_v1.py:228-233
            shapes[p + "/conv2/weights"] = (k, k, f, f)
            shapes[p + "/conv2/biases"] = (f,)
        last = f
    for l in range(cfg.scale_space_num - 2, -1, -1):                # ARU_v1.py:251-292
        f = cfg.feat(l)
        p = f"aru_net/featMapG/unet_up_{l}"
        shapes[p + "/deconv/weights"] = (k, k, f, last)             # [kh,kw,Cout,Cin]
        shapes[p + "/deconv/bias"] = (f,)
        shapes[p + "/conv1/weights"] = (k, k, 2 * f, f)
        shapes[p + "/conv1/biases"] = (f,)
        if cfg.use_residual:
            for r in range(cfg.res_depth):
                shapes[p + f"/convR_{r}/weights"] = (k, k, f, f)
                shapes[p + f"/convR_{r}/biases"] = (f,)
        else:                                                      # ARU_v1.py:283-288
            shapes[p + "/conv2/weights"] = (k, k, f, f)
            shapes[p + "/conv2/biases"] = (f,)
        last = f
    shapes["aru_net/logit/class/weights"] = (4, 4, cfg.feat_root, cfg.n_classes)   # ARU_v1.py:158
    shapes["aru_net/logit/class/biases"] = (cfg.n_classes,)
    return shapes


_MSG = ("GraphLSTM1/message_fn_default/head_0/calculation_interaction_features/"
        "concat_u_and_h/interaction_features")
_UPD = "GraphLSTM1/update_function_LSTM"
_CLS = "Classification/logits"
GATES = ("ingate", "outgate", "forgetgate", "cellinput")


def attention_scope(head: int) -> str:
    """variable scope of head k's attention MLP (message_fn_chunk.py:422 inside :174 inside :166; _get_interaction_features adds
    'calculation_interaction_features', _calculate_interaction_features_1 'concat_u_and_h', layers.mlp 'interaction_features')"""
    return (f"GraphLSTM1/message_fn_default/head_{head}/calculation_unnormalized_attention_values/"
            "calculation_interaction_features/concat_u_and_h/interaction_features")


def gnn_tensor_shapes(cfg: GnnConfig) -> "OrderedDict[str, tuple]":
    shapes = OrderedDict()
    if cfg.visual_dims:
        if len(cfg.visual_layers) != len(cfg.visual_dims):
            raise ValueError("visual_layers and visual_dims must have the same length")
        for name, shp in aru_tensor_shapes(cfg.backbone_cfg()).items():   # backbone variables live in the same graph
            shapes[name] = shp
        for i, (c, d) in enumerate(zip(cfg.visual_channels(), cfg.visual_dims)):      # misc.py:365-368
            shapes[f"visual_node_feature_compression_fm_{i}/dense/weights"] = (c, d)
            shapes[f"visual_node_feature_compression_fm_{i}/dense/bias"] = (d,)
        if cfg.visual_edges:                                        # misc.py:465-468: the edges' own compression layers
            for i, (c, d) in enumerate(zip(cfg.visual_channels(), cfg.visual_dims)):
                shapes[f"visual_edge_feature_compression_fm_{i}/dense/weights"] = (c, d)
                shapes[f"visual_edge_feature_compression_fm_{i}/dense/bias"] = (d,)
    if cfg.compress_node_feature_dim > 0:                           # graph_gnn.py:102-109 (scope GraphLSTM1/compress_input)
        shapes["GraphLSTM1/compress_input/ff_compress_input/weights"] = (cfg.u_in_dim, cfg.compress_node_feature_dim)
        shapes["GraphLSTM1/compress_input/ff_compress_input/bias"] = (cfg.compress_node_feature_dim,)
    for k in range(cfg.heads):                                      # message_fn_chunk.py:172-176: one scope per attention head
        msg = _MSG.replace("head_0", f"head_{k}")
        d_in = cfg.message_in_dim
        for i, h in enumerate(cfg.interaction_hidden, start=1):     # layers.py:477-480
            shapes[f"{msg}/fully_connected_layer_h{i}/weights"] = (d_in, h)
            shapes[f"{msg}/fully_connected_layer_h{i}/bias"] = (h,)
            d_in = h
        shapes[f"{msg}/fully_connected_logit_layer_out/weights"] = (d_in, cfg.head_interaction_dim)
        shapes[f"{msg}/fully_connected_logit_layer_out/bias"] = (cfg.head_interaction_dim,)
        if cfg.use_attention:                                       # message_fn_chunk.py:420-446: the attention MLP, one output
            att = attention_scope(k)
            d_in = cfg.message_in_dim
            for i, h in enumerate(cfg.attention_hidden, start=1):
                shapes[f"{att}/fully_connected_layer_h{i}/weights"] = (d_in, h)
                shapes[f"{att}/fully_connected_layer_h{i}/bias"] = (h,)
                d_in = h
            shapes[f"{att}/fully_connected_logit_layer_out/weights"] = (d_in, 1)
            shapes[f"{att}/fully_connected_logit_layer_out/bias"] = (1,)
    for g in GATES:                                                 # update_fn_lstm.py:55-66
        shapes[f"{_UPD}/{g}_activation/dense/weights"] = (cfg.update_in_dim, cfg.hidden_dim)
        shapes[f"{_UPD}/{g}_activation/dense/bias"] = (cfg.hidden_dim,)
    if cfg.output_type_code == 1:                                   # graph_gnn.py:160-163: ff_layer(no bias) under the GNN's scope
        shapes["GraphLSTM1/dense/weights"] = (cfg.u_in_dim, cfg.hidden_dim)
    d_in = 2 * cfg.classifier_node_dim                               # graph_relation.py:253-266 (graph_gnn.py:164-166: [h | x])
    for i, h in enumerate(cfg.classifier_hidden, start=1):
        shapes[f"{_CLS}/fully_connected_layer_h{i}/weights"] = (d_in, h)
        shapes[f"{_CLS}/fully_connected_layer_h{i}/bias"] = (h,)
        d_in = h
    shapes[f"{_CLS}/fully_connected_logit_layer_out/weights"] = (d_in, cfg.num_classes)
    shapes[f"{_CLS}/fully_connected_logit_layer_out/bias"] = (cfg.num_classes,)
    return shapes


# ----------------------------------------------------------------------------------------------
# seeded initialisation with the reference's initialiser rule
# ----------------------------------------------------------------------------------------------
def _init_from_shapes(shapes, seed: int, bias_value: float = 0.1) -> "OrderedDict[str, np.ndarray]":
    """conv/deconv: N(0, sqrt(2/(kh*kw*d2+d3))) (layers.py:223-225,346-348); ff: N(0, sqrt(2/(in+out)))
    (layers.py:80-82); all biases 0.1 (layers.py:64,199,344)."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, shp in shapes.items():
        if len(shp) == 1:
            out[name] = np.full(shp, bias_value, dtype=np.float32)
        elif len(shp) == 4:
            std = np.sqrt(2.0 / (shp[0] * shp[1] * shp[2] + shp[3]))
            out[name] = rng.normal(0.0, std, size=shp).astype(np.float32)
        elif len(shp) == 2:
            std = np.sqrt(2.0 / (shp[0] + shp[1]))
            out[name] = rng.normal(0.0, std, size=shp).astype(np.float32)
        else:
            raise ValueError(f"unexpected rank for {name}: {shp}")
    return out


def init_aru_weights(cfg: AruConfig, seed: int = 1234, bias_jitter: float = 0.0, logit_scale: float = 1.0):
    """`logit_scale` < 1 shrinks the class-logit filter so that random-weight probabilities do not saturate
    at 0/1 (used by the parity tests to keep the probability comparison sensitive)."""
    w = _init_from_shapes(aru_tensor_shapes(cfg), seed)
    if logit_scale != 1.0:
        w["aru_net/logit/class/weights"] = (w["aru_net/logit/class/weights"] * logit_scale).astype(np.float32)
    if bias_jitter:
        # optional: non-constant biases make parity tests sensitive to bias indexing errors
        rng = np.random.default_rng(seed + 1)
        for k in w:
            if w[k].ndim == 1:
                w[k] = (w[k] + rng.normal(0, bias_jitter, size=w[k].shape)).astype(np.float32)
    return w


def init_gnn_weights(cfg: GnnConfig, seed: int = 1234, bias_jitter: float = 0.0):
    w = _init_from_shapes(gnn_tensor_shapes(cfg), seed)
    if bias_jitter:
        rng = np.random.default_rng(seed + 1)
        for k in w:
            if w[k].ndim == 1:
                w[k] = (w[k] + rng.normal(0, bias_jitter, size=w[k].shape)).astype(np.float32)
    return w


# ----------------------------------------------------------------------------------------------
# blob (de)serialisation
# ----------------------------------------------------------------------------------------------
def pack_blob(tensors: Dict[str, np.ndarray]) -> bytes:
    parts = [MAGIC, struct.pack("<I", len(tensors))]
    for name, arr in tensors.items():
        a = np.ascontiguousarray(arr, dtype="<f4")
        nb = name.encode("utf-8")
        parts.append(struct.pack("<H", len(nb)))
        parts.append(nb)
        parts.append(struct.pack("<B", a.ndim))
        parts.append(struct.pack(f"<{a.ndim}I", *a.shape))
        parts.append(a.tobytes())
    return b"".join(parts)


def unpack_blob(blob: bytes) -> "OrderedDict[str, np.ndarray]":
    if blob[:8] != MAGIC:
        raise IOError("not an ASEPW001 weight blob")
    (n,) = struct.unpack_from("<I", blob, 8)
    off = 12
    out = OrderedDict()
    for _ in range(n):
        (nl,) = struct.unpack_from("<H", blob, off); off += 2
        name = blob[off:off + nl].decode("utf-8"); off += nl
        (nd,) = struct.unpack_from("<B", blob, off); off += 1
        dims = struct.unpack_from(f"<{nd}I", blob, off); off += 4 * nd
        cnt = int(np.prod(dims)) if nd else 1
        out[name] = np.frombuffer(blob, dtype="<f4", count=cnt, offset=off).reshape(dims).copy()
        off += 4 * cnt
    if off != len(blob):
        raise IOError("trailing bytes in weight blob")
    return out


def save_weights(path: str, tensors, meta: dict = None) -> None:
    """Write `<path>` as a blob; config/meta travels as a side-car json `<path>.json`."""
    import json
    with open(path, "wb") as f:
        f.write(pack_blob(tensors))
    if meta is not None:
        with open(path + ".json", "w") as f:
            json.dump(meta, f, indent=1)


def load_weights(path: str):
    import json, os
    with open(path, "rb") as f:
        tensors = unpack_blob(f.read())
    meta = None
    if os.path.exists(path + ".json"):
        with open(path + ".json") as f:
            meta = json.load(f)
    return tensors, meta
