"""CPU ORACLE (test infrastructure, NOT product code) -- GNN relation predictor forward pass.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.

PARITY UNPINNED for the float path: the reference graph needs TensorFlow 1.x and the frozen
``mixed_gnn_vn7e*.pb`` (absent); the reference has no tests for it.  This is a numpy restatement of

    article_separation/gnn/model/graph_util/misc.py:7-151     check_and_correct_interacting_nodes
    article_separation/gnn/model/graph/graph_gnn.py:46-167     GraphGNN.infer (batch size 1)
    article_separation/gnn/model/graph/message_fn_chunk.py:148-418  default message function
    article_separation/gnn/model/graph/update_fn_lstm.py:31-101     LSTM update
    article_separation/gnn/model/graph/graph_relation.py:229-287    pair classifier
    article_separation/gnn/model/model_relation.py:326-328          softmax output node
    article_separation/gnn/input/input_dataset.py:444-457          build_full_relations

The integer part (edge correction) is pinned by hand-checkable cases in tests/test_oracle_gnn.py
(the toy inputs of misc.py:641-704 carry no expected outputs in the reference).
"""
import numpy as np

MSG = ("GraphLSTM1/message_fn_default/head_0/calculation_interaction_features/"
       "concat_u_and_h/interaction_features")
UPD = "GraphLSTM1/update_function_LSTM"
CLS = "Classification/logits"


def correct_edges(edges, edge_feat, num_nodes, undirected=True):
    """misc.py:7-151 for one sample.

    symmetrise (append reversed edges, tile features x2, :47-58) -> encode from*N+to (:60) ->
    unique (:64) -> remove self loops with tf.sets.difference, whose result is sorted ascending
    (:68-72) -> for every surviving code the FIRST index in the (symmetrised) list (:76-88) ->
    gather edge features with those indices (:104) -> decode (:91)."""
    edges = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    E = edges.shape[0]
    if edge_feat is not None:
        edge_feat = np.asarray(edge_feat, dtype=np.float32)
        edge_feat = edge_feat.reshape(E, edge_feat.shape[-1] if edge_feat.ndim > 1 else 1)
    if undirected:
        full = np.concatenate([edges, edges[:, ::-1]], axis=0)
        feat_full = np.tile(edge_feat, (2, 1)) if edge_feat is not None else None
    else:
        full, feat_full = edges, edge_feat
    codes_full = full[:, 0] * num_nodes + full[:, 1]
    codes, first = np.unique(codes_full, return_index=True)          # sorted + first occurrence
    keep = (codes // num_nodes) != (codes % num_nodes)
    codes, first = codes[keep], first[keep]
    out = np.stack([codes // num_nodes, codes % num_nodes], axis=1).astype(np.int32)
    out_feat = feat_full[first] if feat_full is not None else None
    return out, out_feat


def build_full_relations(num_nodes):
    """input_dataset.py:444-457: all N*N ordered pairs, row major."""
    a = np.repeat(np.arange(num_nodes, dtype=np.int32), num_nodes)
    b = np.tile(np.arange(num_nodes, dtype=np.int32), num_nodes)
    return np.stack([a, b], axis=1)


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _softmax(x):
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=-1, keepdims=True)


def _transposed_sparse_softmax_values(a_un, frm, to, N):
    """message_fn_chunk.py:203-211,448-452 restated literally: the unnormalised values sit in a sparse [from, to] tensor (indices =
    the interactions, which are sorted by from * N + to); it is TRANSPOSED, soft-maxed over its last axis (for every `to`: over its
    in-edges) and ``.values`` is taken -- the values of the transposed tensor in ITS row-major order, i.e. sorted by (to, from).
    The caller multiplies them element by element with the interaction features, which are in (from, to) order: interaction e gets
    the e-th value of the (to, from)-sorted list (for an undirected graph: the attention of the reverse edge, normalised over the
    in-edges of `from`)."""
    order = np.lexsort((frm, to))                                    # positions of the transposed tensor: (to, from) ascending
    vals = np.asarray(a_un, np.float64)[order]
    rows = np.asarray(to)[order]
    out = np.empty_like(vals)
    start = 0
    while start < len(vals):                                         # tf.sparse.softmax: per row over the stored entries
        stop = start
        while stop < len(vals) and rows[stop] == rows[start]:
            stop += 1
        e = np.exp(vals[start:stop] - vals[start:stop].max())
        out[start:stop] = e / e.sum()
        start = stop
    return out                                                       # .values: NOT permuted back


def forward(num_nodes, edges, node_feat, edge_feat, relations, w, cfg, dtype=np.float32,
            return_hidden=False):
    """== sess.run('output_belong_to_same_instance:0') at batch size 1 -> probs [R, num_classes]."""
    N = int(num_nodes)
    w = {k: v.astype(dtype) for k, v in w.items()}
    u = np.asarray(node_feat, dtype=dtype).reshape(N, -1)
    fed = u                                                          # inputs['node_features'] (graph_gnn.py:158-166 reads these)
    if getattr(cfg, "compress_node_feature_dim", 0) > 0:            # graph_gnn.py:102-109: ff_layer(tanh) on the fed features
        u = np.tanh(u @ w["GraphLSTM1/compress_input/ff_compress_input/weights"]
                    + w["GraphLSTM1/compress_input/ff_compress_input/bias"]).astype(dtype)
    ce, cf = correct_edges(edges, edge_feat, N, cfg.undirected_graph)
    cf = cf.astype(dtype) if cf is not None else np.zeros((ce.shape[0], 0), dtype)
    frm, to = ce[:, 0], ce[:, 1]
    H = cfg.hidden_dim
    h = np.zeros((N, H), dtype)
    c = np.zeros((N, H), dtype)
    deg = np.bincount(to, minlength=N).astype(dtype)                 # in-degree (message_fn_chunk.py:369-386)
    for _ in range(cfg.num_transition_steps):
        du = u[to] - u[frm]
        dh = h[to] - h[frm]
        # order fixed by message_fn_chunk.py:313-350: u_from,u_to,u_diff,u_sq | edge | h_from,h_to,h_diff,h_sq
        z = np.concatenate([u[frm], u[to], du, du * du, cf, h[frm], h[to], dh, dh * dh], axis=1)
        use_att = bool(getattr(cfg, "use_attention", False))
        heads = int(getattr(cfg, "num_attention_heads", 1)) if use_att else 1
        per_head = []
        for k in range(heads):                                       # message_fn_chunk.py:172-224
            msg = MSG.replace("head_0", f"head_{k}")
            hid = z
            for i in range(1, len(cfg.interaction_hidden) + 1):
                hid = np.maximum(hid @ w[f"{msg}/fully_connected_layer_h{i}/weights"]
                                 + w[f"{msg}/fully_connected_layer_h{i}/bias"], 0)
            m = np.tanh(hid @ w[f"{msg}/fully_connected_logit_layer_out/weights"]
                        + w[f"{msg}/fully_connected_logit_layer_out/bias"])
            if use_att:
                att = (f"GraphLSTM1/message_fn_default/head_{k}/calculation_unnormalized_attention_values/"
                       "calculation_interaction_features/concat_u_and_h/interaction_features")
                hid = z
                for i in range(1, len(cfg.attention_hidden) + 1):
                    hid = np.maximum(hid @ w[f"{att}/fully_connected_layer_h{i}/weights"]
                                     + w[f"{att}/fully_connected_layer_h{i}/bias"], 0)
                a_un = (hid @ w[f"{att}/fully_connected_logit_layer_out/weights"]
                        + w[f"{att}/fully_connected_logit_layer_out/bias"])[:, 0]                 # :446 squeeze
                # :76-110: the message function runs once per CHUNK of 100000 // N target nodes (all of them for N <= 316) on the
                # interactions that end in the chunk, and the per-chunk results are added; the soft-max rows (in-edges of a target)
                # are complete inside a chunk, but the value pairing below happens among the chunk's interactions only
                weights_e = np.zeros(len(to), np.float64)
                chunk_nodes = max(1, 100000 // N)
                for c0 in range(0, N, chunk_nodes):
                    sel = np.nonzero((to >= c0) & (to < c0 + chunk_nodes))[0]           # keeps the (from, to) order
                    if len(sel):
                        weights_e[sel] = _transposed_sparse_softmax_values(a_un[sel], frm[sel], to[sel], N)    # :203-211
                weights_e = weights_e.astype(dtype)
            else:
                weights_e = (1.0 / deg[to]).astype(dtype) if len(to) else np.zeros(0, dtype)      # :369-386
            xk = np.zeros((N, m.shape[1]), dtype)
            att_feat = m * weights_e[:, None] if len(to) else m       # :214 attenuated interaction features
            agg = getattr(cfg, "aggregation_type", "sum")
            if len(to) and agg == "sum":
                np.add.at(xk, to, att_feat)                           # :398-417: sparse [from, to], tf.sparse.reduce_sum over axis 0
            elif len(to) and agg == "max":
                # :57-62 tf.sparse.reduce_max over axis 0: the maximum over the STORED entries of a column (the implicit zeros do not
                # take part, so a target's value may be negative); a column without entries reduces to 0
                filled = np.full_like(xk, -np.inf)
                np.maximum.at(filled, to, att_feat)
                has = np.bincount(to, minlength=N) > 0
                xk[has] = filled[has]
            elif len(to):
                raise ValueError(f"aggregation_type {agg!r}")
            per_head.append(xk)
        if not use_att or getattr(cfg, "multihead_attention_merge_type", "concat") == "average":
            x = (sum(per_head) / dtype(heads)).astype(dtype)          # :229-233
        else:
            x = np.concatenate(per_head, axis=1)                      # :234-237
        parts = [x]                                                   # update_fn_lstm.py:41-50
        if getattr(cfg, "incorporate_hidden_features_in_update", True):
            parts.append(h)
        if getattr(cfg, "incorporate_node_input_features_in_update", True):
            parts.append(u)
        v = np.concatenate(parts, axis=1)
        gate = {g: v @ w[f"{UPD}/{g}_activation/dense/weights"] + w[f"{UPD}/{g}_activation/dense/bias"]
                for g in ("ingate", "outgate", "forgetgate", "cellinput")}
        i_g, o_g, f_g = _sigmoid(gate["ingate"]), _sigmoid(gate["outgate"]), _sigmoid(gate["forgetgate"])
        g_g = np.tanh(gate["cellinput"])
        c = f_g * c + i_g * g_g                                      # update_fn_lstm.py:74-76
        h = o_g * np.tanh(c)
    out_type = getattr(cfg, "output_type", "hidden")
    hc = h
    if out_type == "add_final_hidden_and_input":                     # graph_gnn.py:160-163: out += ff_layer(x, no bias, no activation)
        hc = (h + fed @ w["GraphLSTM1/dense/weights"]).astype(dtype)
    elif out_type == "concat_final_hidden_and_input":                # graph_gnn.py:164-166
        hc = np.concatenate([h, fed], axis=1)
    elif out_type != "hidden":
        raise ValueError(f"output_type {out_type!r}")
    rel = build_full_relations(N) if relations is None else np.asarray(relations, dtype=np.int64).reshape(-1, 2)
    feat = np.concatenate([hc[rel[:, 0]], hc[rel[:, 1]]], axis=1)
    for i in range(1, len(cfg.classifier_hidden) + 1):
        feat = np.maximum(feat @ w[f"{CLS}/fully_connected_layer_h{i}/weights"]
                          + w[f"{CLS}/fully_connected_layer_h{i}/bias"], 0)
    logits = feat @ w[f"{CLS}/fully_connected_logit_layer_out/weights"] + w[f"{CLS}/fully_connected_logit_layer_out/bias"]
    probs = _softmax(logits).astype(dtype)
    if return_hidden:
        return probs, h
    return probs


def visual_node_features(image, regions, num_points, w, cfg, return_maps=False, scope_kind="node"):
    """graph_relation.py:84-127 + misc.py:249-381 at batch size 1 -> [N, sum(layer_compressed_dim)] float32.

    image [h,w] float32 as fed (0..255) -> (normalize_image when cfg.mvn: per-image standardisation over the true
    shape, misc.py:272-279; folded into the backbone config) -> ARU_v1 backbone end points -> per node the paraxial
    rectangle of its region (relative coordinates, misc.py:486-508; no points -> zeros) -> floor-scaled ROI clamped
    into the map, at least one cell (misc.py:322-341) -> per-channel max (misc.py:345-359) -> ff + ReLU
    (misc.py:365-368)."""
    from oracle import aru_oracle
    bcfg = cfg.backbone_cfg()
    wb = {k: v for k, v in w.items() if k.startswith("aru_net/")}
    img = np.asarray(image, dtype=np.float32)
    if img.ndim == 3:
        img = img[:, :, 0]
    _, inter = aru_oracle.forward_torch(img, wb, bcfg, return_intermediates=True)
    regions = np.asarray(regions, dtype=np.float32)
    N = regions.shape[0]
    feats, maps = [], []
    for i, name in enumerate(cfg.visual_layers):
        fm = inter[name]                                   # [fh, fw, C]
        fh, fw, _ = fm.shape
        vmax = np.empty((N, fm.shape[2]), dtype=np.float32)
        for n in range(N):
            k = int(num_points[n])
            if k == 0:
                xmin = xmax = ymin = ymax = np.float32(0)
            else:
                xmin, xmax = regions[n, 0, :k].min(), regions[n, 0, :k].max()
                ymin, ymax = regions[n, 1, :k].min(), regions[n, 1, :k].max()
            x0 = max(min(int(np.floor(np.float32(xmin) * np.float32(fw))), fw - 1), 0)
            x1 = max(min(int(np.floor(np.float32(xmax) * np.float32(fw))), fw - 1), 0)
            y0 = max(min(int(np.floor(np.float32(ymin) * np.float32(fh))), fh - 1), 0)
            y1 = max(min(int(np.floor(np.float32(ymax) * np.float32(fh))), fh - 1), 0)
            nx, ny = max(x1 - x0 + 1, 1), max(y1 - y0 + 1, 1)
            vmax[n] = fm[y0:y0 + ny, x0:x0 + nx].max(axis=(0, 1))
        scope = f"visual_{scope_kind}_feature_compression_fm_{i}/dense"        # misc.py:365 / :467
        feats.append(np.maximum(vmax @ w[scope + "/weights"].astype(np.float32)
                                + w[scope + "/bias"].astype(np.float32), 0).astype(np.float32))
        maps.append(vmax)
    out = np.concatenate(feats, axis=1)
    return (out, maps) if return_maps else out


def forward_visual(num_nodes, edges, node_feat, edge_feat, image, regions, num_points, relations, w, cfg,
                   edge_regions=None, edge_num_points=None):
    """== sess.run('output_belong_to_same_instance:0') of a graph exported with image_input, batch size 1.
    cfg.visual_edges (graph_relation.py:141-172, misc.py:384-470): the same ROI max on the interactions' regions, compressed by the
    visual_edge_feature_compression_fm_<i> layers, concatenated behind the fed edge features BEFORE the GNN (whose edge correction
    then keeps the first occurrence's features)."""
    N = int(num_nodes)
    vis = visual_node_features(image, regions, num_points, w, cfg)
    geo = np.asarray(node_feat, dtype=np.float32).reshape(N, -1) if node_feat is not None else np.zeros((N, 0), np.float32)
    u = np.concatenate([geo, vis], axis=1)
    if getattr(cfg, "visual_edges", False):
        E = np.asarray(edges).reshape(-1, 2).shape[0]
        evis = visual_node_features(image, edge_regions, edge_num_points, w, cfg, scope_kind="edge")
        egeo = np.asarray(edge_feat, dtype=np.float32).reshape(E, -1) if edge_feat is not None else np.zeros((E, 0), np.float32)
        edge_feat = np.concatenate([egeo, evis], axis=1)
    probs = forward(N, edges, u, edge_feat, relations, w, cfg)
    return probs, u
