"""TEST INFRASTRUCTURE: the relation net's op graph the way TensorFlow 1.x freezes it -- as far as an importer can tell the
exporter's options apart (``/root/reference/article_separation/gnn/model/graph/graph_gnn.py:19-23,102-109,134-166``,
``message_fn_chunk.py:35-41,167-176,356-363,420-422``):

  * every variable is a Const + ``/read`` Identity, created ONCE (the transition steps share weights, AUTO_REUSE);
  * the transition steps are unrolled in Python: step t has its own ops; the edge MLP runs inside the chunking ``tf.while_loop``
    of MessageFnChunk, so its MatMul reads the weights through an ``Enter`` op: ONE MatMul on ``fully_connected_layer_h1/weights``
    per step -- the only place ``num_transition_steps`` is visible in a frozen graph;
  * ``compress_node_feature_dim`` adds ``GraphLSTM1/compress_input/ff_compress_input/{weights,bias}`` + MatMul / BiasAdd / Tanh;
  * ``use_attention`` adds ``.../head_k/calculation_unnormalized_attention_values/...`` variables, ``num_attention_heads`` > 1 the
    scopes ``head_1``, ...; ``output_type='add_final_hidden_and_input'`` an un-named ``GraphLSTM1/dense/weights`` (no bias),
    ``'concat_final_hidden_and_input'`` a pair classifier that is 2 x u wider.
The graph is serialised by google.protobuf (tests/tf_graphdef_proto.py), not by the product's encoder; only structure the importer
reads is modelled (data-flow ops between the MatMuls are abbreviated)."""
import numpy as np

import tf_graphdef_proto as tp

F32 = tp.DType(tp.DT_FLOAT)
MSG = "GraphLSTM1/message_fn_default/head_0/calculation_interaction_features/concat_u_and_h/interaction_features"
UPD = "GraphLSTM1/update_function_LSTM"
CLS = "Classification/logits"


def _mlp_layers(weights, scope):
    """layer names of an MLP of layers.py:468-490 as far as ``weights`` holds them: hidden layers h1, h2, ... + the output layer"""
    out = []
    while f"{scope}/fully_connected_layer_h{len(out) + 1}/weights" in weights:
        out.append(f"fully_connected_layer_h{len(out) + 1}")
    return out + ["fully_connected_logit_layer_out"]


def build(ns, weights, num_transition_steps, prefix="graph/", extra_consts=None, merge_concat=None, aggregation="sum",
          lstm_inputs=(True, True), node_feature_dim=None):
    """aggregation: 'sum' / 'max' -> the SparseReduceSum / SparseReduceMax op of message_fn_chunk.py:398-417 (inside its map_fn loop);
    lstm_inputs = (incorporate_hidden_features_in_update, incorporate_node_input_features_in_update): which tensors the gates' ConcatV2
    joins behind x (update_fn_lstm.py:41-50); node_feature_dim: static last dimension of the node_features placeholder"""
    nodes, made = [], set()

    def add(name, op, inputs=(), **attrs):
        nodes.append(tp.node(ns, name, op, inputs, **attrs))
        return name

    def var(name):
        full = prefix + name
        if full not in made:
            made.add(full)
            arr = np.asarray(weights[name], np.float32)
            nodes.append(tp.node(ns, full, "Const", (), dtype=F32, value=arr))
            add(full + "/read", "Identity", [full], T=F32)
        return full + "/read"

    for ph in ("num_nodes", "interacting_nodes", "node_features", "edge_features", "relations_to_consider_belong_to_same_instance"):
        if ph == "node_features" and node_feature_dim is not None:
            add(ph, "Placeholder", dtype=F32, shape=tp.Shape(-1, -1, node_feature_dim))
        else:
            add(ph, "Placeholder", dtype=F32)
    u = "node_features"
    if "GraphLSTM1/compress_input/ff_compress_input/weights" in weights:
        s = prefix + "GraphLSTM1/compress_input/ff_compress_input"
        u = add(s + "/Tanh", "Tanh", [add(s + "/BiasAdd", "BiasAdd", [add(s + "/MatMul", "MatMul", [u, var("GraphLSTM1/compress_input/ff_compress_input/weights")], T=F32),
                                                                  var("GraphLSTM1/compress_input/ff_compress_input/bias")], T=F32)], T=F32)
    h = add(prefix + "GraphLSTM1/zeros", "Fill", ["num_nodes"], T=F32)
    for t in range(num_transition_steps):
        sfx = "" if t == 0 else f"_{t}"
        loop = prefix + f"GraphLSTM1/message_fn_default{sfx}/head_0/while"
        z = add(loop + "/concat", "ConcatV2", [u, h, "edge_features"], T=F32, N=3)
        x = z
        for layer in _mlp_layers(weights, MSG):
            enter_w = add(f"{loop}/{layer}/MatMul/Enter", "Enter", [var(f"{MSG}/{layer}/weights")], T=F32, frame_name=loop, is_constant=True)
            enter_b = add(f"{loop}/{layer}/BiasAdd/Enter", "Enter", [var(f"{MSG}/{layer}/bias")], T=F32, frame_name=loop, is_constant=True)
            x = add(f"{loop}/{layer}/BiasAdd", "BiasAdd", [add(f"{loop}/{layer}/MatMul", "MatMul", [x, enter_w], T=F32), enter_b], T=F32)
        # further attention heads / the attention MLPs (message_fn_chunk.py:172-224): their variables, one MatMul each
        head_out = [x]
        k = 0
        while True:
            msg_k = MSG.replace("head_0", f"head_{k}")
            att_k = f"GraphLSTM1/message_fn_default/head_{k}/calculation_unnormalized_attention_values/calculation_interaction_features/concat_u_and_h/interaction_features"
            if k > 0:
                if f"{msg_k}/fully_connected_layer_h1/weights" not in weights:
                    break
                y = z
                for layer in _mlp_layers(weights, msg_k):
                    y = add(f"{loop}/head_{k}/{layer}/BiasAdd", "BiasAdd", [add(f"{loop}/head_{k}/{layer}/MatMul", "MatMul", [y, var(f"{msg_k}/{layer}/weights")], T=F32),
                                                                               var(f"{msg_k}/{layer}/bias")], T=F32)
                head_out.append(y)
            if f"{att_k}/fully_connected_layer_h1/weights" in weights:
                y = z
                for layer in _mlp_layers(weights, att_k):
                    y = add(f"{loop}/head_{k}/att/{layer}/BiasAdd", "BiasAdd", [add(f"{loop}/head_{k}/att/{layer}/MatMul", "MatMul", [y, var(f"{att_k}/{layer}/weights")], T=F32),
                                                                                   var(f"{att_k}/{layer}/bias")], T=F32)
                head_out[-1] = add(f"{loop}/head_{k}/mul", "Mul", [head_out[-1], y], T=F32)
            k += 1
        # message_fn_chunk.py:398-417: per feature component a sparse [from, to] tensor reduced over axis 0 (inside a map_fn loop); the
        # degree count of the balanced weighting is a SparseReduceSum of its own (:369-386)
        add(f"{loop}/SparseReduceSum", "SparseReduceSum", ["interacting_nodes"], T=F32)
        red = "SparseReduceMax" if aggregation == "max" else "SparseReduceSum"
        head_out = [add(f"{loop}/head_{i}/map/while/{red}", red, [y], T=F32) for i, y in enumerate(head_out)]
        if len(head_out) > 1:
            concat = merge_concat if merge_concat is not None else True
            if concat:
                x = add(prefix + f"GraphLSTM1/message_fn_default{sfx}/concat", "ConcatV2", head_out, T=F32, N=len(head_out))
            else:                                             # message_fn_chunk.py:229-233
                x = add(prefix + f"GraphLSTM1/message_fn_default{sfx}/truediv", "RealDiv",
                        [add(prefix + f"GraphLSTM1/message_fn_default{sfx}/AddN", "AddN", head_out, T=F32, N=len(head_out))], T=F32)
        else:
            x = head_out[0]
        gates = []
        joined = [x] + ([h] if lstm_inputs[0] else []) + ([u] if lstm_inputs[1] else [])
        for g in ("ingate", "outgate", "forgetgate", "cellinput"):
            s = prefix + f"{UPD}{sfx}/{g}_activation/dense"
            v = add(prefix + f"{UPD}{sfx}/{g}_activation/concat", "ConcatV2", joined, T=F32, N=len(joined))     # update_fn_lstm.py:93
            gates.append(add(s + "/BiasAdd", "BiasAdd", [add(s + "/MatMul", "MatMul", [v, var(f"{UPD}/{g}_activation/dense/weights")], T=F32),
                                                          var(f"{UPD}/{g}_activation/dense/bias")], T=F32))
        h = add(prefix + f"{UPD}{sfx}/mul_2", "Mul", gates[:2], T=F32)
    x = h
    if "GraphLSTM1/dense/weights" in weights:                 # graph_gnn.py:160-163 output_type='add_final_hidden_and_input'
        s = prefix + "GraphLSTM1/dense"
        x = add(prefix + "GraphLSTM1/add", "Add", [h, add(s + "/Wx", "MatMul", ["node_features", var("GraphLSTM1/dense/weights")], T=F32)], T=F32)
    elif weights[f"{CLS}/fully_connected_layer_h1/weights"].shape[0] != 2 * weights[f"{UPD}/ingate_activation/dense/weights"].shape[1]:
        x = add(prefix + "GraphLSTM1/concat", "ConcatV2", [h, "node_features"], T=F32, N=2)     # graph_gnn.py:164-166
    i = 1
    while f"{CLS}/fully_connected_layer_h{i}/weights" in weights:
        s = prefix + f"{CLS}/fully_connected_layer_h{i}"
        x = add(s + "/Relu", "Relu", [add(s + "/BiasAdd", "BiasAdd", [add(s + "/MatMul", "MatMul", [x, var(f"{CLS}/fully_connected_layer_h{i}/weights")], T=F32),
                                                                      var(f"{CLS}/fully_connected_layer_h{i}/bias")], T=F32)], T=F32)
        i += 1
    s = prefix + f"{CLS}/fully_connected_logit_layer_out"
    x = add(s + "/BiasAdd", "BiasAdd", [add(s + "/MatMul", "MatMul", [x, var(f"{CLS}/fully_connected_logit_layer_out/weights")], T=F32),
                                        var(f"{CLS}/fully_connected_logit_layer_out/bias")], T=F32)
    add("output_belong_to_same_instance", "Softmax", [x], T=F32)
    for name, arr in (extra_consts or {}).items():             # variables of options the engine does not serve
        nodes.append(tp.node(ns, prefix + name, "Const", (), dtype=F32, value=np.asarray(arr, np.float32)))
        add(prefix + name + "/read", "Identity", [prefix + name], T=F32)
    return tp.graphdef(ns, nodes)
