"""GNN oracle: hand-checkable edge-correction cases (misc.py:7-151) and structural properties."""
import numpy as np
import pytest

from oracle import gnn_oracle as G


def _cfg_w(seed=5):
    from citlab_article_separation_new_amd.config import GnnConfig
    from citlab_article_separation_new_amd.weights import init_gnn_weights
    cfg = GnnConfig()
    return cfg, init_gnn_weights(cfg, seed, bias_jitter=0.05)


def test_edge_correction_hand_case():
    # N=4; edges with a duplicate, a reversed duplicate and a self loop
    edges = np.array([[2, 1], [0, 3], [1, 2], [2, 2], [2, 1]], np.int32)
    feat = np.array([[10.], [20.], [30.], [40.], [50.]], np.float32)
    e, f = G.correct_edges(edges, feat, 4, undirected=True)
    # symmetrised list: idx0 (2,1) 1 (0,3) 2 (1,2) 3 (2,2) 4 (2,1) | 5 (1,2) 6 (3,0) 7 (2,1) 8 (2,2) 9 (1,2)
    # sorted codes without self loops: (0,3) (1,2) (2,1) (3,0); first occurrences: 1, 2, 0, 6 -> feats 20, 30, 10, 20
    assert e.tolist() == [[0, 3], [1, 2], [2, 1], [3, 0]]
    assert f[:, 0].tolist() == [20., 30., 10., 20.]
    e2, f2 = G.correct_edges(edges, feat, 4, undirected=False)
    assert e2.tolist() == [[0, 3], [1, 2], [2, 1]]
    assert f2[:, 0].tolist() == [20., 30., 10.]


def test_reference_toy_graph_is_already_canonical():
    # the 4-node / 8-edge toy graph of message_fn_chunk.py:456-483 is symmetric, sorted and loop free
    edges = np.array([[0, 1], [0, 2], [0, 3], [1, 0], [1, 2], [2, 0], [2, 1], [3, 0]], np.int32)
    e, _ = G.correct_edges(edges, None, 4, undirected=True)
    assert e.tolist() == edges.tolist()


def test_full_relations_order():
    r = G.build_full_relations(3)
    assert r.tolist() == [[0, 0], [0, 1], [0, 2], [1, 0], [1, 1], [1, 2], [2, 0], [2, 1], [2, 2]]


def test_forward_shapes_precision_and_isolated_nodes():
    cfg, w = _cfg_w()
    rng = np.random.default_rng(1)
    N = 9
    edges = rng.integers(0, 6, size=(14, 2)).astype(np.int32)        # nodes 6..8 isolated
    u = rng.random((N, 7), dtype=np.float32)
    ef = rng.random((14, 2), dtype=np.float32)
    p32, h32 = G.forward(N, edges, u, ef, None, w, cfg, return_hidden=True)
    p64, h64 = G.forward(N, edges, u, ef, None, w, cfg, dtype=np.float64, return_hidden=True)
    assert p32.shape == (N * N, 2)
    assert np.allclose(p32.sum(axis=1), 1.0, atol=1e-6)
    assert np.abs(p32 - p64).max() < 1e-5 and np.abs(h32 - h64).max() < 1e-5


def test_permutation_equivariance_of_the_oracle():
    cfg, w = _cfg_w()
    rng = np.random.default_rng(2)
    N = 12
    edges = rng.integers(0, N, size=(30, 2)).astype(np.int32)
    u = rng.random((N, 7), dtype=np.float32)
    ef = rng.random((30, 2), dtype=np.float32)
    p0 = G.forward(N, edges, u, ef, None, w, cfg, dtype=np.float64)[:, 1].reshape(N, N)
    perm = rng.permutation(N)
    inv = np.argsort(perm)
    p1 = G.forward(N, inv[edges], u[perm], ef, None, w, cfg, dtype=np.float64)[:, 1].reshape(N, N)
    assert np.abs(p1 - p0[np.ix_(perm, perm)]).max() < 1e-12


def test_output_types_on_a_hand_case():
    """graph_gnn.py:158-166: 'add' = h + x W (no bias), 'concat' = [h | x]; with zero classifier weights on the h part the
    probabilities depend on x alone and can be computed by hand"""
    from citlab_article_separation_new_amd.config import GnnConfig
    from citlab_article_separation_new_amd.weights import init_gnn_weights
    from oracle import gnn_oracle as G
    N = 3
    edges = np.array([[0, 1], [1, 2]], np.int32)
    u = np.array([[1.0, 0, 0, 0, 0, 0, 0], [0, 2.0, 0, 0, 0, 0, 0], [0, 0, 3.0, 0, 0, 0, 0]], np.float32)
    ef = np.zeros((2, 2), np.float32)
    cfg = GnnConfig(output_type="concat_final_hidden_and_input", classifier_hidden=[], num_classes=2)
    w = init_gnn_weights(cfg, 1)
    H, U = cfg.hidden_dim, 7
    wo = np.zeros((2 * (H + U), 2), np.float32)
    wo[H + 0, 1] = 1.0                                       # class-1 logit = x_a[0] (feature 0 of the first node of the pair)
    w["Classification/logits/fully_connected_logit_layer_out/weights"] = wo
    w["Classification/logits/fully_connected_logit_layer_out/bias"] = np.zeros(2, np.float32)
    p = G.forward(N, edges, u, ef, None, w, cfg).reshape(N, N, 2)
    assert np.allclose(p[0, :, 1], 1 / (1 + np.exp(-1.0)), atol=1e-6) and np.allclose(p[1:, :, 1], 0.5, atol=1e-6)
    cfg = GnnConfig(output_type="add_final_hidden_and_input", classifier_hidden=[], num_classes=2)
    w = init_gnn_weights(cfg, 1)
    assert w["GraphLSTM1/dense/weights"].shape == (U, H)
    p0, h = G.forward(N, edges, u, ef, None, w, cfg, return_hidden=True)
    w2 = dict(w); w2["GraphLSTM1/dense/weights"] = np.zeros((U, H), np.float32)
    p1 = G.forward(N, edges, u, ef, None, w2, GnnConfig(classifier_hidden=[], num_classes=2))
    p2 = G.forward(N, edges, u, ef, None, w2, cfg)
    assert np.array_equal(p1, p2) and np.abs(p0 - p1).max() > 1e-4      # a zero projection is the plain output; a random one is not


def test_attention_values_are_paired_like_the_reference_does():
    """message_fn_chunk.py:203-211,448-452: values of softmax(transpose(sparse [from, to])) are read in (to, from) order and
    multiplied with the interaction features in (from, to) order.  Hand case, directed edges 0->1, 0->2, 1->2 (sorted by from, to):
    the transposed tensor has rows to=1: {from 0}, to=2: {from 0, from 1}; its values in row-major order are
    [softmax over {a01}] = [1], then softmax over {a02, a12}.  So interaction 0 (0->1) is weighted 1, interaction 1 (0->2) gets
    softmax(a02, a12)[0] and interaction 2 (1->2) softmax(a02, a12)[1] -- here the orders coincide; with the edge 2->0 added the
    (to, from) order starts with it and every weight moves one place."""
    from oracle import gnn_oracle as G
    frm, to = np.array([0, 0, 1]), np.array([1, 2, 2])
    a = np.array([0.3, 1.0, 2.0])
    s = np.exp([1.0, 2.0]) / np.exp([1.0, 2.0]).sum()
    assert np.allclose(G._transposed_sparse_softmax_values(a, frm, to, 3), [1.0, s[0], s[1]])
    # edges sorted by (from, to): 0->1, 0->2, 1->2, 2->0 ; transposed order (to, from): (0<-2), (1<-0), (2<-0), (2<-1)
    frm, to = np.array([0, 0, 1, 2]), np.array([1, 2, 2, 0])
    a = np.array([0.3, 1.0, 2.0, -1.0])
    got = G._transposed_sparse_softmax_values(a, frm, to, 3)
    assert np.allclose(got, [1.0, 1.0, s[0], s[1]])          # interaction 0 (0->1) gets the value of (0<-2), interaction 1 that of (1<-0), ...


def test_single_head_attention_with_constant_scores_is_the_balanced_sum():
    """with a zero attention MLP every in-edge of a node gets 1 / indegree: on an UNDIRECTED graph (where the pairing of the values
    is the reverse edge, whose target's in-degree equals ... only for regular graphs) -- so use a regular graph: a ring"""
    from citlab_article_separation_new_amd.config import GnnConfig
    from citlab_article_separation_new_amd.weights import init_gnn_weights, attention_scope
    from oracle import gnn_oracle as G
    N = 8
    edges = np.array([[i, (i + 1) % N] for i in range(N)], np.int32)       # ring: every node has in-degree 2 after symmetrisation
    rng = np.random.default_rng(0)
    u, ef = rng.random((N, 7), dtype=np.float32), rng.random((N, 2), dtype=np.float32)
    plain = GnnConfig()
    att = GnnConfig(use_attention=True)
    w = init_gnn_weights(att, 2, bias_jitter=0.05)
    for k in w:
        if k.startswith(attention_scope(0)):
            w[k] = np.zeros_like(w[k])
    p_att = G.forward(N, edges, u, ef, None, w, att)
    p_plain = G.forward(N, edges, u, ef, None, {k: v for k, v in w.items() if "unnormalized_attention" not in k}, plain)
    assert np.abs(p_att - p_plain).max() < 1e-6


def test_max_aggregation_on_a_hand_case():
    """message_fn_chunk.py:16,57-62,398-417 aggregation_type='max' = tf.sparse.reduce_max over axis 0 of the sparse [from, to] tensor
    of attenuated features: per target the maximum over its STORED in-edges (implicit zeros do not take part: a negative maximum
    stays negative), 0 for a target without in-edges.  Directed graph 0->2, 1->2, 0->1; node 0 has no in-edge.
    With the LSTM gates reading x alone (both incorporate_* off) and a first step from h = 0, x is recoverable from the cell state."""
    from citlab_article_separation_new_amd.config import GnnConfig
    from citlab_article_separation_new_amd.weights import init_gnn_weights
    from oracle import gnn_oracle as G
    N = 3
    edges = np.array([[0, 2], [1, 2], [0, 1]], np.int32)
    rng = np.random.default_rng(0)
    u = rng.normal(size=(N, 7)).astype(np.float32)
    ef = rng.normal(size=(3, 2)).astype(np.float32)
    base = dict(undirected_graph=False, num_transition_steps=1, incorporate_hidden_features_in_update=False,
                incorporate_node_input_features_in_update=False)
    cfg_max, cfg_sum = GnnConfig(aggregation_type="max", **base), GnnConfig(**base)
    w = init_gnn_weights(cfg_sum, 4, bias_jitter=0.3)
    assert w["GraphLSTM1/update_function_LSTM/ingate_activation/dense/weights"].shape == (32, 32)       # x alone
    # the messages by hand: m_e = tanh(MLP(z_e)), attenuated by 1 / indegree(target)
    ce, cf = G.correct_edges(edges, ef, N, False)
    assert ce.tolist() == [[0, 1], [0, 2], [1, 2]]
    frm, to = ce[:, 0], ce[:, 1]
    h0 = np.zeros((N, 32), np.float32)
    du = u[to] - u[frm]
    z = np.concatenate([u[frm], u[to], du, du * du, cf, h0[frm], h0[to], h0[frm], h0[frm]], axis=1)
    hid = np.maximum(z @ w[G.MSG + "/fully_connected_layer_h1/weights"] + w[G.MSG + "/fully_connected_layer_h1/bias"], 0)
    m = np.tanh(hid @ w[G.MSG + "/fully_connected_logit_layer_out/weights"] + w[G.MSG + "/fully_connected_logit_layer_out/bias"])
    x_max = np.stack([np.zeros(32), m[0] / 1.0, np.maximum(m[1], m[2]) / 2.0]).astype(np.float32)
    x_sum = np.stack([np.zeros(32), m[0] / 1.0, (m[1] + m[2]) / 2.0]).astype(np.float32)
    assert (np.maximum(m[1], m[2]) < 0).any(), "the case must contain a negative maximum (stored entries only)"

    def hidden_from_x(x):
        gate = {g: x @ w[f"{G.UPD}/{g}_activation/dense/weights"] + w[f"{G.UPD}/{g}_activation/dense/bias"]
                for g in ("ingate", "outgate", "cellinput")}
        sig = lambda v: 1 / (1 + np.exp(-v))
        return sig(gate["outgate"]) * np.tanh(sig(gate["ingate"]) * np.tanh(gate["cellinput"]))
    _, h_max = G.forward(N, edges, u, ef, None, w, cfg_max, return_hidden=True)
    _, h_sum = G.forward(N, edges, u, ef, None, w, cfg_sum, return_hidden=True)
    assert np.allclose(h_max, hidden_from_x(x_max), atol=1e-6) and np.allclose(h_sum, hidden_from_x(x_sum), atol=1e-6)
    assert np.abs(h_max[2] - h_sum[2]).max() > 1e-4 and np.allclose(h_max[:2], h_sum[:2])     # one in-edge: max = sum; none: 0
    with pytest.raises(ValueError):
        G.forward(N, edges, u, ef, None, w, GnnConfig(aggregation_type="mean", **base))


def test_lstm_input_flags_change_the_gate_width():
    """update_fn_lstm.py:13-16,43-50: v = [x] + [h] + [u]"""
    from citlab_article_separation_new_amd.config import GnnConfig
    from citlab_article_separation_new_amd.weights import gnn_tensor_shapes
    k = "GraphLSTM1/update_function_LSTM/forgetgate_activation/dense/weights"
    assert gnn_tensor_shapes(GnnConfig())[k] == (32 + 32 + 7, 32)
    assert gnn_tensor_shapes(GnnConfig(incorporate_hidden_features_in_update=False))[k] == (32 + 7, 32)
    assert gnn_tensor_shapes(GnnConfig(incorporate_node_input_features_in_update=False))[k] == (64, 32)
    assert gnn_tensor_shapes(GnnConfig(incorporate_hidden_features_in_update=False, incorporate_node_input_features_in_update=False))[k] == (32, 32)
