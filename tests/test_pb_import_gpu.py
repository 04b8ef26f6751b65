"""Frozen graphs as ANOTHER exporter would write them, through ``load_graph`` and ``get_net_output`` on the GPU:
a graph whose scope names share nothing with ARU_v1's, and one with inference batch normalisation behind every layer
(FusedBatchNorm nodes / the folded Mul-Add pair).  Both must match the oracle evaluated on the ORIGINAL weights (with the
batch norm applied unfolded, ``forward_torch(bn=...)``).  Graphs are serialised by google.protobuf."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
pytest.importorskip("google.protobuf")
sys.path.insert(0, os.path.dirname(__file__))
import tf_aru_graph  # noqa: E402
from test_pb_import_protobuf import _bn_params, _hashed  # noqa: E402


@pytest.mark.parametrize("variant", ["hashed_names", "bn_fused", "bn_mul_add", "ru_no_softmax"])
def test_foreign_frozen_graph_loads_and_matches_oracle(tmp_path, variant):
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from citlab_article_separation_new_amd.config import AruConfig
    from citlab_article_separation_new_amd.weights import init_aru_weights
    from oracle import aru_oracle
    cfg = AruConfig(graph="RU", apply_softmax=False) if variant == "ru_no_softmax" else AruConfig()
    w = init_aru_weights(cfg, 31, bias_jitter=0.05, logit_scale=0.05)
    bn = _bn_params(cfg, w, 4) if variant.startswith("bn") else None
    opts = {"hashed_names": dict(rename=_hashed), "bn_fused": dict(rename=_hashed, bn=bn, bn_style="fused"),
            "bn_mul_add": dict(rename=_hashed, bn=bn, bn_style="mul_add", add_op="AddV2"),
            "ru_no_softmax": dict(rename=_hashed, output_softmax=False, read_identities=False)}[variant]
    pb = tmp_path / "foreign_net.pb"
    pb.write_bytes(tf_aru_graph.build_aru_pb(w, cfg, **opts))
    graph = helper.load_graph(str(pb))
    assert graph.cfg.use_attention == cfg.use_attention and graph.cfg.apply_softmax == cfg.apply_softmax
    assert graph.cfg.scale_space_num == 5 and graph.cfg.res_depth == 3 and graph.cfg.feat_root == 8
    img = np.random.default_rng(8).random((203, 310), dtype=np.float32)
    out = helper.get_net_output(img, graph, "0")
    ref = aru_oracle.forward_torch(img, w, cfg, bn=bn)
    err = float(np.abs(out - ref).max())
    scale = max(1.0, float(np.abs(ref).max()))
    print(f"\n{variant}: max|d| = {err:.2e} (output range {scale:.2f})")
    assert err <= 1e-4 * scale
    if bn:                                                   # the batch norm really matters on this input
        assert float(np.abs(aru_oracle.forward_torch(img, w, cfg) - ref).max()) > 1e-3
    graph.close()


@pytest.mark.parametrize("kw", [{"activation_name": "elu"}, {"activation_name": "leaky"}, {"graph": "U", "activation_name": "leaky"},
                                {"graph": "U"}], ids=lambda kw: ",".join(f"{k}={v}" for k, v in kw.items()))
def test_foreign_frozen_graph_of_an_aru_variant_loads_and_matches_oracle(tmp_path, kw):
    """ARU_v1.py:43,70-75,228-233: elu / leaky activations and the non-residual 'U' graph, frozen with hashed scope names: recognised
    from the op graph, run by the engine's layer-by-layer path, compared with the oracle on the original weights."""
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from citlab_article_separation_new_amd.config import AruConfig
    from citlab_article_separation_new_amd.weights import init_aru_weights
    from oracle import aru_oracle
    cfg = AruConfig(**kw)
    w = init_aru_weights(cfg, 37, bias_jitter=0.05, logit_scale=0.05)
    pb = tmp_path / "variant_net.pb"
    pb.write_bytes(tf_aru_graph.build_aru_pb(w, cfg, rename=_hashed))
    graph = helper.load_graph(str(pb))
    assert graph.cfg.graph == cfg.graph and graph.cfg.activation_name == cfg.activation_name
    img = np.random.default_rng(9).random((131, 207), dtype=np.float32)
    out = helper.get_net_output(img, graph, "0")
    ref = aru_oracle.forward_torch(img, w, cfg)
    err = float(np.abs(out - ref).max())
    print(f"\n{kw}: max|d| = {err:.2e}")
    assert err <= 1e-4
    graph.close()


@pytest.mark.parametrize("kw,merge_concat", [(dict(use_attention=True, num_attention_heads=4), True),
                                             (dict(use_attention=True, num_attention_heads=2, multihead_attention_merge_type="average"), False)],
                         ids=["four_heads_concat", "two_heads_average"])
def test_relation_net_with_attention_read_from_the_graph_matches_the_oracle(tmp_path, kw, merge_concat):
    """message_fn_chunk.py:35-41,199-245: a TF1-layout export of an attention net loads without hints (heads, merge type, widths from
    the graph) and gives the oracle's probabilities"""
    import tf_gnn_graph
    import tf_graphdef_proto as tp
    from citlab_article_separation_new_amd import gnn_io, synth
    from citlab_article_separation_new_amd.config import GnnConfig
    from citlab_article_separation_new_amd.weights import init_gnn_weights
    from oracle import gnn_oracle
    src = GnnConfig(**kw)
    w = init_gnn_weights(src, 23, bias_jitter=0.05)
    pb = tmp_path / "rel_att.pb"
    pb.write_bytes(tf_gnn_graph.build(tp.build_messages(), w, 3, merge_concat=merge_concat).SerializeToString())
    graph = gnn_io.load_graph(str(pb))
    assert graph.cfg.use_attention and graph.cfg.num_attention_heads == src.num_attention_heads
    assert graph.cfg.multihead_attention_merge_type == src.multihead_attention_merge_type
    N = 50
    g = synth.synth_graph(4, N=N, n_pairs=200, node_dim=7)
    probs = gnn_io.gnn_forward(graph, N, g["interacting_nodes"], g["node_features"], g["edge_features"])
    ref = gnn_oracle.forward(N, g["interacting_nodes"], g["node_features"], g["edge_features"], None, w, src)
    assert float(np.abs(probs - ref).max()) <= 1e-5
    graph.close()


@pytest.mark.parametrize("output_type", ["add_final_hidden_and_input", "concat_final_hidden_and_input"])
def test_relation_net_output_type_read_from_the_graph_matches_the_oracle(tmp_path, output_type):
    """graph_gnn.py:23,158-166: output_type add / concat of a TF1-layout export load without a hint and give the oracle's
    probabilities"""
    import tf_gnn_graph
    import tf_graphdef_proto as tp
    from citlab_article_separation_new_amd import gnn_io, synth
    from citlab_article_separation_new_amd.config import GnnConfig
    from citlab_article_separation_new_amd.weights import init_gnn_weights
    from oracle import gnn_oracle
    src = GnnConfig(output_type=output_type)
    w = init_gnn_weights(src, 19, bias_jitter=0.05)
    pb = tmp_path / "rel_out.pb"
    pb.write_bytes(tf_gnn_graph.build(tp.build_messages(), w, 3).SerializeToString())
    graph = gnn_io.load_graph(str(pb))
    assert graph.cfg.output_type == output_type
    N = 40
    g = synth.synth_graph(3, N=N, n_pairs=150, node_dim=7)
    probs = gnn_io.gnn_forward(graph, N, g["interacting_nodes"], g["node_features"], g["edge_features"])
    ref = gnn_oracle.forward(N, g["interacting_nodes"], g["node_features"], g["edge_features"], None, w, src)
    assert float(np.abs(probs - ref).max()) <= 1e-5
    graph.close()


@pytest.mark.parametrize("steps,compress", [(2, 0), (4, 0), (2, 6)])
def test_relation_net_options_read_from_the_op_graph_match_the_oracle(tmp_path, steps, compress):
    """a GraphDef laid out like a TF1 export (tests/tf_gnn_graph.py, serialised by google.protobuf) with a
    transition-step count other than 3 / with the compress_input layer (graph_gnn.py:19-20,102-109) loads through
    gnn_io.load_graph without any hint and gives the oracle's probabilities (generic kernels: widths differ from 32 / 32 / 32 in the
    compressed case; the fused MFMA step in the others)."""
    import tf_gnn_graph
    import tf_graphdef_proto as tp
    from citlab_article_separation_new_amd import gnn_io, synth
    from citlab_article_separation_new_amd.config import GnnConfig
    from citlab_article_separation_new_amd.weights import init_gnn_weights
    from oracle import gnn_oracle
    src = (GnnConfig(num_transition_steps=steps) if not compress else
           GnnConfig(node_feature_dim=15, compress_node_feature_dim=compress, num_transition_steps=steps, hidden_dim=24,
                     interaction_dim=20, interaction_hidden=[28], classifier_hidden=[40, 12]))
    w = init_gnn_weights(src, 17, bias_jitter=0.05)
    pb = tmp_path / "rel.pb"
    pb.write_bytes(tf_gnn_graph.build(tp.build_messages(), w, steps).SerializeToString())
    graph = gnn_io.load_graph(str(pb))
    assert graph.cfg.num_transition_steps == steps and graph.cfg.compress_node_feature_dim == compress
    N = 40
    g = synth.synth_graph(2, N=N, n_pairs=150, node_dim=src.node_feature_dim)
    probs = gnn_io.gnn_forward(graph, N, g["interacting_nodes"], g["node_features"], g["edge_features"])
    ref = gnn_oracle.forward(N, g["interacting_nodes"], g["node_features"], g["edge_features"], None, w, src)
    assert float(np.abs(probs - ref).max()) <= 1e-5
    # the step count matters: the same weights with 3 steps give something else
    other = gnn_oracle.forward(N, g["interacting_nodes"], g["node_features"], g["edge_features"], None, w,
                               type(src)(**{**src.to_dict(), "num_transition_steps": 3})) if steps != 3 else None
    assert other is None or float(np.abs(other - ref).max()) > 1e-4
    graph.close()
