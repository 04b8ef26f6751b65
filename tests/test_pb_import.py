"""TF-free frozen-graph (.pb) importer: synthesised GraphDefs (Const nodes under the reference's variable names,
both tensor_content and packed float_val encodings, a 'graph/' prefix, Identity '/read' nodes) must round-trip
into the engine's weight sets, and the hyper-parameters must be recovered from shapes / op list."""
import numpy as np
import pytest

from citlab_article_separation_new_amd import pb_import
from citlab_article_separation_new_amd.config import AruConfig, GnnConfig
from citlab_article_separation_new_amd.weights import init_aru_weights, init_gnn_weights


def _aru_pb(cfg, w, prefix="graph/"):
    extra = [{"name": f"{prefix}aru_net/attMapG/AvgPool_{i}", "op": "AvgPool"} for i in range(cfg.num_scales_att - 1)]
    extra.append({"name": "output", "op": "Softmax", "input": [prefix + "aru_net/logit/logits"]})
    return pb_import.weights_to_graphdef(w, prefix, extra if cfg.use_attention else extra[-1:])


@pytest.mark.parametrize("kw", [{}, {"graph": "RU"}, {"scale_space_num": 3, "res_depth": 2, "n_classes": 3, "num_scales_att": 2},
                                {"feat_root": 16}])
def test_aru_roundtrip(tmp_path, kw):
    cfg = AruConfig(**kw)
    w = init_aru_weights(cfg, 5, bias_jitter=0.03)
    path = tmp_path / "net.pb"
    path.write_bytes(_aru_pb(cfg, w))
    tensors, got = pb_import.aru_from_nodes(pb_import.read_graph(str(path)))
    assert list(tensors) == list(w)
    for k in w:
        assert np.array_equal(tensors[k], w[k]), k
    for f in ("channels", "n_classes", "feat_root", "scale_space_num", "res_depth", "apply_softmax", "filter_size"):
        assert getattr(got, f) == getattr(cfg, f), f
    assert got.use_attention == cfg.use_attention
    if cfg.use_attention:
        assert got.num_scales_att == cfg.num_scales_att


def test_gnn_roundtrip_and_load_graph(tmp_path):
    from citlab_article_separation_new_amd import gnn_io, net_post_processing_helper as helper
    cfg = GnnConfig()
    w = init_gnn_weights(cfg, 9, bias_jitter=0.03)
    p = tmp_path / "gnn.pb"
    p.write_bytes(pb_import.weights_to_graphdef(w, "graph/", meta={"num_transition_steps": 3}))
    g = gnn_io.load_graph(str(p))
    assert g.cfg.node_feature_dim == 7 and g.cfg.edge_feature_dim == 2 and g.cfg.classifier_hidden == [64, 32]
    assert all(np.array_equal(g.tensors[k], w[k]) for k in w)
    acfg = AruConfig()
    aw = init_aru_weights(acfg, 1)
    q = tmp_path / "sep.pb"
    q.write_bytes(_aru_pb(acfg, aw, prefix=""))                 # no graph prefix at all
    ag = helper.load_graph(str(q))
    assert ag.cfg.feat_root == 8 and ag.cfg.scale_space_num == 5 and ag.cfg.num_scales_att == 3
    assert np.array_equal(ag.tensors["aru_net/logit/class/weights"], aw["aru_net/logit/class/weights"])


def test_wire_format_details_and_errors(tmp_path):
    # splat constants (one float_val for the whole tensor), int tensors, unknown fields are skipped
    splat = pb_import._enc_varint((1 << 3) | 0) + pb_import._enc_varint(pb_import.DT_FLOAT)
    splat += pb_import._enc_field(2, pb_import._enc_field(2, pb_import._enc_varint(8) + pb_import._enc_varint(3)))
    splat += pb_import._enc_varint((5 << 3) | 5) + np.float32(0.1).tobytes()
    t = pb_import._parse_tensor(memoryview(splat))
    assert t.shape == (3,) and np.all(t == np.float32(0.1))
    nodes = pb_import.parse_graphdef(pb_import.encode_graphdef([
        {"name": "a", "op": "Const", "value": np.arange(6, dtype=np.int32).reshape(2, 3), "packed": True},
        {"name": "b", "op": "Identity", "input": ["a"]}]))
    assert nodes[0]["value"].tolist() == [[0, 1, 2], [3, 4, 5]] and nodes[1]["input"] == ["a"]
    with pytest.raises(IOError):
        pb_import.parse_graphdef(b"")
    with pytest.raises(IOError):
        pb_import.aru_from_nodes(nodes)
    cfg = AruConfig()
    w = init_aru_weights(cfg, 2)
    del w["aru_net/featMapG/unet_up_1/convR_1/biases"]
    with pytest.raises(IOError):
        pb_import.aru_from_nodes(pb_import.parse_graphdef(_aru_pb(cfg, w)))
