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
