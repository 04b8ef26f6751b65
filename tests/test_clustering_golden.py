"""Host clustering code vs golden vectors captured from the IMPORTED reference
(tests/golden/make_clustering_golden.py).  Labels/classes are compared exactly ("article ids bit-identical")."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import clustering_cases as cc  # noqa: E402

with open(os.path.join(HERE, "golden", "clustering_golden.json")) as f:
    GOLD = json.load(f)


class Flags:
    def __init__(self, params):
        self.clustering_params = params


def _id(c):
    return f'{c["kind"]}-n{c["n"]}-{c["dtype"]}-{c.get("method", "")}'


@pytest.mark.parametrize("case", GOLD["textblock"], ids=_id)
def test_textblock_clustering_matches_reference(case):
    from citlab_article_separation_new_amd.clustering import TextblockClustering
    confs = cc.make_confs(case["kind"], case["n"], case["seed"], case["dtype"])
    assert cc.digest(confs) == case["sha256"], "input generator drifted; regenerate the golden file"
    tb = TextblockClustering(Flags({}))
    tb.set_confs(confs.copy())
    tb.calc(case["method"])
    assert [int(v) for v in tb.tb_labels] == case["tb_labels"]
    if case["n"] > 2:
        assert [[int(v) for v in c] for c in tb.tb_classes] == case["tb_classes"]
        assert tb.num_classes == case["num_classes"] and tb.num_noise == case["num_noise"]
    assert tb.get_info(case["method"]) == case["info"]
    want = case["rel_LLH"]
    if np.isfinite(want):
        assert tb.rel_LLH == pytest.approx(want, rel=1e-12, abs=1e-12)
    else:
        assert str(float(tb.rel_LLH)) == str(want)


@pytest.mark.parametrize("case", GOLD["dbscan"], ids=lambda c: f'{c["kind"]}-n{c["n"]}-{c["dtype"]}-mn{c["params"]["min_neighbors_for_cluster"]}')
def test_dbscan_relation_matches_reference(case):
    from citlab_article_separation_new_amd.clustering import DBScanRelation
    confs = cc.make_confs(case["kind"], case["n"], case["seed"], case["dtype"])
    assert cc.digest(confs) == case["sha256"]
    labels = DBScanRelation(**case["params"]).cluster_relations(case["n"], confs.copy())
    assert [int(v) for v in labels] == case["labels"]


def test_unknown_method_and_bad_shape():
    from citlab_article_separation_new_amd.clustering import TextblockClustering, DBScanRelation
    tb = TextblockClustering(Flags({}))
    tb.set_confs(np.full((3, 3), 0.7, np.float32))
    with pytest.raises(NotImplementedError):
        tb.calc("nope")
    with pytest.raises(AssertionError):
        DBScanRelation().cluster_relations(3, np.zeros(8))
    with pytest.raises(AssertionError):
        DBScanRelation(weight_handling="median")


def test_split_list_and_rescale_points():
    from citlab_article_separation_new_amd.host_util import split_list, rescale_points
    for c in GOLD["split_list"]:
        assert split_list(list(range(c["n_items"])), c["n"]) == c["result"]
    for c in GOLD["rescale_points"]:
        assert [list(p) for p in rescale_points([tuple(p) for p in c["points"]], c["scale"])] == c["result"]
