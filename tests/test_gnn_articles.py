"""Planted-article cases on the CPU: the oracle's confidences clustered by the PRODUCT's clustering code must give the
labels the REFERENCE's clustering code gave (tests/golden/gnn_articles_golden.json, written by
tests/golden/make_gnn_articles_golden.py), for every method and both confidence dtypes of the reference CLI -- and the
cases must not be degenerate (several articles, singletons, confidences on both sides of 0.5)."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import gnn_article_cases as gac  # noqa: E402

GOLDEN = {c["name"]: c for c in json.load(open(os.path.join(HERE, "golden", "gnn_articles_golden.json")))["cases"]}


class Flags:
    clustering_params = {}


def cluster(conf, method):
    from citlab_article_separation_new_amd.clustering import TextblockClustering
    tb = TextblockClustering(Flags())
    tb.set_confs(conf.copy())
    tb.calc(method)
    return [int(v) for v in tb.tb_labels]


@pytest.mark.parametrize("case", gac.CASES, ids=lambda c: c["name"])
def test_oracle_confidences_give_the_reference_labels(case):
    g, w, cfg, probs = gac.build(case)
    gold = GOLDEN[case["name"]]
    assert g["planted"].tolist() == gold["planted"]
    conf = probs[:, 1]
    assert (conf > 0.5).any() and (conf < 0.5).any()
    assert abs(float(np.abs(conf - 0.5).min()) - gold["min_abs_conf_minus_half"]) < 1e-6
    for vname, cv in gac.conf_variants(case, probs).items():
        assert cv.dtype == (np.float32 if vname == "float32" else np.float64)
        for method in gac.METHODS:
            labels = cluster(cv, method)
            assert labels == gold["labels"][f"{vname}/{method}"], (vname, method)
            sizes = np.bincount(labels)[1:]
            assert (sizes >= 2).sum() >= 3 and (sizes == 1).sum() >= 1, "degenerate clustering"
