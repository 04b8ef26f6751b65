"""Generates tests/golden/gnn_articles_golden.json: for every planted-article case the labels that the REFERENCE's
clustering code (/root/reference/article_separation/gnn/clustering/textblock_clustering.py, imported here with the two
shims of make_clustering_golden.py) assigns to the ORACLE's confidences, for every method and for both dtypes the
reference CLI can hand to it (float32; float64 after the int32 mask product, run_gnn_clustering.py:163,186).

Run:  python tests/golden/make_gnn_articles_golden.py
"""
import json
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, "/root/reference")
sys.modules["kneed"] = types.ModuleType("kneed")
np.math = math

from article_separation.gnn.clustering.textblock_clustering import TextblockClustering  # noqa: E402

import gnn_article_cases as gac  # noqa: E402


class Flags:
    clustering_params = {}


def main():
    out = {"cases": []}
    for case in gac.CASES:
        g, w, cfg, probs = gac.build(case)
        rec = {"name": case["name"], "planted": g["planted"].tolist(), "labels": {}}
        conf = probs[:, 1]
        rec["min_abs_conf_minus_half"] = float(np.abs(conf - 0.5).min())
        rec["frac_above_half"] = float((conf > 0.5).mean())
        for vname, cv in gac.conf_variants(case, probs).items():
            for method in gac.METHODS:
                tb = TextblockClustering(Flags())
                tb.set_confs(cv.copy())
                tb.calc(method)
                labels = [int(v) for v in tb.tb_labels]
                rec["labels"][f"{vname}/{method}"] = labels
                sizes = np.bincount(labels)[1:]
                print(case["name"], vname, method, "articles>=2:", int((sizes >= 2).sum()), "singletons:",
                      int((sizes == 1).sum()))
        print(case["name"], "min|conf-0.5| = %.3e" % rec["min_abs_conf_minus_half"])
        out["cases"].append(rec)
    path = os.path.join(HERE, "gnn_articles_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
