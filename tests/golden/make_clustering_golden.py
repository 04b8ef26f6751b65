"""Generates tests/golden/clustering_golden.json by IMPORTING the reference implementation

    /root/reference/article_separation/gnn/clustering/{dbscan,textblock_clustering}.py
    /root/reference/python_util/basic/misc.py (split_list), python_util/geometry/point.py (rescale_points)

in the authoring container (the reference does not travel to the GPU box; only these vectors do).
Two shims are needed to import it here: a stub module for `kneed` (only the elbow
method uses it) and `numpy.math = math` (removed in numpy 2).

Run:  python tests/golden/make_clustering_golden.py
"""
import json
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")
sys.modules["kneed"] = types.ModuleType("kneed")
np.math = math

from article_separation.gnn.clustering.dbscan import DBScanRelation  # noqa: E402
from article_separation.gnn.clustering.textblock_clustering import TextblockClustering  # noqa: E402
from python_util.basic.misc import split_list  # noqa: E402
from python_util.geometry.point import rescale_points  # noqa: E402

import clustering_cases as cc  # noqa: E402


class Flags:
    def __init__(self, params):
        self.clustering_params = params


def to_py(x):
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (np.floating,)):
        return float(x)
    if isinstance(x, (list, tuple)):
        return [to_py(v) for v in x]
    return x


def main():
    import scipy, sklearn
    out = {"versions": {"numpy": np.__version__, "scipy": scipy.__version__, "sklearn": sklearn.__version__},
           "textblock": [], "dbscan": [], "split_list": [], "rescale_points": []}
    for case in cc.CASES:
        confs = cc.make_confs(case["kind"], case["n"], case["seed"], case["dtype"])
        for method in cc.METHODS:
            if method in ("linkage",) and case["n"] < 3:
                pass
            tb = TextblockClustering(Flags({}))
            tb.set_confs(confs.copy())
            try:
                tb.calc(method)
            except Exception as e:  # keep the reference's failure modes as golden facts, too
                out["textblock"].append({**case, "method": method, "sha256": cc.digest(confs), "error": type(e).__name__})
                continue
            out["textblock"].append({
                **case, "method": method, "sha256": cc.digest(confs),
                "tb_labels": to_py(list(tb.tb_labels)), "tb_classes": to_py(tb.tb_classes),
                "num_classes": int(tb.num_classes), "num_noise": int(tb.num_noise),
                "rel_LLH": float(tb.rel_LLH), "info": tb.get_info(method),
            })
        for variant in cc.DBSCAN_VARIANTS:
            if case["n"] < 3:
                continue
            db = DBScanRelation(**variant)
            labels = db.cluster_relations(case["n"], confs.copy())
            out["dbscan"].append({**case, "params": variant, "sha256": cc.digest(confs), "labels": to_py(labels)})
    for n_items, n_chunks in [(10, 3), (7, 7), (3, 5), (50, 8), (0, 2), (101, 4)]:
        out["split_list"].append({"n_items": n_items, "n": n_chunks,
                                  "result": split_list(list(range(n_items)), n_chunks)})
    for pts, sc in [([(10, 20), (33, 47)], 0.5), ([(1, 1), (2999, 4499)], 1 / 3.0), ([(7, 9)], 2.5)]:
        out["rescale_points"].append({"points": pts, "scale": sc, "result": to_py(rescale_points(pts, sc))})
    path = os.path.join(HERE, "clustering_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes;", len(out["textblock"]), "textblock cases,", len(out["dbscan"]), "dbscan cases")


if __name__ == "__main__":
    main()
