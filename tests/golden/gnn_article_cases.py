"""Planted-article relation graphs shared by the golden generator, the CPU tests, the GPU tests and smoke()
(see oracle/gnn_cases.py for how a case is built).  A case is regenerated from its parameters; the golden file stores
the labels the REFERENCE clustering code assigns to the oracle's confidences."""
import numpy as np

CASES = [
    # name, planted_articles_case kwargs
    {"name": "c4_n200", "seed": 7, "N": 200, "n_pairs": 10000, "n_articles": 6, "n_outliers": 3, "wrong_side": 0.15},
    {"name": "c4_n200_b", "seed": 8, "N": 200, "n_pairs": 10000, "n_articles": 9, "n_outliers": 5, "wrong_side": 0.15},
    {"name": "n60", "seed": 21, "N": 60, "n_pairs": 600, "n_articles": 4, "n_outliers": 2, "wrong_side": 0.10},
    {"name": "smoke_n40", "seed": 33, "N": 40, "n_pairs": 200, "n_articles": 3, "n_outliers": 2, "wrong_side": 0.10},
]
METHODS = ("dbscan", "greedy", "linkage", "dbscan_std")
WEIGHT_SEED = 1234


def build(case):
    """-> (graph, weights, cfg, oracle probs [N*N, 2])"""
    from citlab_article_separation_new_amd.config import GnnConfig
    from citlab_article_separation_new_amd.weights import init_gnn_weights
    from oracle import gnn_cases
    cfg = GnnConfig()
    w0 = init_gnn_weights(cfg, WEIGHT_SEED, bias_jitter=0.05)
    kw = {k: v for k, v in case.items() if k != "name"}
    g, w, probs = gnn_cases.planted_articles_case(w0, cfg, **kw)
    return g, w, cfg, probs


def pair_mask(case):
    """int32 0/1 matrix like run_gnn_clustering.py:163 builds for heading / separator masking: symmetric, ~3 % of the
    pairs zeroed.  ``mask * confs`` promotes float32 confidences to float64 exactly like the reference (:186)."""
    n = case["N"]
    rng = np.random.default_rng(case["seed"] + 1000)
    m = np.ones((n, n), np.int32)
    iu, ju = np.triu_indices(n, k=1)
    z = rng.random(iu.shape[0]) < 0.03
    m[iu[z], ju[z]] = 0
    m[ju[z], iu[z]] = 0
    return m


def conf_variants(case, probs):
    """the two dtypes the clustering sees in the reference CLI: plain float32, and float64 after masking"""
    n = case["N"]
    conf = np.asarray(probs, np.float32)[:, 1].reshape(n, n)
    return {"float32": conf, "masked_float64": pair_mask(case) * conf}
