"""TEST INFRASTRUCTURE (not product code): relation-graph cases whose article clustering is NOT degenerate.

With purely random weights the pair classifier answers 0.4-0.6 for every pair and every clustering method returns
"one article" or "all singletons" -- a label comparison on such a case cannot fail.  ``planted_articles_case`` builds
a page with planted articles instead:

  * node features: one centre per article in feature space + per-article spread; a few outlier nodes;
  * message / LSTM weights: the seeded reference initialiser (weights.init_gnn_weights);
  * pair classifier: hand-built so that ``p(same) = sigmoid(s * (t - |h_a - h_b|_1))`` on the hidden states the
    message passing produced (layer 1: +-(h_a - h_b) -> ReLU, layer 2: sums the two halves = |.|, logits: -s * sum + s*t),
    plus small dense noise on every classifier tensor so that no weight is exactly zero;
  * ``t`` / ``s`` are calibrated on the ORACLE's hidden states so that confidences straddle 0.5: most intra-article
    pairs above, most inter-article pairs below, a controlled fraction on the wrong side.

The case is accepted only if the (golden-pinned) clustering code finds >= 3 articles with >= 2 members and >= 1
singleton on the oracle confidences.  Used by tests/test_gnn_articles_gpu.py, tests/test_oracle_gnn.py and smoke().
"""
import numpy as np

from . import gnn_oracle

CLS = gnn_oracle.CLS


def planted_graph(seed, N=200, n_pairs=10000, n_articles=6, n_outliers=3, node_dim=7, edge_dim=2):
    rng = np.random.default_rng(seed)
    member = rng.integers(0, n_articles, size=N)
    member[:n_articles] = np.arange(n_articles)                       # every article has a member
    centres = rng.random((n_articles, node_dim))
    spread = rng.uniform(0.01, 0.06, size=n_articles)
    u = centres[member] + rng.normal(0, 1, size=(N, node_dim)) * spread[member, None]
    outliers = rng.choice(N, size=n_outliers, replace=False)
    u[outliers] = rng.random((n_outliers, node_dim)) * 1.6 - 0.3      # outside the articles' range
    member[outliers] = -1
    iu, ju = np.triu_indices(N, k=1)
    n_pairs = min(n_pairs, iu.shape[0])
    sel = rng.choice(iu.shape[0], size=n_pairs, replace=False)
    flip = rng.random(n_pairs) < 0.5
    a, b = iu[sel], ju[sel]
    edges = np.stack([np.where(flip, b, a), np.where(flip, a, b)], axis=1).astype(np.int32)
    rng.shuffle(edges, axis=0)
    ef = (rng.random((n_pairs, edge_dim)) < 0.15).astype(np.float32)
    return {"num_nodes": N, "interacting_nodes": edges, "node_features": u.astype(np.float32), "edge_features": ef,
            "planted": member}


def l1_classifier(w, cfg, threshold, sharpness, seed, noise=0.01):
    """Overwrites the Classification/logits tensors of ``w`` (default widths 64,32 -> 2)."""
    H = cfg.hidden_dim
    h1, h2 = cfg.classifier_hidden
    assert h1 == 2 * H and h2 == H and cfg.num_classes == 2, "the hand-built classifier needs the default 64,32 -> 2"
    rng = np.random.default_rng(seed)
    eye = np.eye(H, dtype=np.float32)
    c1 = np.concatenate([np.concatenate([eye, -eye], axis=1), np.concatenate([-eye, eye], axis=1)], axis=0)   # [2H, 2H]
    c2 = np.concatenate([eye, eye], axis=0)                                                                  # [2H, H]
    c3 = np.zeros((H, 2), np.float32)
    c3[:, 1] = -sharpness
    out = dict(w)
    out[f"{CLS}/fully_connected_layer_h1/weights"] = (c1 + rng.normal(0, noise, c1.shape)).astype(np.float32)
    out[f"{CLS}/fully_connected_layer_h1/bias"] = rng.normal(0, noise, (h1,)).astype(np.float32)
    out[f"{CLS}/fully_connected_layer_h2/weights"] = (c2 + rng.normal(0, noise, c2.shape)).astype(np.float32)
    out[f"{CLS}/fully_connected_layer_h2/bias"] = np.abs(rng.normal(0, noise, (h2,))).astype(np.float32)
    out[f"{CLS}/fully_connected_logit_layer_out/weights"] = (c3 + rng.normal(0, noise * sharpness, c3.shape)).astype(np.float32)
    out[f"{CLS}/fully_connected_logit_layer_out/bias"] = np.array([0.0, sharpness * threshold], np.float32)
    return out


def calibrate_l1_classifier(w_init, cfg, N, edges, node_feat, edge_feat, same, wrong_side=0.08, seed=1, hidden_gain=1.0):
    """Weights = ``w_init`` with the hand-built L1 classifier, its sharpness scaled to the spread of the oracle's
    hidden-state distances on this graph and its bias set so that the fraction ``wrong_side`` of the pairs marked in
    the boolean matrix ``same`` ends up below 0.5 (calibrated on a float64 oracle run of the noisy classifier itself)."""
    _, h = gnn_oracle.forward(N, edges, node_feat, edge_feat, None, w_init, cfg, return_hidden=True)
    d = np.abs(h[:, None, :] - h[None, :, :]).sum(-1)
    off = ~np.eye(N, dtype=bool)
    s = float(hidden_gain * 4.0 / max(np.std(d[off]), 1e-6))
    w = l1_classifier(w_init, cfg, 0.0, s, seed)
    p64 = gnn_oracle.forward(N, edges, node_feat, edge_feat, None, w, cfg, dtype=np.float64)
    logit = (np.log(p64[:, 1]) - np.log(p64[:, 0])).reshape(N, N)
    shift = -float(np.quantile(logit[np.asarray(same, bool) & off], wrong_side))
    w[f"{CLS}/fully_connected_logit_layer_out/bias"] = np.array([0.0, shift], np.float32)
    return w


def planted_articles_case(w_init, cfg, seed=7, N=200, n_pairs=10000, n_articles=6, n_outliers=3, wrong_side=0.08,
                          hidden_gain=1.0):
    """-> (graph dict, weights dict, oracle probs [N*N, 2] float32).  ``w_init`` = seeded random weights."""
    g = planted_graph(seed, N, n_pairs, n_articles, n_outliers, cfg.node_feature_dim, cfg.edge_feature_dim)
    same = (g["planted"][:, None] == g["planted"][None, :]) & (g["planted"][:, None] >= 0)
    w = calibrate_l1_classifier(w_init, cfg, N, g["interacting_nodes"], g["node_features"], g["edge_features"], same,
                                wrong_side, seed + 1, hidden_gain)
    probs = gnn_oracle.forward(N, g["interacting_nodes"], g["node_features"], g["edge_features"], None, w, cfg)
    return g, w, probs
