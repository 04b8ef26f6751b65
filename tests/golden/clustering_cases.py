"""Deterministic confidence-matrix cases shared by the golden-vector generator and the tests.

Matrices are regenerated from (kind, N, seed, dtype); the golden file stores their SHA-256 so that a
drift of the generator (numpy version) is detected instead of silently changing the inputs."""
import hashlib

import numpy as np


def make_confs(kind: str, n: int, seed: int, dtype: str) -> np.ndarray:
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        m = rng.random((n, n))
    elif kind == "blocks":                      # article blocks: high confidence inside, low across
        k = max(1, n // 6)
        lab = rng.integers(0, k, size=n)
        same = lab[:, None] == lab[None, :]
        m = np.where(same, rng.uniform(0.55, 0.999, (n, n)), rng.uniform(0.001, 0.45, (n, n)))
        flip = rng.random((n, n)) < 0.04        # a few contradicting edges
        m = np.where(flip, 1.0 - m, m)
    elif kind == "ties":                        # exact 0.0, 1.0 and 0.5 entries
        m = rng.choice(np.array([0.0, 0.25, 0.5, 0.5, 0.75, 1.0]), size=(n, n))
        m = np.where(rng.random((n, n)) < 0.5, m, rng.random((n, n)))
    elif kind == "symmetric":
        a = rng.random((n, n))
        m = (a + a.T) / 2
    else:
        raise ValueError(kind)
    return m.astype(dtype)


def digest(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


CASES = []
for _n in (2, 3, 12, 50, 200):
    for _kind in ("uniform", "blocks", "ties", "symmetric"):
        for _dtype in ("float32", "float64"):
            if _n == 200 and _dtype == "float64" and _kind != "blocks":
                continue
            CASES.append({"kind": _kind, "n": _n, "seed": 1000 + _n * 7 + len(_kind), "dtype": _dtype})

METHODS = ("dbscan", "greedy", "dbscan_std", "linkage")
DBSCAN_VARIANTS = (
    {"min_neighbors_for_cluster": 1, "confidence_threshold": 0.5, "cluster_agreement_threshold": 0.5},
    {"min_neighbors_for_cluster": 2, "confidence_threshold": 0.6, "cluster_agreement_threshold": 0.4},
    {"min_neighbors_for_cluster": 3, "confidence_threshold": 0.3, "cluster_agreement_threshold": 0.7,
     "assign_noise_clusters": False},
)
