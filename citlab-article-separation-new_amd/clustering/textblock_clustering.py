"""Text-block clustering front end (confidences -> labels/classes/relative log-likelihood).

Host-side restatement of ``article_separation/gnn/clustering/textblock_clustering.py:11-328`` with the
same public interface (``TextblockClustering(flags)``, ``set_confs``, ``calc(method)``, ``get_info``,
result attributes ``tb_labels / tb_classes / num_classes / num_noise / rel_LLH``), pinned by
``tests/golden/clustering_golden.json``.  ``flags`` only needs a ``clustering_params`` dict.

dtype rule: the matrix keeps the dtype it arrives in (float32 from the net,
float64 after masking); the 0/1 clamps use ``nextafter`` in that dtype.
"""
import logging
import math

import numpy as np
from scipy.cluster.hierarchy import cut_tree, fcluster, linkage
from scipy.stats import gmean
from sklearn.cluster import dbscan as sk_dbscan
from sklearn.metrics import silhouette_score

from .dbscan import DBScanRelation

DEFAULT_PARAMS = {
    # dbscan
    "min_neighbors_for_cluster": 1, "confidence_threshold": 0.5, "cluster_agreement_threshold": 0.5,
    "assign_noise_clusters": True,
    # linkage
    "method": "centroid", "criterion": "distance", "t": -1.0, "max_clusters": 100,
    # greedy
    "max_iteration": 1000,
    # dbscan_std
    "epsilon": 0.5, "min_samples": 1,
}


class TextblockClustering(object):
    def __init__(self, flags):
        self._flags = flags
        self.clustering_params = dict(DEFAULT_PARAMS)
        given = getattr(flags, "clustering_params", None) or {}
        for key in given:
            if key not in self.clustering_params:
                logging.critical(f"Given input_params-key '{key}' is not used by class 'TextblockClustering'!")
        self.clustering_params.update(given)
        self.tb_labels = None
        self.tb_classes = None
        self.num_classes = 0
        self.num_noise = 0
        self.rel_LLH = 0.0
        self._conf_mat = None
        self._mat_dim = None
        self._dist_mat = None
        self._cond_dists = None
        self._delta_mat = None
        self._dbscanner = None

    def print_params(self):
        logging.info("CLUSTERING:")
        for k, v in sorted(self.clustering_params.items()):
            logging.info(f"  {k}: {v}")

    def get_info(self, method):
        p = self.clustering_params
        if not hasattr(self, f"_{method}"):
            return None
        return {
            "dbscan": f'dbscan_conf{p["confidence_threshold"]}_cluster{p["cluster_agreement_threshold"]}',
            "dbscan_std": f'dbscan_std_eps{p["epsilon"]}_samples{p["min_samples"]}',
            "linkage": f'linkage_{p["method"]}_{p["criterion"]}_t{p["t"]}',
            "greedy": f'greedy_iter{p["max_iteration"]}',
        }.get(method)

    # -- input ---------------------------------------------------------------------------------
    def set_confs(self, confs, symmetry_fn=gmean):
        mat = np.array(confs)
        self._mat_dim = mat.shape[0]
        # clamp exact 0 / 1 to the neighbouring float of the matrix dtype (no log(0), no division by 0)
        mat[mat == 0.0] = np.nextafter(0, 1, dtype=mat.dtype)
        mat[mat == 1.0] = np.nextafter(1, 0, dtype=mat.dtype)
        if symmetry_fn:
            mat = symmetry_fn(np.stack([mat, mat.transpose()], axis=-1), axis=-1)
        self._conf_mat = mat
        self._dist_mat = -np.log(mat)
        np.fill_diagonal(self._dist_mat, 0.0)
        self._cond_dists = self._dist_mat[np.triu_indices_from(self._dist_mat, k=1)]
        self._delta_mat = np.log(mat / (1 - mat))
        np.fill_diagonal(self._delta_mat, -math.inf)

    # -- dispatch ------------------------------------------------------------------------------
    def calc(self, method):
        self.tb_labels = None
        self.tb_classes = None
        if self._mat_dim == 2:
            thr = self.clustering_params["confidence_threshold"]
            self.tb_labels = [1, 1] if self._conf_mat[0, 1] >= thr else [1, 2]
        else:
            fn = getattr(self, f"_{method}", None)
            if fn is None:
                raise NotImplementedError(f'Cannot find clustering method "_{method}"!')
            fn()
        self._calc_relative_LLH()

    def _labels2classes(self):
        groups = {}
        for tb, cls in enumerate(self.tb_labels):
            groups.setdefault(cls, []).append(tb)
        self.tb_classes = [sorted(v) for v in groups.values()]

    def _classes2labels(self):
        labels = np.full(self._mat_dim, -1, dtype=int)
        for idx, cls in enumerate(self.tb_classes):
            for tb in cls:
                labels[tb] = idx
        self.tb_labels = labels

    def _calc_relative_LLH(self):
        total = 0.0
        lab = self.tb_labels
        for i in range(self._mat_dim):
            if lab[i] >= 0:
                for k in range(i):
                    if lab[i] == lab[k]:
                        total += (self._delta_mat[i, k] + self._delta_mat[k, i]) / 2
        self.rel_LLH = total

    def _finish(self):
        self.num_classes = len(self.tb_classes)
        self.num_noise = len([l for l in self.tb_labels if l == -1])

    # -- methods -------------------------------------------------------------------------------
    def _dbscan(self):
        p = self.clustering_params
        if not self._dbscanner:
            self._dbscanner = DBScanRelation(
                min_neighbors_for_cluster=p["min_neighbors_for_cluster"],
                confidence_threshold=p["confidence_threshold"],
                cluster_agreement_threshold=p["cluster_agreement_threshold"],
                assign_noise_clusters=p["assign_noise_clusters"])
        self.tb_labels = self._dbscanner.cluster_relations(self._mat_dim, self._conf_mat)
        self._labels2classes()
        self._finish()

    def _dbscan_std(self):
        p = self.clustering_params
        _, self.tb_labels = sk_dbscan(self._dist_mat, metric='precomputed', min_samples=p["min_samples"],
                                      eps=p["epsilon"])
        self._labels2classes()
        self._finish()

    def _greedy(self):
        n = self._mat_dim
        self.tb_labels = np.arange(n, dtype=int)
        self._labels2classes()
        self._calcMat = self._delta_mat.copy()
        budget = self.clustering_params["max_iteration"]
        while budget > 0:
            budget -= 1
            i, j = np.unravel_index(np.argmax(self._calcMat), self._conf_mat.shape)
            if not self._calcMat[i, j] > 0:
                break
            self._greedy_step(i, j)
            self._classes2labels()
            self._calc_relative_LLH()
        self.tb_classes = [c for c in self.tb_classes if len(c) > 0]
        self.num_classes = len(self.tb_classes)
        self._classes2labels()
        self.num_noise = len([l for l in self.tb_labels if l == -1])

    def _greedy_step(self, keep, drop):
        self.tb_classes[keep] = sorted(self.tb_classes[keep] + self.tb_classes[drop])
        self.tb_classes[drop] = []
        m = self._calcMat
        for idx in range(self._mat_dim):
            if idx != keep and idx != drop:
                m[idx, keep] += m[idx, drop]
                m[keep, idx] = m[idx, keep]
        for idx in range(self._mat_dim):
            m[idx, drop] = -math.inf
            m[drop, idx] = m[idx, drop]

    def _linkage(self):
        p = self.clustering_params
        res = linkage(self._cond_dists, method=p["method"])
        if p["t"] == -1:
            heights = res[:, 2]
            t = 1 / 2 * (float(np.mean(heights)) + float(np.median(heights)))
            self.tb_labels = fcluster(res, t=t, criterion=p["criterion"])
        else:
            _, self.tb_labels = self._validate_clusters(res)
        self._labels2classes()
        self._finish()

    def _validate_clusters(self, linkage_res):
        p = self.clustering_params
        max_clusters = min(self._mat_dim, p["max_clusters"])
        tree = np.transpose(cut_tree(linkage_res)[:, ::-1])[:max_clusters, :]
        labels_list = tree.tolist()
        scores = []
        for k, labels in enumerate(labels_list, start=1):
            if k == 1:
                upper = self._conf_mat[np.triu_indices_from(self._conf_mat, k=1)]
                if np.all(upper >= p["confidence_threshold"]):
                    return 1, labels_list[0]
                continue
            try:
                scores.append(silhouette_score(self._dist_mat, labels, metric='precomputed'))
            except ValueError:
                scores.append(0.0)
        if p["t"] == "silhouette":
            k = int(np.argmax(scores)) + 2
            return k, labels_list[k - 1]
        if p["t"] == "merge":
            raise NotImplementedError("t='merge' needs the 'kneed' elbow detector, which is not installed")
        logging.error(f'Clustering param t = {p["t"]} not in validity indices. Defaulting to num_clusters = 1')
        return 1, labels_list[0]
