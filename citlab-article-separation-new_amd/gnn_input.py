"""Graph-json -> GNN feed arrays (host side of the GNN path).

Restates ``article_separation/gnn/input/input_dataset.py``: ``get_input_and_target_from_json`` (:343-375),
``build_full_relations`` (:444-457), ``mask_features`` (:378-383), the feed assembly of ``_parse_function`` /
``_map_element`` (:140-312) at batch size 1, and the ratio-aware image resize of
``python_util/image_processing/image_resizer.py:111-223`` (TF1 legacy bilinear: src = dst * in/out, no half-pixel
offset; ``tf.round`` = round-half-to-even).
"""
import json
import logging

import numpy as np

from .cli_flags import update_params

DEFAULT_INPUT_PARAMS = {
    "node_feature_dim": 4, "edge_feature_dim": 0, "node_input_feature_mask": [], "edge_input_feature_mask": [],
    "num_parallel_load": 4, "prefetch_load": 4, "load_mode": "L", "resize_max_dim": 1024, "resize_min_dim": 256,
    "pad_to_max_dim": False,
}


def get_input_and_target_from_json(path_to_json):
    with open(path_to_json, "r") as f:
        data = json.load(f)
    out = {
        "num_nodes": np.array(data["num_nodes"], dtype=np.int32),
        "interacting_nodes": np.array(data["interacting_nodes"], dtype=np.int32),
        "num_interacting_nodes": np.array(data["num_interacting_nodes"], dtype=np.int32),
        "node_features": np.array(data["node_features"], dtype=np.float32),
        "edge_features": np.array(data["edge_features"], dtype=np.float32),
    }
    if "visual_regions_nodes" in data and "num_points_visual_regions_nodes" in data:
        out["num_points_visual_regions_nodes"] = np.array(data["num_points_visual_regions_nodes"], dtype=np.int32)
        out["visual_regions_nodes"] = np.stack([np.array(data["visual_regions_nodes"][i], dtype=np.float32)
                                                for i in range(data["num_nodes"])])
    if "visual_regions_edges" in data and "num_points_visual_regions_edges" in data:
        out["num_points_visual_regions_edges"] = np.array(data["num_points_visual_regions_edges"], dtype=np.int32)
        out["visual_regions_edges"] = np.stack([np.array(data["visual_regions_edges"][i], dtype=np.float32)
                                                for i in range(data["num_interacting_nodes"])])
    out["gt_relations"] = np.array(data.get("gt_relations", []), dtype=np.int32)
    out["gt_num_relations"] = np.array(data.get("gt_num_relations", 0), dtype=np.int32)
    return out


def mask_features(features, mask):
    """Keep the columns whose mask entry is truthy (tf.gather over tf.where(mask))."""
    idx = np.flatnonzero(np.asarray(mask, dtype=bool))
    return np.asarray(features)[..., idx]


def build_full_relations(num_nodes, gt_relations=None):
    n = int(num_nodes)
    idx = np.tile(np.arange(n, dtype=np.int32), [n, 1])
    relations = np.stack([idx.T, idx], axis=2).reshape([-1, 2])
    gt = np.zeros([n, n], dtype=np.int32)
    if gt_relations is not None and np.size(gt_relations):
        g = np.asarray(gt_relations).reshape(-1, 3)
        gt[g[:, 1], g[:, 2]] = 1
    return relations, np.array(relations.shape[0], dtype=np.int32), gt.reshape([-1])


def compute_new_size(height, width, min_dimension, max_dimension):
    """image_resizer.py:197-223 in float32 like the dynamic-shape TF graph; round half to even."""
    h, w = np.float32(height), np.float32(width)
    small = np.float32(max_dimension) / max(h, w)
    large = max(np.float32(min_dimension) / min(h, w), np.float32(1.0))
    scale = min(small, large)
    return int(np.rint(h * scale)), int(np.rint(w * scale))


def resize_bilinear_tf1(image, new_h, new_w):
    """tf.image.resize(BILINEAR, align_corners=False) of TF 1.x: src = dst * (in/out), clamped at the border."""
    img = np.asarray(image)
    if img.dtype not in (np.uint8, np.float32):                # (uint8 pixels widen exactly: they are gathered first, converted after)
        img = img.astype(np.float32)
    H, W = img.shape[:2]
    ys = np.arange(new_h, dtype=np.float32) * np.float32(H / new_h)
    xs = np.arange(new_w, dtype=np.float32) * np.float32(W / new_w)
    y0 = np.floor(ys).astype(np.int64); x0 = np.floor(xs).astype(np.int64)
    y1 = np.minimum(y0 + 1, H - 1); x1 = np.minimum(x0 + 1, W - 1)
    wy = (ys - y0).astype(np.float32)[:, None, None]; wx = (xs - x0).astype(np.float32)[None, :, None]
    if img.ndim == 2:
        img = img[:, :, None]
    r0, r1 = img[y0], img[y1]
    f = lambda a: a.astype(np.float32, copy=False)
    top = f(r0[:, x0]) * (1 - wx) + f(r0[:, x1]) * wx
    bot = f(r1[:, x0]) * (1 - wx) + f(r1[:, x1]) * wx
    return (top * (1 - wy) + bot * wy).astype(np.float32)


class InputGNN(object):
    """``InputGNN(flags)`` with ``flags.input_params`` (dict) and ``flags.image_input`` (bool)."""

    def __init__(self, flags):
        self._flags = flags
        self.input_params = dict(DEFAULT_INPUT_PARAMS)
        given = getattr(flags, "input_params", None) or {}
        for k in given:
            if k not in self.input_params:
                logging.critical(f"Given input_params-key '{k}' is not used by class 'InputGNN'!")
        self.input_params.update(given)
        if not (self.input_params["resize_max_dim"] > 0 and self.input_params["resize_min_dim"] > 0):
            raise ValueError("Error in resizing parameters for input image.")

    def _masked(self, feats, which):
        mask = self.input_params[f"{which}_input_feature_mask"]
        dim = self.input_params[f"{which}_feature_dim"]
        if len(mask) > 0:
            if len(mask) != dim:
                raise ValueError(f"Length of {which} feature mask ({len(mask)}) doesn't match provided {which} "
                                 f"feature dim ({dim}).")
            return mask_features(feats, mask)
        return feats

    def feed_from_json(self, json_path, image=None):
        """-> feed dict keyed by the exported placeholder names (batch size 1), ready for ``GnnSession.run``."""
        d = get_input_and_target_from_json(json_path)
        n = int(d["num_nodes"])
        feed = {
            "num_nodes:0": np.array([n], np.int32),
            "num_interacting_nodes:0": np.array([int(d["num_interacting_nodes"])], np.int32),
            "interacting_nodes:0": d["interacting_nodes"].reshape(-1, 2)[None],
        }
        if self.input_params["node_feature_dim"] > 0:
            feed["node_features:0"] = self._masked(d["node_features"], "node").astype(np.float32)[None]
        if self.input_params["edge_feature_dim"] > 0:
            ef = d["edge_features"].reshape(int(d["num_interacting_nodes"]), -1)
            feed["edge_features:0"] = self._masked(ef, "edge").astype(np.float32)[None]
        if getattr(self._flags, "image_input", False) and image is not None:
            img = np.asarray(image)                                  # uint8 as decoded, or float32 (values 0..255 either way)
            if img.ndim == 2:
                img = img[:, :, None]
            nh, nw = compute_new_size(img.shape[0], img.shape[1], self.input_params["resize_min_dim"],
                                      self.input_params["resize_max_dim"])
            feed["image:0"] = resize_bilinear_tf1(img, nh, nw)[None]
            feed["image_shape:0"] = np.array([[nh, nw, img.shape[2]]], np.int32)
            for k in ("visual_regions_nodes", "num_points_visual_regions_nodes"):
                if k in d:
                    feed[k + ":0"] = d[k][None]
        rel, _, _ = build_full_relations(n, d["gt_relations"])
        feed["relations_to_consider_belong_to_same_instance:0"] = rel[None]
        return feed
