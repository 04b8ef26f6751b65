"""Drop-in mirror of the reference's GNN inference boundary

    article_separation/gnn/io.py:12-25            load_graph(pb_path)
    article_separation/gnn/run_gnn_clustering.py:216-269
        sess = tf.Session(graph=graph); out = sess.run(output_node, feed_dict=...)

``load_graph`` returns a :class:`GnnGraph`; ``GnnSession(graph).run(fetch, feed_dict)`` accepts the
reference's feed keys *by tensor name* (``run_gnn_clustering.py:76-148``, names fixed at export time by
``model_relation.py:258-337``) and returns ``[1, R, num_classes]`` float32 like the frozen graph did.
The TensorFlow runtime underneath is replaced by ``csrc/libasep_hip.so`` (``include/asep_hip.h``).
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from .config import GnnConfig
from .weights import load_weights, pack_blob

OUTPUT_NODE = "output_belong_to_same_instance:0"
FEED_NAMES = (
    "num_nodes:0", "num_interacting_nodes:0", "interacting_nodes:0", "node_features:0", "edge_features:0",
    "image:0", "image_shape:0", "visual_regions_nodes:0", "num_points_visual_regions_nodes:0",
    "visual_regions_edges:0", "num_points_visual_regions_edges:0",
    "relations_to_consider_belong_to_same_instance:0",
)


class GnnGraph:
    def __init__(self, tensors, cfg: GnnConfig, path: str = None):
        self.tensors = tensors
        self.cfg = cfg
        self.path = path
        self._blob = None
        self._handles = {}
        self._backbones = {}

    def blob(self) -> bytes:
        """GNN + compression-layer tensors (the backbone goes into its own ARU-Net handle)."""
        if self._blob is None:
            self._blob = pack_blob({k: v for k, v in self.tensors.items() if not k.startswith("aru_net/")})
        return self._blob

    def backbone_graph(self):
        """ARU-Net over the ``aru_net/...`` tensors of the same frozen graph (graph_relation.py:17-22)."""
        from .net_post_processing_helper import AruGraph
        return AruGraph({k: v for k, v in self.tensors.items() if k.startswith("aru_net/")}, self.cfg.backbone_cfg(),
                        self.path)

    def handle(self, device_id: int = 0):
        if device_id not in self._handles:
            lib = _lib.init_device(device_id)
            c = self.cfg
            def widths(name, lst, lo, hi=_lib.MLP_MAX_HIDDEN):
                """the hidden-width list of an MLP as `hi` struct fields (0 = absent)"""
                lst = [int(v) for v in lst]
                if not lo <= len(lst) <= hi or any(v < 1 for v in lst):
                    raise _lib.AsepError(f"{name} = {lst}: the engine serves {lo} to {hi} hidden layers of positive width")
                return lst + [0] * (hi - len(lst))
            ih = widths("num_hidden_units_interaction_fct", c.interaction_hidden, 1)
            ah = widths("num_hidden_units_attention_fct", c.attention_hidden, 1) if c.use_attention else [0, 0, 0, 0]
            ch = widths("num_hidden_units (classifier)", c.classifier_hidden, 1)
            if c.visual_edges and not c.visual_dims:
                raise _lib.AsepError("visual_edges needs the visual branch (visual_dims / visual_layers)")
            cfg = _lib.GnnCfg(c.u_dim, c.edge_in_dim, c.num_transition_steps, c.hidden_dim,
                              c.interaction_dim, ih[0], ch[0], ch[1], c.num_classes, int(c.undirected_graph),
                              c.u_in_dim if c.compress_node_feature_dim > 0 else 0, c.output_type_code,
                              c.num_attention_heads if c.use_attention else 0,
                              {"concat": 0, "average": 1}[c.multihead_attention_merge_type], ah[0],
                              c.aggregation_code, ih[1], ih[2], ih[3], ah[1], ah[2], ah[3], ch[2], ch[3],
                              int(c.incorporate_hidden_features_in_update), int(c.incorporate_node_input_features_in_update),
                              c.visual_edge_dim)
            blob = self.blob()
            h = lib.asep_gnn_load(blob, len(blob), C.byref(cfg))
            if not h:
                raise _lib.AsepError("asep_gnn_load failed: " + _lib.last_error())
            if c.visual_dims:
                bb = self.backbone_graph()
                names = (C.c_char_p * len(c.visual_layers))(*[n.encode() for n in c.visual_layers])
                rc = lib.asep_gnn_attach_backbone(h, bb.handle(device_id), len(c.visual_layers), names)
                if rc < 0:
                    lib.asep_gnn_free(h)
                    raise _lib.AsepError("asep_gnn_attach_backbone failed: " + _lib.last_error())
                self._backbones[device_id] = bb          # keeps the ARU-Net handle alive
            self._handles[device_id] = h
        return self._handles[device_id]

    def close(self):
        if self._handles:
            lib = _lib.load_library()
            for h in self._handles.values():
                lib.asep_gnn_free(h)
            self._handles = {}
            for bb in self._backbones.values():
                bb.close()
            self._backbones = {}

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def load_graph(pb_path, visual_layers=None, num_transition_steps=None) -> GnnGraph:
    """gnn/io.py:12-25.  Accepts a TF1 frozen graph (``*.pb``, decoded without TensorFlow by ``pb_import.py``)
    or the engine's ``*.asepw`` container (+ ``.json`` side-car).  ``visual_layers`` names the backbone end points
    of a graph exported with ``--image_input`` (``--feature_map_generation_params from_layer=[...]``); the names
    are not recoverable from the constants alone, the default is ``scale_0_unet_up_<level>_conv`` per map.
    ``num_transition_steps`` is read from the op graph (one edge-MLP MatMul per step); only a constants-only container
    without that structure needs it passed in."""
    if isinstance(pb_path, GnnGraph):
        return pb_path
    if not os.path.isfile(pb_path):
        raise IOError(f"No such model file: {pb_path}")
    if str(pb_path).endswith(".pb"):
        from . import pb_import
        tensors, cfg = pb_import.gnn_from_nodes(pb_import.read_graph(pb_path), visual_layers=visual_layers,
                                                num_transition_steps=num_transition_steps)
    else:
        tensors, meta = load_weights(pb_path)
        cfg = GnnConfig(**(meta or {}).get("gnn_cfg", {}))
    if cfg.visual_dims:                                      # ASEP_COMPUTE_DTYPE: the conv backbone of the visual branch (graph stays fp32)
        from .net_post_processing_helper import compute_dtype_from_env
        cfg.backbone = dict(cfg.backbone, compute_dtype=compute_dtype_from_env(cfg.backbone.get("compute_dtype", cfg.backbone_cfg().compute_dtype)))
    return GnnGraph(tensors, cfg, pb_path)


def _key(k):
    """feed_dict keys may be tensor names ('num_nodes:0'), bare names, or objects with a .name."""
    name = getattr(k, "name", k)
    if not isinstance(name, str):
        raise KeyError(f"unsupported feed key {k!r}")
    return name if ":" in name else name + ":0"


class GnnSession:
    """Stands in for ``tf.Session(graph=graph)`` in ``run_gnn_clustering.py:221``."""

    def __init__(self, graph: GnnGraph, gpu_devices="0"):
        self.graph = graph
        self.device = 0 if gpu_devices in (None, "") else int(str(gpu_devices).split(",")[0])

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def run(self, fetches, feed_dict):
        name = _key(fetches)
        if name != OUTPUT_NODE:
            raise KeyError(f"The name '{name}' refers to a Tensor which does not exist (only {OUTPUT_NODE})")
        feed = {_key(k): v for k, v in feed_dict.items()}
        for k in feed:
            if k not in FEED_NAMES:
                raise KeyError(f"The name '{k}' refers to a Tensor which does not exist")
        cfg = self.graph.cfg
        num_nodes = np.asarray(feed["num_nodes:0"]).reshape(-1)
        if num_nodes.shape[0] != 1:
            raise ValueError("batch size must be 1 (input_dataset.py:134)")
        N = int(num_nodes[0])
        E = int(np.asarray(feed["num_interacting_nodes:0"]).reshape(-1)[0])
        edges = np.ascontiguousarray(np.asarray(feed["interacting_nodes:0"], dtype=np.int32)[0][:E])
        if "node_features:0" in feed:
            u = np.ascontiguousarray(np.asarray(feed["node_features:0"], dtype=np.float32)[0][:N])
        elif cfg.visual_dims and cfg.node_feature_dim == 0:
            u = np.zeros((N, 0), dtype=np.float32)
        else:
            raise KeyError("feed_dict lacks node_features:0")
        if u.shape[1] != cfg.node_feature_dim:
            raise ValueError(f"node_features has dim {u.shape[1]}, model expects {cfg.node_feature_dim}")
        ef = None
        if cfg.edge_feature_dim:
            ef = np.ascontiguousarray(np.asarray(feed["edge_features:0"], dtype=np.float32)[0][:E])
            if ef.shape[1] != cfg.edge_feature_dim:
                raise ValueError(f"edge_features has dim {ef.shape[1]}, model expects {cfg.edge_feature_dim}")
        rel = np.ascontiguousarray(
            np.asarray(feed["relations_to_consider_belong_to_same_instance:0"], dtype=np.int32)[0])
        if cfg.visual_dims:
            for k in ("image:0", "visual_regions_nodes:0", "num_points_visual_regions_nodes:0"):
                if k not in feed:
                    raise KeyError(f"this graph was exported with image_input: feed_dict lacks {k}")
            image = np.asarray(feed["image:0"], dtype=np.float32)
            if image.shape[0] != 1:
                raise ValueError("batch size must be 1 (input_dataset.py:134)")
            image = image[0]
            if "image_shape:0" in feed:                     # crop to the true shape (no padding at batch size 1)
                ish = np.asarray(feed["image_shape:0"]).reshape(-1, 3)[0]
                image = image[:int(ish[0]), :int(ish[1])]
            regions = np.asarray(feed["visual_regions_nodes:0"], dtype=np.float32)[0][:N]
            npts = np.asarray(feed["num_points_visual_regions_nodes:0"], dtype=np.int32)[0][:N]
            eregions = enpts = None
            if cfg.visual_edges:                            # graph_relation.py:141-146
                for k in ("visual_regions_edges:0", "num_points_visual_regions_edges:0"):
                    if k not in feed:
                        raise KeyError(f"this graph assigns visual features to edges: feed_dict lacks {k}")
                eregions = np.asarray(feed["visual_regions_edges:0"], dtype=np.float32)[0][:E]
                enpts = np.asarray(feed["num_points_visual_regions_edges:0"], dtype=np.int32)[0][:E]
            probs = gnn_forward_visual(self.graph, N, edges, u, ef, image, regions, npts, rel, self.device,
                                       edge_regions=eregions, edge_num_points=enpts)
        else:
            probs = gnn_forward(self.graph, N, edges, u, ef, rel, self.device)
        return probs[None]


def gnn_forward(graph: GnnGraph, num_nodes, edges, node_feat, edge_feat, relations=None, device=0):
    """One page through the engine -> probabilities [R, num_classes] (R = N*N when relations is None)."""
    lib = _lib.init_device(device)
    N = int(num_nodes)
    edges = np.ascontiguousarray(edges, dtype=np.int32).reshape(-1, 2)
    E = edges.shape[0]
    u = np.ascontiguousarray(node_feat, dtype=np.float32).reshape(N, -1)
    ef = np.ascontiguousarray(edge_feat, dtype=np.float32).reshape(E, -1) if edge_feat is not None else None
    if relations is None:
        R, rel_p = N * N, None
    else:
        rel = np.ascontiguousarray(relations, dtype=np.int32).reshape(-1, 2)
        R, rel_p = rel.shape[0], rel.ctypes.data
    out = np.empty((R, graph.cfg.num_classes), dtype=np.float32)
    rc = lib.asep_gnn_forward(graph.handle(device), N, E, edges.ctypes.data if E else None, u.ctypes.data,
                              ef.ctypes.data if (ef is not None and E) else None, R, rel_p, out.ctypes.data)
    _lib.check(rc, "asep_gnn_forward")
    return out


def gnn_forward_visual(graph: GnnGraph, num_nodes, edges, node_feat, edge_feat, image, regions, num_points,
                       relations=None, device=0, edge_regions=None, edge_num_points=None):
    """graph_relation.py:84-139 + GNN: image float32 [h,w(,1)] as fed (0..255), regions [N,2,P] relative
    coordinates, num_points [N] -> probabilities [R, num_classes].  ``edge_regions`` [E,2,P] / ``edge_num_points`` [E]: the
    interactions' regions of a graph with ``visual_edges`` (graph_relation.py:141-172)."""
    lib = _lib.init_device(device)
    cfg = graph.cfg
    N = int(num_nodes)
    edges = np.ascontiguousarray(edges, dtype=np.int32).reshape(-1, 2)
    E = edges.shape[0]
    u = np.ascontiguousarray(node_feat, dtype=np.float32).reshape(N, cfg.node_feature_dim)
    ef = np.ascontiguousarray(edge_feat, dtype=np.float32).reshape(E, -1) if edge_feat is not None else None
    img = np.asarray(image, dtype=np.float32)
    if img.ndim == 3:
        if img.shape[2] != 1:
            raise ValueError("the ARU_v1 backbone takes one image channel")
        img = img[:, :, 0]
    img = np.ascontiguousarray(img)
    reg = np.ascontiguousarray(regions, dtype=np.float32)
    if reg.ndim != 3 or reg.shape[0] != N or reg.shape[1] != 2:
        raise ValueError(f"visual_regions_nodes must be [N, 2, P], got {reg.shape}")
    npts = np.ascontiguousarray(num_points, dtype=np.int32).reshape(N)
    ereg_p = enp_p = None
    if cfg.visual_edges:
        if edge_regions is None or edge_num_points is None:
            raise ValueError("this graph assigns visual features to edges: edge_regions / edge_num_points required")
        ereg = np.ascontiguousarray(edge_regions, dtype=np.float32)
        if ereg.shape != (E, 2, reg.shape[2]):
            raise ValueError(f"visual_regions_edges must be [E, 2, P] = {(E, 2, reg.shape[2])}, got {ereg.shape}")
        enp = np.ascontiguousarray(edge_num_points, dtype=np.int32).reshape(E)
        if E:
            ereg_p, enp_p = ereg.ctypes.data, enp.ctypes.data
    if relations is None:
        R, rel_p = N * N, None
    else:
        rel = np.ascontiguousarray(relations, dtype=np.int32).reshape(-1, 2)
        R, rel_p = rel.shape[0], rel.ctypes.data
    out = np.empty((R, cfg.num_classes), dtype=np.float32)
    rc = lib.asep_gnn_forward_visual(graph.handle(device), N, E, edges.ctypes.data if E else None,
                                     u.ctypes.data if u.size else None,
                                     ef.ctypes.data if (ef is not None and ef.size and E) else None, img.ctypes.data,
                                     img.shape[0], img.shape[1], reg.ctypes.data, reg.shape[2], npts.ctypes.data,
                                     ereg_p, enp_p, R, rel_p, out.ctypes.data)
    _lib.check(rc, "asep_gnn_forward_visual")
    return out


def gnn_forward_visual_dev(graph: GnnGraph, num_nodes, num_edges, d_edges, d_node_feat, d_edge_feat, d_image, h, w, d_regions,
                           num_region_points, d_num_points, num_relations, d_relations, d_probs_out, stream=None, device=0,
                           d_edge_regions=None, d_edge_num_points=None):
    """``asep_gnn_forward_visual_dev``: every array is a device address (int), nothing is synchronised -- the call returns
    once backbone, ROI kernels and the graph are queued on ``stream`` (a hipStream_t handle as int, None = null stream)."""
    lib = _lib.init_device(device)
    rc = lib.asep_gnn_forward_visual_dev(graph.handle(device), int(num_nodes), int(num_edges), d_edges, d_node_feat,
                                         d_edge_feat, d_image, int(h), int(w), d_regions, int(num_region_points),
                                         d_num_points, d_edge_regions, d_edge_num_points, int(num_relations), d_relations,
                                         d_probs_out, stream)
    _lib.check(rc, "asep_gnn_forward_visual_dev")


def gnn_forward_visual_batch_dev(graph: GnnGraph, pages, h, w, num_region_points, stream=None, device=0):
    """``asep_gnn_forward_visual_batch_dev``: ``pages`` is a sequence of dicts with the fields of ``asep_gnn_page`` (device
    addresses as ints; ``d_relations`` may be None = all N*N ordered pairs) or a ready ``(_lib.GnnPage * n)`` array.  The
    backbones of all pages run as one grouped forward; nothing is synchronised."""
    lib = _lib.init_device(device)
    if not isinstance(pages, C.Array):
        arr = (_lib.GnnPage * len(pages))()
        for q, d in zip(arr, pages):
            for k, v in d.items():
                setattr(q, k, v)
        pages = arr
    rc = lib.asep_gnn_forward_visual_batch_dev(graph.handle(device), len(pages), pages, int(h), int(w), int(num_region_points),
                                               stream)
    _lib.check(rc, "asep_gnn_forward_visual_batch_dev")
    return pages


STEP_MODES = {0: "generic", 1: "mfma_registers", 2: "mfma_lds", 3: "factored"}


def step_mode(graph: GnnGraph, device=0) -> str:
    """which message-passing kernel the engine picked for this model (include/asep_hip.h asep_gnn_step_mode)"""
    lib = _lib.init_device(device)
    return STEP_MODES[_lib.check(lib.asep_gnn_step_mode(graph.handle(device)), "asep_gnn_step_mode")]


def gnn_node_features(graph: GnnGraph, num_nodes, device=0):
    """Concatenated [geometric | visual] node features of the last visual forward (tests)."""
    lib = _lib.init_device(device)
    out = np.empty((int(num_nodes), graph.cfg.u_in_dim), dtype=np.float32)
    _lib.check(lib.asep_gnn_get_node_features(graph.handle(device), out.ctypes.data, out.size),
               "asep_gnn_get_node_features")
    return out


def gnn_hidden(graph: GnnGraph, num_nodes, device=0):
    lib = _lib.init_device(device)
    out = np.empty((int(num_nodes), graph.cfg.hidden_dim), dtype=np.float32)
    _lib.check(lib.asep_gnn_get_hidden(graph.handle(device), out.ctypes.data, out.size), "asep_gnn_get_hidden")
    return out


def correct_edges(graph: GnnGraph, num_nodes, edges, edge_feat=None, device=0):
    """Device version of misc.py:7-151 -> (edges' [E',2] int32, features' [E',e] float32 or None)."""
    lib = _lib.init_device(device)
    edges = np.ascontiguousarray(edges, dtype=np.int32).reshape(-1, 2)
    E = edges.shape[0]
    mult = 2 if graph.cfg.undirected_graph else 1
    out_e = np.empty((max(mult * E, 1), 2), dtype=np.int32)
    ed = graph.cfg.edge_feature_dim
    ef = out_f = None
    if edge_feat is not None and ed:
        ef = np.ascontiguousarray(edge_feat, dtype=np.float32).reshape(E, ed)
        out_f = np.empty((max(mult * E, 1), ed), dtype=np.float32)
    n = lib.asep_gnn_correct_edges(graph.handle(device), int(num_nodes), E, edges.ctypes.data if E else None,
                                   ef.ctypes.data if ef is not None and E else None, out_e.ctypes.data,
                                   out_f.ctypes.data if out_f is not None else None)
    _lib.check(n, "asep_gnn_correct_edges")
    return out_e[:n].copy(), (out_f[:n].copy() if out_f is not None else None)
