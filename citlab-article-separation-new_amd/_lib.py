"""ctypes binding of ``csrc/libasep_hip.so`` (declared in ``include/asep_hip.h``).

There is deliberately NO CPU fallback: if the HIP library is missing, cannot be loaded, or no
gfx950 device is present, every compute entry point raises (the oracle under ``oracle/`` is test
infrastructure and is never imported from here).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libasep_hip.so")


class AsepError(RuntimeError):
    pass


MLP_MAX_HIDDEN = 4       # GNN_MLP_MAX (csrc/gnn_kernels.h): hidden layers of the interaction / attention / classifier MLPs
ABI_VERSION = 6          # ASEP_ABI_VERSION of include/asep_hip.h this table was written against


class _SizedCfg(C.Structure):
    """configuration structs start with their own size (``struct_size``): filled in here, checked by the load functions"""

    def __init__(self, *args, **kw):
        super().__init__(C.sizeof(type(self)), *args, **kw)


class AruCfg(_SizedCfg):
    _fields_ = [(n, C.c_int32) for n in (
        "struct_size", "channels", "n_classes", "feat_root", "scale_space_num", "res_depth", "num_scales_att",
        "use_attention", "mvn", "apply_softmax", "compute_dtype", "activation", "plain_u")]


class GnnCfg(_SizedCfg):
    _fields_ = [(n, C.c_int32) for n in (
        "struct_size", "node_feature_dim", "edge_feature_dim", "num_transition_steps", "hidden_dim", "interaction_dim",
        "interaction_hidden", "cls_hidden1", "cls_hidden2", "num_classes", "undirected_graph", "compress_input_dim", "output_type", "attention_heads", "attention_merge", "attention_hidden",
        "aggregation_type", "interaction_hidden2", "interaction_hidden3", "interaction_hidden4", "attention_hidden2", "attention_hidden3",
        "attention_hidden4", "cls_hidden3", "cls_hidden4", "lstm_use_hidden", "lstm_use_input", "visual_edge_dims")]


class GnnPage(C.Structure):
    """asep_gnn_page (include/asep_hip.h): one page of asep_gnn_forward_visual_batch_dev, device addresses"""
    _fields_ = [("N", C.c_int32), ("E", C.c_int32), ("R", C.c_int32), ("d_edges", C.c_void_p), ("d_node_feat", C.c_void_p),
                ("d_edge_feat", C.c_void_p), ("d_image", C.c_void_p), ("d_regions", C.c_void_p), ("d_num_points", C.c_void_p),
                ("d_edge_regions", C.c_void_p), ("d_edge_num_points", C.c_void_p), ("d_relations", C.c_void_p), ("d_probs_out", C.c_void_p)]


# name -> (restype, argtypes); mirrors include/asep_hip.h one to one
_P = C.c_void_p
SIGNATURES = {
    "asep_device_count": (C.c_int, []),
    "asep_init": (C.c_int, [C.c_int]),
    "asep_last_error": (C.c_char_p, []),
    "asep_version": (C.c_char_p, []),
    "asep_abi_version": (C.c_int, []),
    "asep_engine_switches": (C.c_char_p, []),
    "asep_aru_load": (_P, [_P, C.c_size_t, C.POINTER(AruCfg)]),
    "asep_aru_free": (None, [_P]),
    "asep_aru_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P, C.c_float]),
    "asep_aru_forward_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P, C.c_float, _P]),
    "asep_aru_forward_batch_dev": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, _P, _P, _P, C.c_float, _P]),
    "asep_aru_trim": (C.c_int, [_P]),
    "asep_aru_forward_batch_dev2": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P, C.c_float, _P]),
    "asep_aru_get_endpoint": (C.c_long, [_P, C.c_char_p, _P, C.c_size_t, C.POINTER(C.c_int32)]),
    "asep_aru_profile": (C.c_int, [_P, C.c_int]),
    "asep_aru_profile_report": (C.c_long, [_P, C.c_char_p, C.c_size_t]),
    "asep_aru_flops": (C.c_double, [_P, C.c_int, C.c_int]),
    "asep_host_alloc": (_P, [C.c_size_t]),
    "asep_host_free": (None, [_P]),
    "asep_host_register": (C.c_int, [_P, C.c_size_t]),
    "asep_host_unregister": (C.c_int, [_P]),
    "asep_gnn_load": (_P, [_P, C.c_size_t, C.POINTER(GnnCfg)]),
    "asep_gnn_free": (None, [_P]),
    "asep_gnn_correct_edges": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P]),
    "asep_gnn_forward": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, C.c_int, _P, _P]),
    "asep_gnn_forward_dev": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, C.c_int, _P, _P, _P]),
    "asep_gnn_get_hidden": (C.c_int, [_P, _P, C.c_size_t]),
    "asep_gnn_flops": (C.c_double, [_P, C.c_int, C.c_int, C.c_int]),
    "asep_gnn_attach_backbone": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_char_p)]),
    "asep_gnn_forward_visual": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P,
                                          C.c_int, _P, _P]),
    "asep_gnn_forward_visual_dev": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P,
                                              C.c_int, _P, _P, _P]),
    "asep_gnn_forward_visual_batch_dev": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P]),
    "asep_gnn_step_mode": (C.c_int, [_P]),
    "asep_gnn_get_node_features": (C.c_int, [_P, _P, C.c_size_t]),
    "asep_post_create": (_P, []),
    "asep_post_free": (None, [_P]),
    "asep_prep_scaled_size": (C.c_int, [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "asep_prep_scale_gray": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_double, _P, _P]),
    "asep_prep_scale_gray_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_double, _P, _P, _P]),
    "asep_post_cc_filter": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "asep_post_morph_rect": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "asep_post_separator": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, _P, _P]),
    "asep_post_separator_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, _P, _P, _P]),
    "asep_post_boundary_segments": (C.c_long, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_long]),
    "asep_post_boundary_segments_dev": (C.c_long, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_long, _P]),
    "asep_post_boundary_segments_enqueue_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_long, _P,
                                                          _P]),
    "asep_prep_gray_u8_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "asep_post_box_sums_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "asep_swt_distance_transform": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.POINTER(C.c_int32), _P]),
    "asep_swt_distance_transform_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "asep_swt_line_features": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "asep_swt_line_features_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
}

_lib = None


def load_library(path: str = None):
    """dlopen the HIP library (once).  ``torch`` is imported first on purpose: PyTorch-ROCm ships its
    own ``libamdhip64`` and both runtimes must resolve to the same copy inside one process."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("ASEP_HIP_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise AsepError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C {os.path.dirname(LIB_PATH)}`); there is no CPU fallback")
    try:
        import torch  # noqa: F401  (HIP runtime de-duplication, see docstring)
    except Exception:  # pragma: no cover - torch is plumbing only
        pass
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.asep_abi_version() != ABI_VERSION:
        raise AsepError(f"{path} implements ABI version {lib.asep_abi_version()}, this binding was written against {ABI_VERSION}: "
                        f"rebuild the library (make -C {os.path.dirname(LIB_PATH)})")
    _lib = lib
    return lib


def last_error() -> str:
    return load_library().asep_last_error().decode("utf-8", "replace")


def check(rc: int, what: str):
    if rc < 0:
        raise AsepError(f"{what} failed ({rc}): {last_error()}")
    return rc


_initialised = {}


def init_device(device_id: int = 0):
    lib = load_library()
    if device_id not in _initialised:
        if lib.asep_device_count() <= 0:
            raise AsepError("no HIP device visible: the MI355X (gfx950) engine has no CPU fallback")
        check(lib.asep_init(device_id), "asep_init")
        _initialised[device_id] = True
    return lib
