"""Drop-in mirror of the reference's ARU-Net inference boundary

    article_separation/image_segmentation/net_post_processing/net_post_processing_helper.py

Same function names, argument meaning and return types; the TensorFlow session underneath is
replaced by the gfx950 engine in ``csrc/libasep_hip.so`` (C ABI: ``include/asep_hip.h``).

    load_graph(path)                      helper:36-53   -> AruGraph (weights + config + device handle)
    get_net_output(image, graph, gpu)     helper:56-72   -> float32 [H, W, n_classes]
    apply_threshold(net_output, thr)      helper:75-78
    get_scaling_factor(...)               python_util/image_processing/image_stats.py:10-20
"""
import ctypes as C
import os
import weakref

import numpy as np

from . import _lib
from .config import AruConfig
from .weights import load_weights, pack_blob


# asep_aru_cfg.compute_dtype (include/asep_hip.h): "f32" = fp32 tensors and fp32 products; "bf16" = bf16 tensors and products, fp32 accumulation;
# "f32s" = fp32 tensors and accumulation, every product of the wide convolutions as six bf16 x bf16 partial products of the three-way
# bfloat16 split of both factors (csrc/split_kernels.h) -- fp32 results on the bf16 matrix pipeline
COMPUTE_DTYPES = {"f32": 0, "bf16": 1, "f32s": 2}


class AruGraph:
    """What ``load_graph`` returns in place of a ``tf.Graph``: named weights + hyper-parameters.
    Device handles are created lazily per GPU (one model instance per (process, device))."""

    def __init__(self, tensors, cfg: AruConfig, path: str = None):
        self.tensors = tensors
        self.cfg = cfg
        self.path = path
        self._blob = None
        self._handles = {}

    # -- tensor-name contract of the frozen graph (helper:69-70) ------------------------------
    input_name = "inImg:0"
    output_name = "output:0"

    def blob(self) -> bytes:
        if self._blob is None:
            self._blob = pack_blob(self.tensors)
        return self._blob

    def handle(self, device_id: int = 0, lane: int = 0):
        """the model instance on a device; ``lane`` > 0: a further instance (own activation arena and side streams) for a
        caller that runs pages on several streams at once"""
        key = device_id if lane == 0 else (device_id, lane)
        if key not in self._handles:
            lib = _lib.init_device(device_id)
            c = self.cfg
            if c.compute_dtype not in COMPUTE_DTYPES:
                raise ValueError(f"compute_dtype must be one of {sorted(COMPUTE_DTYPES)}, got {c.compute_dtype!r}")
            cfg = _lib.AruCfg(c.channels, c.n_classes, c.feat_root, c.scale_space_num, c.res_depth,
                              c.num_scales_att, int(c.use_attention), int(c.mvn), int(c.apply_softmax),
                              COMPUTE_DTYPES[c.compute_dtype], c.activation_code, 0 if c.use_residual else 1)
            blob = self.blob()
            h = lib.asep_aru_load(blob, len(blob), C.byref(cfg))
            if not h:
                raise _lib.AsepError("asep_aru_load failed: " + _lib.last_error())
            self._handles[key] = h
        return self._handles[key]

    def close(self):
        if self._handles:
            lib = _lib.load_library()
            for h in self._handles.values():
                lib.asep_aru_free(h)
            self._handles = {}

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def flops(self, H: int, W: int) -> float:
        return _lib.load_library().asep_aru_flops(self.handle(_device_of(None)), H, W)


def load_graph(path_to_pb) -> AruGraph:
    """helper:36-53.  Accepts a TF1 frozen graph (``*.pb``: the constants are extracted without TensorFlow,
    ``pb_import.py``) or the engine's own weight container (``*.asepw`` + ``.json`` side-car written by
    ``weights.save_weights``)."""
    if isinstance(path_to_pb, AruGraph):
        return path_to_pb
    if not os.path.isfile(path_to_pb):
        raise IOError(f"No such model file: {path_to_pb}")
    if str(path_to_pb).endswith(".pb"):
        from . import pb_import
        tensors, cfg = pb_import.aru_from_nodes(pb_import.read_graph(path_to_pb))
    else:
        tensors, meta = load_weights(path_to_pb)
        cfg = AruConfig(**(meta or {}).get("aru_cfg", {}))
    cfg.compute_dtype = compute_dtype_from_env(cfg.compute_dtype)
    return AruGraph(tensors, cfg, path_to_pb)


def compute_dtype_from_env(default: str) -> str:
    """ASEP_COMPUTE_DTYPE=bf16|f32|f32s selects the engine's arithmetic for models loaded from FILES (the reference's command lines have
    no flag for it: BASELINE configs[4] runs the same CLIs with "bf16 convs").  The precision is an engine option, not a
    property of the weights; unset = what the model's side-car says, else fp32."""
    v = os.environ.get("ASEP_COMPUTE_DTYPE", "").strip().lower()
    if not v:
        return default
    if v not in COMPUTE_DTYPES:
        raise ValueError(f"ASEP_COMPUTE_DTYPE must be one of {sorted(COMPUTE_DTYPES)}, got {v!r}")
    return v


def _device_of(gpu_device) -> int:
    """Reference semantics (helper:58-66): a string like "0" selects the visible device list; ''/None
    meant "CPU only" for TensorFlow.  This engine has no CPU path, so ''/None map to device 0."""
    if gpu_device in (None, ""):
        return 0
    return int(str(gpu_device).split(",")[0])


class _PinnedPool:
    """Page-locked numpy buffers (``asep_host_alloc``).  An array handed out owns its buffer until the last view of it
    is garbage-collected; the buffer then returns to the pool, so a loop over pages re-uses the same few buffers and
    every transfer of ``get_net_output`` is a DMA instead of a staged copy from / to pageable memory."""
    KEEP = 4                                  # free buffers kept per size

    def __init__(self, lib):
        self.lib = lib
        self.free = {}

    def _give(self, ptr, nbytes):
        lst = self.free.setdefault(nbytes, [])
        if len(lst) < self.KEEP:
            lst.append(ptr)
        else:
            self.lib.asep_host_free(ptr)

    def array(self, shape, dtype):
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        lst = self.free.get(nbytes)
        ptr = lst.pop() if lst else self.lib.asep_host_alloc(nbytes)
        if not ptr:
            return np.empty(shape, dtype=dtype)               # pageable fallback: correct, only slower
        buf = (C.c_char * nbytes).from_address(ptr)
        weakref.finalize(buf, self._give, ptr, nbytes)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)


_pinned = {}
_copy_threads = None


def _parallel_copy(dst, src, min_rows=512, n_threads=4):
    """dst[...] = src (with dtype conversion) in row bands on a few threads: numpy releases the GIL while it copies, and
    one core moves only ~10 GB/s -- a 3000 x 4500 float page would spend 5 ms here, half of the net's run time"""
    global _copy_threads
    rows = dst.shape[0]
    if rows < 2 * min_rows:
        np.copyto(dst, src, casting="unsafe")
        return
    if _copy_threads is None:
        from concurrent.futures import ThreadPoolExecutor
        _copy_threads = ThreadPoolExecutor(n_threads)
    step = -(-rows // n_threads)
    list(_copy_threads.map(lambda a: np.copyto(dst[a:a + step], src[a:a + step], casting="unsafe"), range(0, rows, step)))


def _pinned_pool(dev, lib):
    if dev not in _pinned:
        _pinned[dev] = _PinnedPool(lib)
    return _pinned[dev]


def get_net_output(image, pb_graph: AruGraph, gpu_device="0"):
    """helper:56-72: image [H,W] (or [1,H,W,1]) -> net output [H,W,n_classes] float32."""
    out, _, _ = get_net_output_fused(image, pb_graph, gpu_device, want_u8=False, threshold=None)
    return out


def get_net_output_fused(image, pb_graph: AruGraph, gpu_device="0", want_u8=True, threshold=0.05):
    """One engine call that also returns the two direct consumers of the probability map:
    ``uint8(prob*255)`` (separator_net_post_processor.py:147) and ``apply_threshold`` of it
    (helper:75-78), computed in the net's epilogue on the device."""
    image = np.asarray(image)
    if image.ndim == 4:
        if image.shape[0] != 1:
            raise ValueError("batch size must be 1 (as in the reference)")
        image = image[0]
    if image.ndim == 3:
        if image.shape[2] != pb_graph.cfg.channels:
            raise ValueError(f"expected {pb_graph.cfg.channels} channel(s), got {image.shape[2]}")
        image = image[:, :, 0] if pb_graph.cfg.channels == 1 else image
    if image.ndim != 2:
        raise ValueError(f"unsupported image shape {image.shape}")
    H, W = image.shape
    ncls = pb_graph.cfg.n_classes
    dev = _device_of(gpu_device)
    lib = _lib.init_device(dev)
    pool = _pinned_pool(dev, lib)
    # the feed casts float64 -> float32: done while copying into a page-locked staging buffer (one pass over the page)
    img = pool.array((H, W), np.float32)
    _parallel_copy(img, image)
    out = pool.array((H, W, ncls), np.float32)
    u8 = pool.array((H, W, ncls), np.uint8) if want_u8 else None
    mask = pool.array((H, W, ncls), np.uint8) if (want_u8 and threshold is not None) else None
    rc = lib.asep_aru_forward(pb_graph.handle(dev), img.ctypes.data, H, W, out.ctypes.data,
                              u8.ctypes.data if u8 is not None else None,
                              mask.ctypes.data if mask is not None else None,
                              float(threshold) if threshold is not None else 0.0)
    _lib.check(rc, "asep_aru_forward")
    return out, u8, mask


def get_endpoint(pb_graph: AruGraph, name: str, gpu_device="0"):
    """Named intermediate tensor of the last forward (ARU_v1.py:11-29 end-point names), NHWC."""
    dev = _device_of(gpu_device)
    lib = _lib.init_device(dev)
    dims = (C.c_int32 * 3)()
    n = _lib.check(lib.asep_aru_get_endpoint(pb_graph.handle(dev), name.encode(), None, 0, dims),
                   "asep_aru_get_endpoint")
    out = np.empty((dims[0], dims[1], dims[2]), dtype=np.float32)
    _lib.check(lib.asep_aru_get_endpoint(pb_graph.handle(dev), name.encode(), out.ctypes.data, n, dims),
               "asep_aru_get_endpoint")
    return out


def apply_threshold(net_output, threshold):
    """helper:75-78."""
    if net_output.dtype == np.uint8:
        threshold *= 255
    return np.array((net_output > threshold) * 255, dtype=np.uint8)


def get_scaling_factor(image_height, image_width, scaling_factor, fixed_height=None, fixed_width=None):
    """Resize factor of a page (``python_util/image_processing/image_stats.py:10-20``; pinned by
    ``tests/golden/pathutil_golden.json``): a relative ``scaling_factor`` above 0.1 multiplies the factor that brings
    the page to the fixed height (else width); without it the fixed extent alone decides, then the bare factor."""
    extents = ((fixed_height, image_height), (fixed_width, image_width))
    if scaling_factor is not None and scaling_factor > 0.1:
        for target, extent in extents:
            if target is not None:
                return scaling_factor * target / extent
    for target, extent in extents:
        if target:
            return target / extent
    return scaling_factor or None
