"""Classical image stages of the two ARU-Net pipelines, executed by ``csrc/libasep_hip.so`` on the GPU.

The reference calls OpenCV on the host for these; the functions below keep the
reference's names / argument meaning where it has a function of its own and name the cv2 call otherwise.

    scale_image / scale_and_gray      net_post_processing_helper.py:14-33
    apply_cc_analysis                 region_net_post_processor_base.py:230-251
    morphology_rect                   cv2.erode / dilate / morphologyEx(MORPH_OPEN / MORPH_CLOSE), rect kernels
    separator_post_process            separator_net_post_processor.py:26-97
    swt_distance_transform            python_util/image_processing/swt_dist_trafo.py:18-29

No CPU fallback: without the HIP library / a gfx950 device every function raises ``AsepError``.
"""
import ctypes as C

import numpy as np

from . import _lib
from .net_post_processing_helper import get_scaling_factor

MORPH_ERODE, MORPH_DILATE, MORPH_OPEN, MORPH_CLOSE = 0, 1, 2, 3

_workspaces = {}


def _workspace(device=0, lane=0):
    """the classical stages' scratch arena of a device; ``lane`` > 0: a second arena for calls that run on another stream
    beside the first (an arena is reused call after call in stream order, so one arena must not serve two streams)"""
    lib = _lib.init_device(device)
    key = device if lane == 0 else (device, lane)
    if key not in _workspaces:
        h = lib.asep_post_create()
        if not h:
            raise _lib.AsepError("asep_post_create failed: " + _lib.last_error())
        _workspaces[key] = h
    return lib, _workspaces[key]


def scaled_size(H, W, sc):
    h, w = C.c_int32(), C.c_int32()
    _lib.check(_lib.load_library().asep_prep_scaled_size(int(H), int(W), float(sc), C.byref(h), C.byref(w)),
               "asep_prep_scaled_size")
    return h.value, w.value


def scale_and_gray(image, fixed_height=None, scaling_factor=1.0, device=0, want_image=True):
    """``load_and_scale_image`` without the file decode (helper:28-33): ``image`` uint8 [H,W,3] BGR (or [H,W] gray)
    -> (scaled image uint8 or None, image_grey float32 [h,w] in 0..1, sc)."""
    image = np.ascontiguousarray(image, dtype=np.uint8)
    if image.ndim == 2:
        image = image[:, :, None]
    H, W, Cn = image.shape
    sc = get_scaling_factor(H, W, scaling_factor, fixed_height=fixed_height)
    lib, ws = _workspace(device)
    h, w = scaled_size(H, W, sc)
    out_img = np.empty((h, w, Cn), dtype=np.uint8) if want_image else None
    gray = np.empty((h, w), dtype=np.float32)
    _lib.check(lib.asep_prep_scale_gray(ws, image.ctypes.data, H, W, Cn, float(sc),
                                        out_img.ctypes.data if out_img is not None else None, gray.ctypes.data),
               "asep_prep_scale_gray")
    if out_img is not None and Cn == 1:
        out_img = out_img[:, :, 0]
    return out_img, gray, sc


def scale_image(image, fixed_height=None, scaling_factor=1.0, device=0):
    """helper:14-26 -> (image, sc)."""
    img, _, sc = scale_and_gray(image, fixed_height, scaling_factor, device)
    return img, sc


def apply_cc_analysis(net_output, threshold, device=0):
    """base:230-251: remove 8-connected components smaller than ``int(net_output.size * threshold)`` pixels."""
    m = np.ascontiguousarray(net_output, dtype=np.uint8)
    if m.ndim != 2:
        raise ValueError("apply_cc_analysis expects a 2-D mask")
    H, W = m.shape
    min_size = int(m.size * threshold)
    lib, ws = _workspace(device)
    out = np.empty((H, W), dtype=np.uint8)
    _lib.check(lib.asep_post_cc_filter(ws, m.ctypes.data, H, W, 1, 0, min_size, out.ctypes.data),
               "asep_post_cc_filter")
    return out


def morphology_rect(mask, op, ksize, device=0):
    """``cv2.morphologyEx(mask, op, cv2.getStructuringElement(cv2.MORPH_RECT, ksize))`` for binary masks;
    ``ksize = (width, height)`` like OpenCV."""
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    H, W = m.shape
    lib, ws = _workspace(device)
    out = np.empty((H, W), dtype=np.uint8)
    _lib.check(lib.asep_post_morph_rect(ws, int(op), m.ctypes.data, H, W, int(ksize[0]), int(ksize[1]),
                                        out.ctypes.data), "asep_post_morph_rect")
    return out


def separator_kernel_sizes(H, W):
    """separator_net_post_processor.py:70,75,85."""
    return int(15 * W / 1000), int(30 * H / 1500), int(10 * W / 1000)


def separator_post_process(net_output, device=0):
    """separator_net_post_processor.py:26-97: thresholded net output uint8 [H,W,C] -> {"horizontal", "vertical"}."""
    m = np.ascontiguousarray(net_output, dtype=np.uint8)
    if m.ndim == 2:
        m = m[:, :, None]
    H, W, Cn = m.shape
    size = H * W
    min_size = int(size * (1 / size * 100))
    k_h, k_v, k_c = separator_kernel_sizes(H, W)
    lib, ws = _workspace(device)
    hz = np.empty((H, W), dtype=np.uint8)
    vt = np.empty((H, W), dtype=np.uint8)
    _lib.check(lib.asep_post_separator(ws, m.ctypes.data, H, W, Cn, 0, min_size, k_h, k_v, k_c, hz.ctypes.data,
                                       vt.ctypes.data), "asep_post_separator")
    return {"horizontal": hz, "vertical": vt}


def boundary_segments(mask, value=255, device=0):
    """Maximal straight boundary segments of the pixels equal to ``value`` (device half of
    ``rasterio.features.shapes``, base:186-197) -> (starts, ends) int32 key arrays for
    ``polygonize.shapes_from_segments``."""
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    H, W = m.shape
    lib, ws = _workspace(device)
    cap = 1 << 14
    while True:
        st = np.empty(cap, dtype=np.int32)
        en = np.empty(cap, dtype=np.int32)
        n = _lib.check(lib.asep_post_boundary_segments(ws, m.ctypes.data, H, W, int(value), st.ctypes.data,
                                                       en.ctypes.data, cap), "asep_post_boundary_segments")
        if n <= cap:
            return st[:n], en[:n]
        cap = int(n)


def boundary_segments_dev(d_mask_ptr, H, W, value=255, device=0, stream=None, capacity=None):
    """Same on a device-resident uint8 mask; only the segment end points cross PCIe."""
    import torch
    lib, ws = _workspace(device)
    cap = int(capacity or (1 << 14))
    tdev = torch.device("cuda", device)
    while True:
        d_keys = torch.empty((2, cap), dtype=torch.int32, device=tdev)
        n = _lib.check(lib.asep_post_boundary_segments_dev(ws, d_mask_ptr, H, W, int(value), d_keys[0].data_ptr(),
                                                           d_keys[1].data_ptr(), cap, stream),
                       "asep_post_boundary_segments_dev")
        if n <= cap:
            host = d_keys[:, :n].cpu().numpy()
            return host[0], host[1]
        cap = int(n)


class DeviceImage:
    """A uint8 image that lives in HBM (held by a torch tensor); ``numpy()`` copies it back on demand."""

    def __init__(self, tensor, device=0):
        self.tensor = tensor
        self.device = device
        self.shape = tuple(tensor.shape)

    @property
    def ptr(self):
        return self.tensor.data_ptr()

    def numpy(self):
        return self.tensor.cpu().numpy()


def swt_distance_transform_device(gray, device=0):
    """swt_dist_trafo.py:18-29 with the result left on the GPU -> :class:`DeviceImage` (for ``swt_line_features``)."""
    import torch
    g = np.require(gray, dtype=np.uint8, requirements=["C", "W"])
    H, W = g.shape
    lib, ws = _workspace(device)
    tdev = torch.device("cuda", device)
    with torch.cuda.device(tdev):
        sp = C.c_void_p(torch.cuda.current_stream(tdev).cuda_stream)
        d_g = torch.from_numpy(g).to(tdev)
        d_out = torch.empty((H, W), dtype=torch.uint8, device=tdev)
        _lib.check(lib.asep_swt_distance_transform_dev(ws, d_g.data_ptr(), H, W, d_out.data_ptr(), sp),
                   "asep_swt_distance_transform_dev")
        torch.cuda.current_stream(tdev).synchronize()
    return DeviceImage(d_out, device)


def swt_distance_transform(gray, device=0, return_details=False):
    """swt_dist_trafo.py:18-29 on an already decoded uint8 gray image."""
    g = np.ascontiguousarray(gray, dtype=np.uint8)
    H, W = g.shape
    lib, ws = _workspace(device)
    out = np.empty((H, W), dtype=np.uint8)
    thr = C.c_int32(-1)
    d2 = np.empty((H, W), dtype=np.int32) if return_details else None
    _lib.check(lib.asep_swt_distance_transform(ws, g.ctypes.data, H, W, out.ctypes.data, C.byref(thr),
                                               d2.ctypes.data if d2 is not None else None),
               "asep_swt_distance_transform")
    if return_details:
        return out, thr.value, d2
    return out


def _line_features_host(swt, box):
    """Reference arithmetic for one crop (heading_net_post_processor.py:218-245) -- used for the rare lines the
    device kernel flags (more than 1024 components in one crop)."""
    from scipy import ndimage
    x0, y0, x1, y1 = box
    crop = swt[max(y0, 0):max(y1, 0), max(x0, 0):max(x1, 0)]
    lab, _ = ndimage.label(crop != 0, structure=np.ones((3, 3), dtype=bool))
    vals, height = [], 0
    for sl in ndimage.find_objects(lab):
        w, h = sl[1].stop - sl[1].start, sl[0].stop - sl[0].start
        if w < 3 or h < 3 or h > 500 or w > 500 or w / h > 8 or h / w > 8:
            continue
        vals.append(np.max(crop[sl]))
        height = max(height, h)
    return (float(np.median(vals)) if vals else 0.0), height


def box_sums_dev(d_img_ptr, shape, boxes, channel=0, device=0, stream=None, lane=0):
    """exact sums of ``img[y0:y1, x0:x1, channel]`` for a device-resident uint8 image [H,W,C] (or [H,W]); ``boxes`` =
    [[x0, y0, x1, y1], ...] with non-negative bounds -> int64 [L]  (heading_net_post_processor.py:247-270)"""
    boxes = np.ascontiguousarray(np.asarray(boxes, dtype=np.int32).reshape(-1, 4))
    out = np.zeros(boxes.shape[0], dtype=np.int64)
    if boxes.shape[0] == 0:
        return out
    H, W = shape[0], shape[1]
    stride = shape[2] if len(shape) > 2 else 1
    lib, ws = _workspace(device, lane)
    _lib.check(lib.asep_post_box_sums_dev(ws, d_img_ptr, H, W, stride, int(channel), boxes.shape[0], boxes.ctypes.data,
                                          out.ctypes.data, stream), "asep_post_box_sums_dev")
    return out


def swt_line_features(swt, boxes, device=0, d_swt_ptr=None, shape=None, stream=None, lane=0):
    """Stroke width (median of the per-component maxima) and text height (largest component height) of every text
    line crop ``swt[y0:y1, x0:x1]``; ``boxes`` = [[x0, y0, x1, y1], ...].  ``swt`` may be a host uint8 image, or
    None together with ``d_swt_ptr`` / ``shape`` for a device-resident image (``swt`` is then fetched lazily, only if
    a line has to fall back to the host).  -> (stroke_widths float64 [L], heights int [L])"""
    boxes = np.ascontiguousarray(np.asarray(boxes, dtype=np.int32).reshape(-1, 4))
    n = boxes.shape[0]
    sw = np.zeros(n, dtype=np.float32)
    hh = np.zeros(n, dtype=np.int32)
    flag = np.zeros(n, dtype=np.int32)
    if n == 0:
        return sw.astype(np.float64), hh
    lib, ws = _workspace(device, lane)
    if isinstance(swt, DeviceImage):
        d_swt_ptr, shape, dimg = swt.ptr, swt.shape, swt
        swt = dimg.numpy
    if d_swt_ptr is not None:
        H, W = shape
        _lib.check(lib.asep_swt_line_features_dev(ws, d_swt_ptr, H, W, n, boxes.ctypes.data, sw.ctypes.data,
                                                  hh.ctypes.data, flag.ctypes.data, stream),
                   "asep_swt_line_features_dev")
    else:
        g = np.ascontiguousarray(swt, dtype=np.uint8)
        H, W = g.shape
        _lib.check(lib.asep_swt_line_features(ws, g.ctypes.data, H, W, n, boxes.ctypes.data, sw.ctypes.data,
                                              hh.ctypes.data, flag.ctypes.data), "asep_swt_line_features")
    sw = sw.astype(np.float64)
    if flag.any():
        host = swt() if callable(swt) else swt
        for i in np.flatnonzero(flag):
            sw[i], hh[i] = _line_features_host(host, boxes[i])
    return sw, hh
