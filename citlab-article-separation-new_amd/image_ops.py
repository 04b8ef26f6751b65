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


def _workspace(device=0):
    lib = _lib.init_device(device)
    if device not in _workspaces:
        h = lib.asep_post_create()
        if not h:
            raise _lib.AsepError("asep_post_create failed: " + _lib.last_error())
        _workspaces[device] = h
    return lib, _workspaces[device]


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
