"""CPU ORACLE (test infrastructure, NOT product code) -- classical image stages around the ARU-Net.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.

PARITY UNPINNED: every function here restates the *documented* behaviour of a third-party call
(OpenCV >= 4.1 / rasterio-GDAL) that the reference makes; neither library is installed in the build
image and the reference holds no fixtures for these stages.  The pins
are hand-checkable cases in tests/test_oracle_classical.py plus cross-checks against scipy.ndimage
where scipy implements the same mathematical operation.

    net_post_processing_helper.py:14-33      scale_image / load_and_scale_image (cv2.resize, cvtColor)
    region_net_post_processor_base.py:230-251  apply_cc_analysis (cv2.connectedComponentsWithStats)
    separator_net_post_processor.py:26-97    post_process (cv2.morphologyEx MORPH_OPEN, cv2.subtract)
    swt_dist_trafo.py:18-66                  distance_transform / connected_components_cv / clean
    heading_net_post_processor.py:218-270    per text line SWT features and mean net probability
"""
import math

import numpy as np
from scipy import ndimage

_EIGHT = np.ones((3, 3), dtype=bool)


# ---------------------------------------------------------------------------------------------
# a1: image scaling
# ---------------------------------------------------------------------------------------------
def cv_round(x):
    """cvRound / saturate_cast<int>(double): round half to even."""
    return int(np.rint(x))


def bgr2gray(bgr):
    """cv2.cvtColor(BGR2GRAY) on uint8, OpenCV 4.x 15-bit fixed point.  A 2-D input is a gray file that cv2.imread
    would have replicated to three equal channels: the weights sum to 2^15, so the value comes back unchanged."""
    bgr = np.asarray(bgr)
    if bgr.ndim == 2:
        return bgr.astype(np.uint8)
    b = bgr[..., 0].astype(np.int64)
    g = bgr[..., 1].astype(np.int64)
    r = bgr[..., 2].astype(np.int64)
    return ((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15).astype(np.uint8)


def area_table(ssize, dsize, scale):
    """computeResizeAreaTab: list of (dst index, src index, weight float32) in OpenCV's order."""
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1 = int(math.ceil(fsx1))
        sx2 = int(math.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def resize_area(img, sc):
    """cv2.resize(img, None, fx=sc, fy=sc, INTER_AREA) for sc < 1 on uint8 [H,W] or [H,W,C].

    dsize = round(src*sc); scale = 1/sc.  Integer scales take the block-mean fast path
    (sum * float(1/area), round half even; 2x2 uses (s+2)>>2); otherwise the float32 table path:
    per source row   buf[dx] = sum_k S[sx_k]*alpha_k   (k ascending, float32)
    per dest row     sum[dx] = sum_j beta_j*buf_j[dx]  (j ascending, float32), then round half even."""
    img = np.asarray(img)
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    H, W, C = img.shape
    dw, dh = cv_round(W * sc), cv_round(H * sc)
    scale = 1.0 / sc
    iscale = cv_round(scale)
    out = np.empty((dh, dw, C), dtype=np.uint8)
    if abs(scale - iscale) < np.finfo(np.float64).eps:
        s = iscale
        # the fast path only covers full blocks; OpenCV falls back per pixel on the ragged border with the
        # number of available samples as divisor
        for dy in range(dh):
            y0, y1 = dy * s, min(dy * s + s, H)
            for dx in range(dw):
                x0, x1 = dx * s, min(dx * s + s, W)
                blk = img[y0:y1, x0:x1].astype(np.int64).reshape(-1, C)
                tot = blk.sum(axis=0)
                if (y1 - y0) == s and (x1 - x0) == s:
                    if s == 2:
                        out[dy, dx] = (tot + 2) >> 2
                    else:
                        v = tot.astype(np.float32) * np.float32(1.0 / (s * s))
                        out[dy, dx] = np.clip(np.rint(v), 0, 255)
                else:
                    cnt = (y1 - y0) * (x1 - x0)
                    v = tot.astype(np.float32) / np.float32(max(cnt, 1))
                    out[dy, dx] = np.clip(np.rint(v), 0, 255)
        return out[:, :, 0] if squeeze else out
    xtab = area_table(W, dw, scale)
    ytab = area_table(H, dh, scale)
    # horizontal pass for every source row (float32, sequential accumulation)
    buf = np.zeros((H, dw, C), dtype=np.float32)
    first = np.ones(dw, dtype=bool)
    for dx, sx, a in xtab:
        term = img[:, sx, :].astype(np.float32) * a
        if first[dx]:
            buf[:, dx, :] = term
            first[dx] = False
        else:
            buf[:, dx, :] = buf[:, dx, :] + term
    acc = np.zeros((dh, dw, C), dtype=np.float32)
    firsty = np.ones(dh, dtype=bool)
    for dy, sy, b in ytab:
        term = buf[sy] * b
        if firsty[dy]:
            acc[dy] = term
            firsty[dy] = False
        else:
            acc[dy] = acc[dy] + term
    out[:] = np.clip(np.rint(acc), 0, 255)
    return out[:, :, 0] if squeeze else out


def _cubic_coeffs(x):
    """interpolateCubic, A = -0.75 (float32 like OpenCV)."""
    A = np.float32(-0.75)
    x = np.float32(x)
    c0 = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A
    c1 = ((A + 2) * x - (A + 3)) * x * x + 1
    c2 = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1
    c3 = np.float32(1.0) - c0 - c1 - c2
    return [np.float32(c0), np.float32(c1), np.float32(c2), np.float32(c3)]


def cubic_table(ssize, dsize, scale):
    """(src base index, 4 int16 weights scaled by 2^11) per destination index."""
    idx = np.empty(dsize, dtype=np.int64)
    wts = np.empty((dsize, 4), dtype=np.int64)
    for dx in range(dsize):
        fx = np.float32((dx + 0.5) * scale - 0.5)
        sx = int(math.floor(fx))
        fx = np.float32(fx - sx)
        idx[dx] = sx
        c = _cubic_coeffs(fx)
        wts[dx] = [int(np.clip(np.rint(np.float32(v) * np.float32(2048.0)), -32768, 32767)) for v in c]
    return idx, wts


def resize_cubic(img, sc):
    """cv2.resize(img, None, fx=sc, fy=sc, INTER_CUBIC) on uint8: 11-bit fixed-point taps, replicated border,
    result = saturate((sum_y beta * (sum_x alpha * S) + 2^21) >> 22)."""
    img = np.asarray(img)
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    H, W, C = img.shape
    dw, dh = cv_round(W * sc), cv_round(H * sc)
    scale = 1.0 / sc
    xi, xw = cubic_table(W, dw, scale)
    yi, yw = cubic_table(H, dh, scale)
    src = img.astype(np.int64)
    hbuf = np.zeros((H, dw, C), dtype=np.int64)
    for k in range(4):
        cols = np.clip(xi - 1 + k, 0, W - 1)
        hbuf += src[:, cols, :] * xw[:, k][None, :, None]
    acc = np.zeros((dh, dw, C), dtype=np.int64)
    for k in range(4):
        rows = np.clip(yi - 1 + k, 0, H - 1)
        acc += hbuf[rows] * yw[:, k][:, None, None]
    out = np.clip((acc + (1 << 21)) >> 22, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def scale_image(image, fixed_height=None, scaling_factor=1.0):
    """net_post_processing_helper.py:14-26 (image_stats.get_scaling_factor inlined for the height case)."""
    H, W = image.shape[:2]
    if fixed_height is not None and scaling_factor is not None and 0.1 < scaling_factor:
        sc = scaling_factor * fixed_height / H
    elif fixed_height:
        sc = fixed_height / H
    else:
        sc = scaling_factor
    if sc < 1.0:
        image = resize_area(image, sc)
    elif sc > 1.0:
        image = resize_cubic(image, sc)
    return image, sc


def scale_and_gray(bgr, fixed_height, scaling_factor):
    """load_and_scale_image minus the file decode: -> (image u8 [h,w,3], image_grey float64 [h,w] in 0..1, sc)."""
    image, sc = scale_image(bgr, fixed_height, scaling_factor)
    return image, bgr2gray(image) / 255.0, sc


# ---------------------------------------------------------------------------------------------
# a9: connected components + rectangular morphology (binary, 0 / non-zero)
# ---------------------------------------------------------------------------------------------
def cc_filter(mask, min_size):
    """apply_cc_analysis: keep 8-connected components with area >= min_size, output 0/255."""
    lab, n = ndimage.label(np.asarray(mask) != 0, structure=_EIGHT)
    if n == 0:
        return np.zeros(mask.shape, dtype=np.uint8)
    area = np.bincount(lab.ravel(), minlength=n + 1)
    keep = area >= min_size
    keep[0] = False
    return (keep[lab] * 255).astype(np.uint8)


def cc_min_size(mask_size, threshold):
    """base:244 ``int(net_output.size * threshold)`` (double arithmetic, may be 99 or 100 for 1/size*100)."""
    return int(mask_size * threshold)


def _line_window_hit(target, k, axis):
    """hit(x) = any ``target`` pixel inside the cv window [x - k//2, x - k//2 + k - 1] clipped to the image."""
    a = k // 2
    b = k - 1 - a
    n = target.shape[axis]
    c = np.concatenate([np.zeros_like(np.take(target, [0], axis=axis), dtype=np.int64),
                        np.cumsum(target, axis=axis, dtype=np.int64)], axis=axis)
    x = np.arange(n)
    lo = np.clip(x - a, 0, n)
    hi = np.clip(x + b + 1, 0, n)
    return (np.take(c, hi, axis=axis) - np.take(c, lo, axis=axis)) > 0


def erode_rect(mask, kw, kh):
    """cv2.erode with a kw x kh rectangle, anchor (kw//2, kh//2), border = +inf (never erodes from outside)."""
    fg = np.asarray(mask) != 0
    if kw < 1 or kh < 1:
        raise ValueError("kernel size must be >= 1")
    if kw > 1:
        fg = ~_line_window_hit(~fg, kw, 1)
    if kh > 1:
        fg = ~_line_window_hit(~fg, kh, 0)
    return (fg * 255).astype(np.uint8)


def dilate_rect(mask, kw, kh):
    """cv2.dilate with the same (un-reflected) window, border = -inf."""
    fg = np.asarray(mask) != 0
    if kw < 1 or kh < 1:
        raise ValueError("kernel size must be >= 1")
    if kw > 1:
        fg = _line_window_hit(fg, kw, 1)
    if kh > 1:
        fg = _line_window_hit(fg, kh, 0)
    return (fg * 255).astype(np.uint8)


def open_rect(mask, kw, kh):
    """cv2.morphologyEx(MORPH_OPEN): erode then dilate with the same kernel and anchor.  For even sizes the
    un-reflected window makes the result shift by one pixel towards +x / +y (OpenCV behaviour)."""
    return dilate_rect(erode_rect(mask, kw, kh), kw, kh)


def erode_rect_bruteforce(mask, kw, kh, dilate=False):
    """O(k) definition-level version used to pin the cumulative-sum implementation above."""
    fg = np.asarray(mask) != 0
    H, W = fg.shape
    ax, ay = kw // 2, kh // 2
    out = np.zeros((H, W), dtype=bool) if dilate else np.ones((H, W), dtype=bool)
    for y in range(H):
        for x in range(W):
            vals = []
            for j in range(kh):
                for i in range(kw):
                    yy, xx = y + j - ay, x + i - ax
                    if 0 <= yy < H and 0 <= xx < W:
                        vals.append(fg[yy, xx])
            out[y, x] = (any(vals) if dilate else all(vals))
    return (out * 255).astype(np.uint8)


def separator_kernel_sizes(H, W):
    """separator_net_post_processor.py:70,75,85."""
    return int(15 * W / 1000), int(30 * H / 1500), int(10 * W / 1000)


def separator_post_process(mask_hwc):
    """separator_net_post_processor.py:26-97 -> {"horizontal": u8[H,W], "vertical": u8[H,W]}."""
    m = np.asarray(mask_hwc)[:, :, 0]
    H, W = m.shape
    post = cc_filter(m, cc_min_size(m.size, 1 / m.size * 100))
    kh_w, kv_h, kc_w = separator_kernel_sizes(H, W)
    horizontal = open_rect(post, kh_w, 1)
    vertical = open_rect(post, 1, kv_h)
    horizontal = np.where(vertical > horizontal, 0, horizontal - np.minimum(vertical, horizontal)).astype(np.uint8)
    horizontal = open_rect(horizontal, kc_w, 1)
    return {"horizontal": horizontal, "vertical": vertical}


# ---------------------------------------------------------------------------------------------
# a12: stroke-width distance transform
# ---------------------------------------------------------------------------------------------
def gaussian5(img):
    """cv2.GaussianBlur(img, (5,5), 0) on uint8: taps [1,4,6,4,1]/16 per axis, BORDER_REFLECT_101, 8.8 fixed point
    -> (sum_ij w_i w_j s_ij + 128) >> 8."""
    s = np.asarray(img).astype(np.int64)
    p = np.pad(s, 2, mode="reflect")
    taps = (1, 4, 6, 4, 1)
    H, W = s.shape
    h = sum(t * p[:, i:i + W] for i, t in enumerate(taps))
    v = sum(t * h[j:j + H, :] for j, t in enumerate(taps))
    return ((v + 128) >> 8).astype(np.uint8)


def otsu_threshold(img):
    """getThreshVal_Otsu_8u (double arithmetic, first maximum wins)."""
    hist = np.bincount(np.asarray(img, dtype=np.uint8).ravel(), minlength=256).astype(np.int64)
    n = int(hist.sum())
    scale = 1.0 / n
    mu = 0.0
    for i in range(256):
        mu += i * float(hist[i])
    mu *= scale
    mu1 = 0.0
    q1 = 0.0
    max_sigma = 0.0
    max_val = 0
    eps = float(np.finfo(np.float32).eps)
    for i in range(256):
        p_i = float(hist[i]) * scale
        mu1 *= q1
        q1 += p_i
        q2 = 1.0 - q1
        if min(q1, q2) < eps or max(q1, q2) > 1.0 - eps:
            continue
        mu1 = (mu1 + i * p_i) / q1
        mu2 = (mu - q1 * mu1) / q2
        sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2)
        if sigma > max_sigma:
            max_sigma = sigma
            max_val = i
    return max_val


def edt_sq(binary):
    """Exact squared Euclidean distance (int64) of every non-zero pixel to the nearest zero pixel."""
    fg = np.asarray(binary) != 0
    if fg.all():
        # no zero pixel: OpenCV's precise transform yields a huge value; the engine reports the same sentinel
        return np.full(fg.shape, np.iinfo(np.int32).max, dtype=np.int64)
    d = ndimage.distance_transform_edt(fg)
    return np.rint(d * d).astype(np.int64)


def swt_distance_transform(gray):
    """swt_dist_trafo.py:18-29 minus the file decode: 255-gray -> blur -> Otsu -> exact EDT -> astype(uint8)
    (truncation; values >= 256 wrap like the C cast)."""
    inv = (255 - bgr2gray(np.asarray(gray, dtype=np.uint8)).astype(np.int64)).astype(np.uint8)
    blur = gaussian5(inv)
    thr = otsu_threshold(blur)
    binary = ((blur > thr) * 255).astype(np.uint8)
    d2 = edt_sq(binary)
    d = np.sqrt(d2.astype(np.float32)).astype(np.float32)
    return (np.floor(d).astype(np.int64) & 255).astype(np.uint8)


def connected_component_boxes(image):
    """swt_dist_trafo.py:31-41: (x, y, w, h) of every 8-connected component of non-zero pixels, in raster
    first-touch order."""
    lab, n = ndimage.label(np.asarray(image) != 0, structure=_EIGHT)
    boxes = []
    for sl in ndimage.find_objects(lab):
        ys, xs = sl
        boxes.append((xs.start, ys.start, xs.stop - xs.start, ys.stop - ys.start))
    return boxes


def clean_connected_components(components, clean_ccs=2):
    """swt_dist_trafo.py:43-66."""
    out = []
    for (x, y, w, h) in components:
        if clean_ccs > 0 and (w < 3 or h < 3 or h > 500 or w > 500):
            continue
        if clean_ccs > 1 and (w / h > 8 or h / w > 8):
            continue
        out.append((x, y, w, h))
    return out


def swt_features_textline(swt, bbox):
    """heading_net_post_processor.py:218-245; bbox = (x, y, width, height) of the line's surrounding polygon."""
    x, y, w, h = bbox
    crop = swt[y:y + h + 1, x:x + w + 1]
    ccs = clean_connected_components(connected_component_boxes(crop))
    vals = []
    height = 0
    for (cx, cy, cw, ch) in ccs:
        vals.append(np.max(crop[cy:cy + ch, cx:cx + cw]))
        height = max(height, ch)
    return (float(np.median(vals)) if vals else 0.0), height


def net_prob_textline(net_output, bbox):
    """heading_net_post_processor.py:247-270 with the already rescaled bbox (x, y, width, height)."""
    x, y, w, h = bbox
    return float(np.sum(net_output[y:y + h, x:x + w]) / (w * h))
