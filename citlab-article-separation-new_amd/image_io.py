"""Image file decode for the two ARU-Net pipelines (host side, Pillow instead of ``cv2.imread``).

    load_image_bgr          cv2.imread(path)                         helper:29   (8-bit, BGR channel order)
    load_image_gray         cv2.imread(path, IMREAD_GRAYSCALE)       swt_dist_trafo.py:19
    get_image_dimensions    image_stats.py:23-29                     (width, height)
    load_and_scale_image    helper:28-33                             decode here, scale + gray on the GPU

Grayscale files are kept single-channel: ``cv2.imread`` would replicate the channel three times and BGR2GRAY of
equal channels returns the value itself (3735 + 19235 + 9798 = 2^15), so the net input is identical.
JPEG decoders differ between libjpeg builds by +-1 in places; PNG / TIFF inputs are bit-identical.
"""
import numpy as np
from PIL import Image, ImageOps

from . import image_ops

Image.MAX_IMAGE_PIXELS = None        # newspaper scans exceed Pillow's decompression-bomb guard


def get_image_dimensions(image_path):
    with Image.open(image_path) as im:
        return im.size


_DEEP_GRAY = ("I;16", "I;16L", "I;16B", "I;16N", "I")      # Pillow modes of 16-bit (and wider) single-channel files


def load_image_bgr(path_to_image):
    """uint8 [H,W,3] in BGR order, or uint8 [H,W] for single-channel files.

    Like ``cv2.imread`` with its default flags: the EXIF orientation is applied, an alpha channel is dropped, palette
    files are expanded, and 16-bit samples are reduced to their high byte (libpng's ``strip_16``; Pillow's own
    ``convert('L')`` would clip everything above 255 to white instead)."""
    with Image.open(path_to_image) as im:
        im = ImageOps.exif_transpose(im)
        if im.mode in _DEEP_GRAY:
            deep = np.asarray(im).astype(np.int64)
            return np.clip(deep >> 8, 0, 255).astype(np.uint8)
        if im.mode == "F":
            return np.clip(np.rint(np.asarray(im, dtype=np.float64)), 0, 255).astype(np.uint8)
        if im.mode in ("L", "1", "LA", "La"):
            return np.asarray(im if im.mode == "L" else im.convert("L"), dtype=np.uint8)
        rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return np.ascontiguousarray(rgb[:, :, ::-1])


def load_image_gray(path_to_image):
    """8-bit gray like ``cv2.imread(path, cv2.IMREAD_GRAYSCALE)``: colour files go through the BGR2GRAY weights
    (OpenCV converts inside the decoder; for PNG/TIFF that is the same fixed-point formula)."""
    img = load_image_bgr(path_to_image)
    if img.ndim == 2:
        return img
    b, g, r = (img[:, :, i].astype(np.int32) for i in range(3))
    return ((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15).astype(np.uint8)


def load_and_scale_image(path_to_image, fixed_height, scaling_factor, device=0):
    """helper:28-33 -> (image uint8 scaled, image_grey float32 [h,w] in 0..1, sc)."""
    image = load_image_bgr(path_to_image)
    return image_ops.scale_and_gray(image, fixed_height, scaling_factor, device=device)
