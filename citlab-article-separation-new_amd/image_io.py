"""Image file decode for the two ARU-Net pipelines (host side, Pillow instead of ``cv2.imread``).

    load_image_bgr          cv2.imread(path)                         helper:29   (8-bit, BGR channel order)
    load_image_gray         cv2.imread(path, IMREAD_GRAYSCALE)       swt_dist_trafo.py:19
    get_image_dimensions    image_stats.py:23-29                     (width, height)
    load_and_scale_image    helper:28-33                             decode here, scale + gray on the GPU

Grayscale files are kept single-channel: ``cv2.imread`` would replicate the channel three times and BGR2GRAY of
equal channels returns the value itself (3735 + 19235 + 9798 = 2^15), so the net input is identical.
JPEG decoders differ between libjpeg builds by +-1 in places; PNG / TIFF inputs are bit-identical.
"""
import ctypes as C
import os
import struct
import threading
import zlib

import numpy as np
from PIL import Image, ImageOps

from . import image_ops

Image.MAX_IMAGE_PIXELS = None        # newspaper scans exceed Pillow's decompression-bomb guard


def get_image_dimensions(image_path):
    with Image.open(image_path) as im:
        return im.size


_DEEP_GRAY = ("I;16", "I;16L", "I;16B", "I;16N", "I")      # Pillow modes of 16-bit (and wider) single-channel files


# ---- plain 8-bit PNG scans: zlib + un-filtering in C (csrc/host_png.c), everything else through Pillow -------------------------
_HOST_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libasep_host.so")
_host = None


def _host_lib():
    """libasep_host.so (include/asep_host.h), or False if it has not been built: the decode then stays with Pillow"""
    global _host
    if _host is None:
        try:
            lib = C.CDLL(_HOST_LIB)
            lib.asep_png_unfilter.restype = C.c_long
            lib.asep_png_unfilter.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_void_p]
            lib.asep_rgb_to_bgr.restype = None
            lib.asep_rgb_to_bgr.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
            _host = lib
        except OSError:
            _host = False
    return _host


_deflate = None
_deflate_state = threading.local()


def _inflate(stream, nbytes):
    """zlib stream -> uint8 array of exactly ``nbytes``, valid until the calling thread's next call (None if the stream is damaged
    or has another length): libdeflate when the system
    has it (libdeflate.so.0 ships with this image; about twice the speed of zlib's inflate on scan data), else zlib"""
    global _deflate
    if _deflate is None:
        try:
            lib = C.CDLL("libdeflate.so.0")
            lib.libdeflate_alloc_decompressor.restype = C.c_void_p
            lib.libdeflate_zlib_decompress.restype = C.c_int
            lib.libdeflate_zlib_decompress.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                                       C.POINTER(C.c_size_t)]
            _deflate = lib
        except (OSError, AttributeError):
            _deflate = False
    if _deflate:
        d = getattr(_deflate_state, "d", None)
        if d is None:
            d = _deflate_state.d = _deflate.libdeflate_alloc_decompressor()     # one per thread, kept for the process's life
        if d:
            out = getattr(_deflate_state, "buf", None)       # scratch of the calling thread, reused from scan to scan (the
            if out is None or out.size != nbytes:            # caller un-filters out of it at once and keeps no reference)
                out = _deflate_state.buf = np.empty(nbytes, dtype=np.uint8)
            got = C.c_size_t(0)
            rc = _deflate.libdeflate_zlib_decompress(d, stream, len(stream), out.ctypes.data, nbytes, C.byref(got))
            if rc == 0 and got.value == nbytes:
                return out
            if rc in (1, 2):                                # LIBDEFLATE_BAD_DATA, SHORT_OUTPUT: not the image the header announces
                return None
            # (3 = INSUFFICIENT_SPACE: more data than announced -- zlib below agrees or refuses)
    try:
        data = zlib.decompress(stream)
    except zlib.error:
        return None
    return np.frombuffer(data, dtype=np.uint8) if len(data) == nbytes else None


_PNG_SIGNATURE = b"\x89PNG\r\n\x1a\n"
# ancillary chunks that change what cv2.imread / Pillow hand out (transparency, orientation): such files go through Pillow
_PNG_NOT_PLAIN = (b"tRNS", b"eXIf", b"PLTE", b"acTL")
# text chunks whose keyword starts like this carry an EXIF / XMP profile (ImageMagick's "Raw profile type exif", Adobe's XMP packet)
_PNG_TEXT_NOT_PLAIN = (b"Raw profile type", b"XML:com.adobe.xmp")


def _load_png_plain(path):
    """8-bit gray or RGB, non-interlaced PNG -> uint8 [H,W] or [H,W,3] BGR, exactly what Pillow decodes; None for every other
    flavour (palette, alpha, 16 bit, interlaced, transparency / EXIF chunks, damaged files -- Pillow reports those)."""
    lib = _host_lib()
    if not lib:
        return None
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:8] != _PNG_SIGNATURE or len(raw) < 33 or raw[12:16] != b"IHDR":
        return None
    if zlib.crc32(raw[12:29]) & 0xffffffff != struct.unpack(">I", raw[29:33])[0]:
        return None                                         # damaged header: Pillow reports it (the pixel data carries zlib's own
                                                            # Adler-32, which the inflate step verifies)
    width, height, depth, colour, compression, filt, interlace = struct.unpack(">IIBBBBB", raw[16:29])
    if depth != 8 or colour not in (0, 2) or compression or filt or interlace or not width or not height:
        return None
    # a header may announce any size: Pillow's decompression-bomb limit applies here too (beyond it Pillow itself reports the file),
    # and a deflate stream expands at most ~1032x, so a size the file cannot possibly hold is a damaged or crafted header
    bpp = 1 if colour == 0 else 3
    stride = width * bpp
    limit = Image.MAX_IMAGE_PIXELS
    if (limit is not None and width * height > limit) or height * (stride + 1) > 1040 * len(raw):
        return None
    pos, parts = 8, []
    while pos + 12 <= len(raw):
        length, kind = struct.unpack(">I4s", raw[pos:pos + 8])
        if kind in _PNG_NOT_PLAIN:
            return None
        if kind in (b"tEXt", b"zTXt", b"iTXt") and raw[pos + 8:pos + 8 + length].startswith(_PNG_TEXT_NOT_PLAIN):
            return None                                     # orientation / metadata profiles in text chunks: Pillow applies them
        if kind == b"IDAT":
            if zlib.crc32(raw[pos + 4:pos + 8 + length]) & 0xffffffff != struct.unpack(">I", raw[pos + 8 + length:pos + 12 + length] or b"\0\0\0\0")[0]:
                return None                                 # damaged chunk: Pillow reports it
            parts.append(raw[pos + 8:pos + 8 + length])
        elif kind == b"IEND":
            break
        pos += 12 + length
    if not parts:
        return None
    try:
        data = _inflate(parts[0] if len(parts) == 1 else b"".join(parts), height * (stride + 1))
        if data is None:
            return None
        out = np.empty((height, width) if bpp == 1 else (height, width, 3), dtype=np.uint8)
        if bpp == 1:
            rc = lib.asep_png_unfilter(data.ctypes.data, height, stride, 1, out.ctypes.data)
        else:
            rgb = np.empty((height, width, 3), dtype=np.uint8)
            rc = lib.asep_png_unfilter(data.ctypes.data, height, stride, 3, rgb.ctypes.data)
            if rc == 0:
                lib.asep_rgb_to_bgr(rgb.ctypes.data, height * width, out.ctypes.data)
    except MemoryError:                                     # let Pillow report what it makes of the file
        return None
    return out if rc == 0 else None


def load_image_bgr(path_to_image):
    """uint8 [H,W,3] in BGR order, or uint8 [H,W] for single-channel files.

    Like ``cv2.imread`` with its default flags: the EXIF orientation is applied, an alpha channel is dropped, palette
    files are expanded, and 16-bit samples are reduced to their high byte (libpng's ``strip_16``; Pillow's own
    ``convert('L')`` would clip everything above 255 to white instead)."""
    if str(path_to_image).lower().endswith(".png"):
        plain = _load_png_plain(path_to_image)
        if plain is not None:
            return plain
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
