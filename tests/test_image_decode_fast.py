"""image_io's plain-PNG path (zlib + csrc/host_png.c un-filtering through include/asep_host.h) against Pillow: bit-identical pixels
for 8-bit gray / RGB files of every row-filter mix, and a clean hand-over to Pillow for every other PNG flavour."""
import io
import struct
import zlib

import numpy as np
import pytest
from PIL import Image

from citlab_article_separation_new_amd import image_io


def _png(width, height, colour, rows_with_filter):
    """a PNG written by hand: rows_with_filter = [(filter type, filtered bytes)], one IDAT per ~1000 bytes"""
    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xffffffff)
    raw = b"".join(bytes([ft]) + bytes(row) for ft, row in rows_with_filter)
    comp = zlib.compress(raw, 6)
    idats = b"".join(chunk(b"IDAT", comp[i:i + 1000]) for i in range(0, len(comp), 1000))
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, 8, colour, 0, 0, 0)) + idats + chunk(b"IEND", b"")


def _filter_rows(img, bpp, types):
    """forward PNG filters (specification 9.2) of an image [H, W*bpp] uint8 with the given filter type per row"""
    H, S = img.shape
    out = []
    prev = np.zeros(S, np.int32)
    for y in range(H):
        cur = img[y].astype(np.int32)
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if S > bpp else np.zeros(S, np.int32)
        upleft = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]]) if S > bpp else np.zeros(S, np.int32)
        ft = types[y]
        if ft == 0:
            f = cur
        elif ft == 1:
            f = cur - left
        elif ft == 2:
            f = cur - prev
        elif ft == 3:
            f = cur - ((left + prev) >> 1)
        else:
            p = left + prev - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
            f = cur - pred
        out.append((ft, (f & 255).astype(np.uint8)))
        prev = cur
    return out


@pytest.mark.skipif(not image_io._host_lib(), reason="libasep_host.so not built")
@pytest.mark.parametrize("W,H,bpp", [(1, 1, 1), (2, 3, 1), (3, 9, 1), (4, 8, 1), (61, 37, 1), (300, 257, 1), (1, 5, 3), (2, 2, 3),
                                     (57, 41, 3), (128, 130, 3)])
@pytest.mark.parametrize("mix", ["paeth", "all", "runs"])
def test_hand_written_png_of_every_filter_mix_decodes_like_pillow(tmp_path, W, H, bpp, mix):
    rng = np.random.default_rng(W * 1000 + H * 7 + bpp)
    img = rng.integers(0, 256, (H, W * bpp), dtype=np.uint8)
    img[H // 2:, : (W * bpp) // 2] = 255                       # flat areas: ties in the Paeth predictor
    if mix == "paeth":
        types = [4] * H                                        # what Pillow's encoder writes: exercises the 4-row wavefront
    elif mix == "all":
        types = [int(t) for t in rng.integers(0, 5, H)]
    else:
        types = ([4] * 5 + [1] + [4] * 3 + [2, 0, 3] + [4] * 9)[:H] + [4] * max(0, H - 21)
    data = _png(W, H, 0 if bpp == 1 else 2, _filter_rows(img, bpp, types))
    p = tmp_path / "t.png"
    p.write_bytes(data)
    want = np.asarray(Image.open(io.BytesIO(data)))
    assert np.array_equal(want.reshape(H, W * bpp), img)       # the hand-written file is a valid PNG of `img`
    got = image_io._load_png_plain(str(p))
    assert got is not None and got.dtype == np.uint8
    assert np.array_equal(got, want if bpp == 1 else want[:, :, ::-1])
    assert np.array_equal(image_io.load_image_bgr(str(p)), got)


@pytest.mark.skipif(not image_io._host_lib(), reason="libasep_host.so not built")
def test_pillow_written_scans_and_the_flavours_left_to_pillow(tmp_path):
    rng = np.random.default_rng(1)
    gray = rng.integers(0, 256, (211, 333), dtype=np.uint8)
    rgb = rng.integers(0, 256, (97, 101, 3), dtype=np.uint8)
    for k, (arr, level) in enumerate([(gray, 1), (gray, 9), (rgb, 1), (rgb, 6)]):
        p = tmp_path / f"w{k}.png"
        Image.fromarray(arr).save(p, compress_level=level)
        got = image_io._load_png_plain(str(p))
        assert got is not None and np.array_equal(got, arr if arr.ndim == 2 else arr[:, :, ::-1])
    # everything else is None here and decoded by Pillow: palette, alpha, 16 bit, 1 bit, transparency chunk, not a PNG, damaged
    others = {
        "pal.png": Image.fromarray(gray).convert("P"),
        "rgba.png": Image.fromarray(np.dstack([rgb, rgb[:, :, 0]])),
        "la.png": Image.fromarray(gray).convert("LA"),
        "deep.png": Image.fromarray((gray.astype(np.uint16) << 8) | 3),
        "bilevel.png": Image.fromarray(gray > 128),
    }
    for name, im in others.items():
        im.save(tmp_path / name)
        assert image_io._load_png_plain(str(tmp_path / name)) is None, name
        assert image_io.load_image_bgr(str(tmp_path / name)).dtype == np.uint8
    Image.fromarray(rgb).save(tmp_path / "trns.png", transparency=(1, 2, 3))
    assert image_io._load_png_plain(str(tmp_path / "trns.png")) is None
    (tmp_path / "no.png").write_bytes(b"\xff\xd8\xff not a png")
    assert image_io._load_png_plain(str(tmp_path / "no.png")) is None
    good = (tmp_path / "w0.png").read_bytes()
    (tmp_path / "cut.png").write_bytes(good[: len(good) // 2])
    assert image_io._load_png_plain(str(tmp_path / "cut.png")) is None
    bad_filter = _png(4, 2, 0, [(0, [1, 2, 3, 4]), (7, [1, 2, 3, 4])])
    (tmp_path / "f7.png").write_bytes(bad_filter)
    assert image_io._load_png_plain(str(tmp_path / "f7.png")) is None


@pytest.mark.skipif(not image_io._host_lib(), reason="libasep_host.so not built")
@pytest.mark.parametrize("use_libdeflate", [True, False])
def test_both_inflate_paths_give_the_same_pixels(tmp_path, monkeypatch, use_libdeflate):
    """the zlib stream goes through libdeflate when the system has it and through zlib otherwise: same pixels, same refusals"""
    if use_libdeflate:
        image_io._inflate(zlib.compress(b"x"), 1)
        if not image_io._deflate:
            pytest.skip("no libdeflate.so.0 on this system")
    else:
        monkeypatch.setattr(image_io, "_deflate", False)
    rng = np.random.default_rng(9)
    arr = rng.integers(0, 256, (301, 203), dtype=np.uint8)
    Image.fromarray(arr).save(tmp_path / "a.png", compress_level=3)
    assert np.array_equal(image_io._load_png_plain(str(tmp_path / "a.png")), arr)
    good = (tmp_path / "a.png").read_bytes()
    (tmp_path / "cut.png").write_bytes(good[: len(good) * 2 // 3])
    assert image_io._load_png_plain(str(tmp_path / "cut.png")) is None
    assert image_io._inflate(zlib.compress(b"abcdef"), 6).tobytes() == b"abcdef"
    assert image_io._inflate(zlib.compress(b"abcdef"), 5) is None and image_io._inflate(zlib.compress(b"abcdef"), 7) is None
    assert image_io._inflate(b"not a zlib stream", 4) is None


@pytest.mark.skipif(not image_io._host_lib(), reason="libasep_host.so not built")
def test_damaged_header_or_pixel_data_is_left_to_pillow(tmp_path):
    """a flipped byte in IHDR (chunk CRC) or in the compressed pixel data (zlib's Adler-32) is not decoded silently"""
    arr = np.random.default_rng(2).integers(0, 256, (64, 80), dtype=np.uint8)
    Image.fromarray(arr).save(tmp_path / "ok.png")
    good = bytearray((tmp_path / "ok.png").read_bytes())
    assert np.array_equal(image_io._load_png_plain(str(tmp_path / "ok.png")), arr)
    bad = bytearray(good)
    bad[19] ^= 1                                            # width
    (tmp_path / "hdr.png").write_bytes(bytes(bad))
    assert image_io._load_png_plain(str(tmp_path / "hdr.png")) is None
    bad = bytearray(good)
    i = bytes(good).index(b"IDAT") + 4 + 40
    bad[i] ^= 0x10
    (tmp_path / "pix.png").write_bytes(bytes(bad))
    assert image_io._load_png_plain(str(tmp_path / "pix.png")) is None


@pytest.mark.skipif(not image_io._host_lib(), reason="libasep_host.so not built")
def test_crafted_sizes_and_text_profiles_are_left_to_pillow(tmp_path, monkeypatch):
    """the fast path trusted IHDR's width x height and allocated before anything was inflated (a tiny crafted file -> a
    MemoryError or an OOM-killed decode worker instead of Pillow's report), and it ignored orientation profiles carried in text
    chunks, so the pixels depended on whether libasep_host.so was built."""
    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xffffffff)
    arr = np.random.default_rng(3).integers(0, 256, (20, 30), dtype=np.uint8)
    rows = _filter_rows(arr, 1, [0] * 20)
    good = _png(30, 20, 0, rows)
    (tmp_path / "ok.png").write_bytes(good)
    assert np.array_equal(image_io._load_png_plain(str(tmp_path / "ok.png")), arr)
    # (a) a header that announces 60000 x 60000 pixels in a file of a few hundred bytes (valid CRC): refused before any allocation
    huge = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 60000, 60000, 8, 0, 0, 0, 0)) + good[33:]
    (tmp_path / "huge.png").write_bytes(huge)
    called = []
    monkeypatch.setattr(image_io, "_inflate", lambda *a: called.append(a) or None)
    assert image_io._load_png_plain(str(tmp_path / "huge.png")) is None and not called
    monkeypatch.undo()
    # (b) beyond Pillow's MAX_IMAGE_PIXELS the file is Pillow's to report
    monkeypatch.setattr(Image, "MAX_IMAGE_PIXELS", 500)
    assert image_io._load_png_plain(str(tmp_path / "ok.png")) is None
    monkeypatch.undo()
    # (c) an allocation failure falls back to Pillow instead of killing the worker
    def boom(*a):
        raise MemoryError
    monkeypatch.setattr(image_io, "_inflate", boom)
    assert image_io._load_png_plain(str(tmp_path / "ok.png")) is None
    monkeypatch.undo()
    # (d) EXIF / XMP profiles in text chunks (ImageMagick's "Raw profile type exif", Adobe's XMP packet): not plain
    for kind, payload in ((b"tEXt", b"Raw profile type exif\0\nexif\n  10\n0000\n"), (b"zTXt", b"Raw profile type APP1\0\0" + zlib.compress(b"x")),
                          (b"iTXt", b"XML:com.adobe.xmp\0\0\0\0\0<x:xmpmeta/>")):
        (tmp_path / "t.png").write_bytes(good[:33] + chunk(kind, payload) + good[33:])
        assert image_io._load_png_plain(str(tmp_path / "t.png")) is None, kind
    (tmp_path / "c.png").write_bytes(good[:33] + chunk(b"tEXt", b"Comment\0scanned 1887") + good[33:])      # a harmless comment stays plain
    assert np.array_equal(image_io._load_png_plain(str(tmp_path / "c.png")), arr)
    # (e) a damaged IDAT chunk CRC (the zlib stream itself intact) is reported by Pillow, not decoded silently
    bad = bytearray(good)
    i = bytes(good).index(b"IDAT")
    n = struct.unpack(">I", good[i - 4:i])[0]
    bad[i + 4 + n] ^= 0xff                                   # first byte of the chunk's CRC
    (tmp_path / "crc.png").write_bytes(bytes(bad))
    assert image_io._load_png_plain(str(tmp_path / "crc.png")) is None
