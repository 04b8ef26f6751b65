"""GPU parity of the classical image stages (C ABI block "classical image stages") against oracle/classical_oracle.py.

All of this is integer / byte work: results must be bit-identical to the oracle (the float sequences of the
INTER_AREA path and the double sequence of Otsu are evaluated without FMA contraction on both sides)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _text_page(rng, H, W):
    """light paper, dark glyph boxes, a few rules; uint8 gray"""
    g = np.clip(rng.normal(225, 6, size=(H, W)), 0, 255)
    y = 10
    while y < H - 30:
        lh = int(rng.integers(8, 24))
        x = 8
        while x < W - 30:
            gw = int(rng.integers(3, 14))
            if rng.random() < 0.8:
                g[y:y + lh, x:x + gw] = np.clip(rng.normal(60, 25, size=(min(lh, H - y), min(gw, W - x))), 0, 255)
            x += gw + int(rng.integers(2, 8))
        y += lh + int(rng.integers(6, 14))
    return g.astype(np.uint8)


def _separator_mask(rng, H, W, noise=0.002):
    m = np.zeros((H, W), np.uint8)
    for _ in range(6):
        y, x0, x1 = int(rng.integers(0, H - 4)), int(rng.integers(0, W // 2)), int(rng.integers(W // 2, W))
        m[y:y + int(rng.integers(2, 5)), x0:x1] = 255
    for _ in range(6):
        x, y0, y1 = int(rng.integers(0, W - 4)), int(rng.integers(0, H // 2)), int(rng.integers(H // 2, H))
        m[y0:y1, x:x + int(rng.integers(2, 5))] = 255
    m[rng.random((H, W)) < noise] = 255
    for _ in range(10):                                     # blobs around the CC-size threshold
        y, x = int(rng.integers(0, H - 12)), int(rng.integers(0, W - 12))
        m[y:y + int(rng.integers(8, 12)), x:x + int(rng.integers(8, 12))] = 255
    return m


@pytest.mark.parametrize("H,W,C,sc", [
    (90, 120, 3, 1 / 3), (91, 122, 3, 1 / 3), (64, 80, 3, 0.5), (63, 81, 1, 0.5), (100, 140, 3, 0.37),
    (450, 300, 3, 150 / 451), (75, 50, 3, 1.5), (40, 33, 1, 2.25), (48, 64, 3, 1.0), (97, 131, 3, 0.9),
])
def test_scale_and_gray_bit_exact(H, W, C, sc):
    from citlab_article_separation_new_amd import image_ops
    from oracle import classical_oracle as co
    rng = np.random.default_rng(H * 1000 + W)
    img = rng.integers(0, 256, size=(H, W, C), dtype=np.uint8)
    ref_img, _ = co.scale_image(img if C == 3 else img[:, :, 0], None, sc)
    ref_gray_u8 = co.bgr2gray(ref_img) if C == 3 else ref_img
    got_img, got_gray, got_sc = image_ops.scale_and_gray(img if C == 3 else img[:, :, 0], None, sc)
    assert got_sc == sc
    assert got_img.shape == ref_img.shape
    assert np.array_equal(got_img, ref_img)
    assert np.array_equal(got_gray, (ref_gray_u8 / 255.0).astype(np.float32))


def test_scale_with_fixed_height_like_the_separator_default():
    from citlab_article_separation_new_amd import image_ops
    from oracle import classical_oracle as co
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, size=(900, 601, 3), dtype=np.uint8)
    ref_img, ref_gray, ref_sc = co.scale_and_gray(img, 300, 1.0)
    got_img, got_gray, sc = image_ops.scale_and_gray(img, 300, 1.0)
    assert sc == ref_sc and got_img.shape[0] == 300
    assert np.array_equal(got_img, ref_img)
    assert np.array_equal(got_gray, ref_gray.astype(np.float32))


@pytest.mark.parametrize("H,W,density,min_size", [(1, 1, 1.0, 1), (5, 7, 0.5, 2), (64, 64, 0.45, 10),
                                                  (200, 333, 0.55, 50), (300, 257, 0.62, 500), (17, 1000, 0.5, 4)])
def test_cc_filter_random_masks(H, W, density, min_size):
    from citlab_article_separation_new_amd import image_ops
    from oracle import classical_oracle as co
    rng = np.random.default_rng(H + W)
    m = ((rng.random((H, W)) < density) * rng.integers(1, 256, size=(H, W))).astype(np.uint8)
    got = image_ops.apply_cc_analysis(m, min_size / m.size * (1 + 1e-9))
    assert np.array_equal(got, co.cc_filter(m, int(m.size * (min_size / m.size * (1 + 1e-9)))))


def test_cc_filter_snake_and_full_image():
    from citlab_article_separation_new_amd import image_ops
    from oracle import classical_oracle as co
    H, W = 129, 200
    m = np.zeros((H, W), np.uint8)
    for y in range(0, H, 2):                                # one long serpentine component
        m[y, :] = 255
        if y + 1 < H:
            m[y + 1, (W - 1) if (y // 2) % 2 == 0 else 0] = 255
    m[5, 7] = 255
    for ms in (1, 1000, 10 ** 6):
        assert np.array_equal(image_ops.apply_cc_analysis(m, ms / m.size * (1 + 1e-9)), co.cc_filter(m, ms))
    full = np.full((70, 130), 255, np.uint8)
    assert np.array_equal(image_ops.apply_cc_analysis(full, 0.5), full)
    assert image_ops.apply_cc_analysis(np.zeros((70, 130), np.uint8), 0.0).sum() == 0


@pytest.mark.parametrize("op", [0, 1, 2, 3])
@pytest.mark.parametrize("kw,kh", [(1, 1), (3, 1), (4, 1), (45, 1), (70, 1), (130, 1), (1, 5), (1, 6), (1, 90),
                                   (7, 4), (64, 3)])
def test_morphology_rect(op, kw, kh):
    from citlab_article_separation_new_amd import image_ops
    from oracle import classical_oracle as co
    rng = np.random.default_rng(kw * 100 + kh)
    H, W = 150, 331
    p = 0.97 if op in (0, 2) else 0.03
    m = ((rng.random((H, W)) < p) * 255).astype(np.uint8)
    m[40:44, 10:300] = 255 if op in (0, 2) else 0
    m[5:140, 100:103] = 255 if op in (0, 2) else 0
    ref = {0: co.erode_rect, 1: co.dilate_rect, 2: co.open_rect,
           3: lambda a, w, h: co.erode_rect(co.dilate_rect(a, w, h), w, h)}[op](m, kw, kh)
    assert np.array_equal(image_ops.morphology_rect(m, op, (kw, kh)), ref)


def test_morphology_rejects_empty_kernel():
    from citlab_article_separation_new_amd import image_ops, _lib
    with pytest.raises(_lib.AsepError):
        image_ops.morphology_rect(np.zeros((8, 8), np.uint8), 2, (0, 1))


@pytest.mark.parametrize("H,W", [(300, 400), (768, 512), (1500, 1000)])
def test_separator_post_process(H, W):
    from citlab_article_separation_new_amd import image_ops
    from oracle import classical_oracle as co
    rng = np.random.default_rng(H)
    m = np.zeros((H, W, 2), np.uint8)
    m[:, :, 0] = _separator_mask(rng, H, W)
    m[:, :, 1] = 255 - m[:, :, 0]
    got = image_ops.separator_post_process(m)
    ref = co.separator_post_process(m)
    assert np.array_equal(got["horizontal"], ref["horizontal"])
    assert np.array_equal(got["vertical"], ref["vertical"])
    assert ref["horizontal"].any() and ref["vertical"].any()


def test_separator_post_process_full_page_and_tiny_page():
    from citlab_article_separation_new_amd import image_ops, _lib
    from oracle import classical_oracle as co
    rng = np.random.default_rng(4500)
    H, W = 4500, 3000
    m = _separator_mask(rng, H, W, noise=0.0005)[:, :, None]
    got = image_ops.separator_post_process(m)
    ref = co.separator_post_process(m)
    assert np.array_equal(got["horizontal"], ref["horizontal"])
    assert np.array_equal(got["vertical"], ref["vertical"])
    # size-independent properties: an opening with an odd (symmetric) kernel is idempotent and anti-extensive;
    # the even kernels of this page size (k_v = 90) shift by one pixel per application (OpenCV anchor rule)
    once = image_ops.morphology_rect(m[:, :, 0], 2, (1, 91))
    assert np.array_equal(image_ops.morphology_rect(once, 2, (1, 91)), once)
    assert np.all(m[:, :, 0][once > 0] > 0)
    with pytest.raises(_lib.AsepError):                      # W < 67: int(15*W/1000) == 0, cv2 would assert
        image_ops.separator_post_process(np.zeros((100, 60, 1), np.uint8))


@pytest.mark.parametrize("H,W,seed", [(40, 60, 0), (257, 300, 1), (700, 513, 2)])
def test_swt_distance_transform(H, W, seed):
    from citlab_article_separation_new_amd import image_ops
    from oracle import classical_oracle as co
    rng = np.random.default_rng(seed)
    g = _text_page(rng, H, W)
    if seed == 2:
        g[100:400, 50:400] = 15                              # a large dark block: distances beyond 127
    out, thr, d2 = image_ops.swt_distance_transform(g, return_details=True)
    inv = (255 - g.astype(np.int64)).astype(np.uint8)
    blur = co.gaussian5(inv)
    assert thr == co.otsu_threshold(blur)
    binary = ((blur > thr) * 255).astype(np.uint8)
    assert np.array_equal(d2.astype(np.int64), co.edt_sq(binary))
    assert np.array_equal(out, co.swt_distance_transform(g))
    if seed == 2:
        assert out.max() > 127


def test_swt_degenerate_images():
    from citlab_article_separation_new_amd import image_ops
    from oracle import classical_oracle as co
    for g in (np.full((33, 47), 200, np.uint8), np.zeros((20, 70), np.uint8)):
        assert np.array_equal(image_ops.swt_distance_transform(g), co.swt_distance_transform(g))
    g = np.full((50, 50), 255, np.uint8)
    g[:, :25] = 0                                            # left half dark: distances grow to 25
    assert np.array_equal(image_ops.swt_distance_transform(g), co.swt_distance_transform(g))


@pytest.mark.parametrize("H,W,density", [(1, 1, 1.0), (37, 130, 0.3), (64, 64, 0.6), (200, 333, 0.9), (90, 70, 0.5)])
def test_boundary_segments_give_the_same_polygons(H, W, density):
    from citlab_article_separation_new_amd import image_ops, polygonize
    rng = np.random.default_rng(H * 7 + W)
    m = ((rng.random((H, W)) < density) * 255).astype(np.uint8)
    m[rng.random((H, W)) < 0.05] = 77                       # other values are background for value == 255
    if H == 90:
        m[10:60, 5:60] = 255                                # a big blob with holes and islands inside
        m[20:50, 15:50][rng.random((30, 35)) < 0.3] = 0
    starts, ends = image_ops.boundary_segments(m, 255)
    assert starts.dtype == np.int32 and starts.size == ends.size
    assert polygonize.shapes_from_segments(starts, ends, H, W) == polygonize.shapes(m)
    assert polygonize.shapes_from_segments(starts, ends, H, W, connectivity=4) == polygonize.shapes(m, connectivity=4)


def test_boundary_segments_page_sized_and_capacity_retry():
    from citlab_article_separation_new_amd import image_ops, polygonize
    rng = np.random.default_rng(3)
    m = _separator_mask(rng, 1500, 1000, noise=0.01)         # ~60k one-pixel specks: exceeds the first capacity guess
    starts, ends = image_ops.boundary_segments(m, 255)
    assert starts.size > (1 << 14)
    polys = polygonize.shapes_from_segments(starts, ends, 1500, 1000)
    assert np.array_equal(polygonize.rasterize(polys, 1500, 1000), m)
    assert polys == polygonize.shapes(m)
    s0, e0 = image_ops.boundary_segments(np.zeros((50, 60), np.uint8))
    assert s0.size == 0 and polygonize.shapes_from_segments(s0, e0, 50, 60) == []


def test_boundary_segments_enqueue_form_equals_the_synchronous_one():
    """asep_post_boundary_segments_enqueue_dev leaves counts and keys on the device without a host round trip: two masks
    queued back to back give the key sets of the synchronous entry point; a capacity that is too small is reported by the
    counts (which may exceed it) and never written past"""
    import ctypes as C
    import torch
    from citlab_article_separation_new_amd import _lib, image_ops
    rng = np.random.default_rng(11)
    masks = [_separator_mask(rng, 300, 200, noise=0.002), _separator_mask(rng, 300, 200, noise=0.02)]
    lib, ws = image_ops._workspace(0)
    dev = torch.device("cuda", 0)
    sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    cap = 1 << 10
    d_masks = [torch.from_numpy(m).to(dev) for m in masks]
    d_keys = torch.full((2, 2, cap + 8), -7, dtype=torch.int32, device=dev)
    d_tot = torch.empty((2, 2), dtype=torch.int64, device=dev)
    for i, d_m in enumerate(d_masks):
        _lib.check(lib.asep_post_boundary_segments_enqueue_dev(ws, d_m.data_ptr(), 300, 200, 255, d_keys[i, 0].data_ptr(),
                                                               d_keys[i, 1].data_ptr(), cap, d_tot[i].data_ptr(), sp), "enqueue")
    tot, keys = d_tot.cpu().numpy(), d_keys.cpu().numpy()
    assert (keys[:, :, cap:] == -7).all()                    # nothing beyond the capacity
    for i, m in enumerate(masks):
        starts, ends = image_ops.boundary_segments(m, 255)
        assert tot[i, 0] == tot[i, 1] == starts.size
        if starts.size <= cap:
            assert np.array_equal(np.sort(keys[i, 0, :starts.size]), np.sort(starts))
            assert np.array_equal(np.sort(keys[i, 1, :ends.size]), np.sort(ends))
    assert tot[0, 0] <= cap < tot[1, 0]                      # both branches were exercised
    assert lib.asep_post_boundary_segments_enqueue_dev(ws, d_masks[0].data_ptr(), 300, 200, 255, None, None, cap, None, sp) < 0


@pytest.mark.parametrize("H,W", [(1, 1), (7, 5), (64, 64), (333, 257), (900, 601)])
def test_gray_u8_on_the_device_equals_the_fixed_point_formula(H, W):
    """asep_prep_gray_u8_dev = cv2.imread(path, IMREAD_GRAYSCALE) of a decoded BGR image: (B 3735 + G 19235 + R 9798 + 2^14) >> 15,
    also for pixel counts that are not a multiple of the four a thread converts and for saturated colours"""
    import ctypes as C
    import torch
    from citlab_article_separation_new_amd import _lib, image_ops
    from citlab_article_separation_new_amd.heading_net_post_processor import bgr_to_gray_u8
    rng = np.random.default_rng(H * 31 + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    img.reshape(-1, 3)[: min(8, H * W)] = [[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [254, 255, 255],
                                           [1, 1, 1], [128, 127, 129]][: min(8, H * W)]
    lib, ws = image_ops._workspace(0)
    dev = torch.device("cuda", 0)
    sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    d_img = torch.from_numpy(img).to(dev)
    d_out = torch.full((H * W + 16,), 201, dtype=torch.uint8, device=dev)
    _lib.check(lib.asep_prep_gray_u8_dev(ws, d_img.data_ptr(), H, W, d_out.data_ptr(), sp), "asep_prep_gray_u8_dev")
    out = d_out.cpu().numpy()
    assert np.array_equal(out[: H * W].reshape(H, W), bgr_to_gray_u8(img))
    assert (out[H * W:] == 201).all()
    assert lib.asep_prep_gray_u8_dev(ws, None, H, W, d_out.data_ptr(), sp) < 0


def test_box_sums_on_the_device_are_the_numpy_slice_sums():
    """asep_post_box_sums_dev: exact integer sums of img[y0:y1, x0:x1, channel], boxes clipped like numpy slices with
    non-negative bounds (beyond the image, empty, one pixel, the whole image), any channel of an interleaved image"""
    import torch
    from citlab_article_separation_new_amd import _lib, image_ops
    rng = np.random.default_rng(5)
    H, W = 700, 530
    for C_ in (1, 2, 3):
        img = rng.integers(0, 256, (H, W, C_), dtype=np.uint8)
        img[100:400, 50:500] = 255                              # a saturated block: 135000 x 255 needs more than 24 bits
        boxes = [[0, 0, W, H], [50, 100, 500, 400], [3, 4, 4, 5], [10, 10, 10, 40], [500, 650, 900, 900], [W, H, W + 5, H + 5],
                 [20, 30, 10, 20]]
        boxes += [[int(x), int(y), int(x + w), int(y + h)] for x, y, w, h in
                  zip(rng.integers(0, W, 40), rng.integers(0, H, 40), rng.integers(1, 300, 40), rng.integers(1, 80, 40))]
        d_img = torch.from_numpy(img).to("cuda:0")
        for ch in range(C_):
            got = image_ops.box_sums_dev(d_img.data_ptr(), img.shape, boxes, channel=ch)
            want = [int(img[y0:max(y0, y1), x0:max(x0, x1), ch].astype(np.int64).sum()) for x0, y0, x1, y1 in boxes]
            assert got.dtype == np.int64 and got.tolist() == want
    assert image_ops.box_sums_dev(d_img.data_ptr(), img.shape, []).size == 0
    with pytest.raises(_lib.AsepError):
        image_ops.box_sums_dev(d_img.data_ptr(), img.shape, boxes, channel=3)


def _line_boxes(rng, H, W, n):
    boxes = []
    for _ in range(n):
        w, h = int(rng.integers(20, min(400, W))), int(rng.integers(8, 60))
        x, y = int(rng.integers(0, W - w)), int(rng.integers(0, H - h))
        boxes.append([x, y, x + w + 1, y + h + 1])
    return boxes


def test_swt_line_features_match_reference_arithmetic():
    from citlab_article_separation_new_amd import image_ops
    from oracle import classical_oracle as co
    rng = np.random.default_rng(11)
    H, W = 600, 700
    g = _text_page(rng, H, W)
    swt = co.swt_distance_transform(g)
    boxes = _line_boxes(rng, H, W, 60)
    boxes += [[0, 0, W, H], [W - 5, H - 5, W + 50, H + 50], [10, 10, 10, 40], [300, 300, 290, 310], [5, 5, 8, 8]]
    sw, hh = image_ops.swt_line_features(swt, boxes)
    for i, (x0, y0, x1, y1) in enumerate(boxes):
        w, h = x1 - x0 - 1, y1 - y0 - 1                          # oracle takes (x, y, width, height) and adds 1
        if w < 0 or h < 0:
            exp = (0.0, 0)
        else:
            exp = co.swt_features_textline(swt, (x0, y0, w, h))
        assert (sw[i], hh[i]) == exp, (i, boxes[i], (sw[i], hh[i]), exp)
    assert (sw > 0).sum() > 30 and (hh > 0).sum() > 30


def test_swt_line_features_overflow_falls_back_to_the_host():
    from citlab_article_separation_new_amd import image_ops
    from oracle import classical_oracle as co
    swt = np.zeros((200, 900), np.uint8)
    swt[::4, ::4] = 3                                           # 50 x 225 isolated pixels: > 1024 components
    swt[50:70, 100:110] = 7
    swt[50:53, 100] = 0
    sw, hh = image_ops.swt_line_features(swt, [[0, 0, 900, 200], [90, 40, 130, 80]])
    assert (sw[0], hh[0]) == co.swt_features_textline(swt, (0, 0, 899, 199))
    assert (sw[1], hh[1]) == co.swt_features_textline(swt, (90, 40, 39, 39))


def _spiral(n):
    m = np.zeros((n, n), np.uint8)
    x0, y0, x1, y1 = 0, 0, n - 1, n - 1
    while x0 <= x1 and y0 <= y1:
        m[y0, x0:x1 + 1] = 255
        m[y0:y1 + 1, x1] = 255
        if y1 > y0:
            m[y1, x0 + 2:x1 + 1] = 255
        if x1 > x0 + 2 and y1 > y0 + 2:
            m[y0 + 2:y1 + 1, x0 + 2] = 255
        x0, y0, x1, y1 = x0 + 4, y0 + 4, x1 - 4, y1 - 4    # loosely nested rings joined at one corner each turn
        if x0 <= x1:
            m[y0 - 2, x0 - 2:x0 + 1] = 255
            m[y0 - 2:y0 + 1, x0] = 255
    return m


@pytest.mark.parametrize("name", ["spiral", "diagonals", "comb", "checker", "staircase"])
def test_cc_filter_adversarial_shapes(name):
    """Shapes that stress the lock-free union-find: long dependency chains, components that only connect through
    diagonal pixels, thousands of runs merging into one root, and many tiny components."""
    from citlab_article_separation_new_amd import image_ops
    from oracle import classical_oracle as co
    n = 257
    if name == "spiral":
        m = _spiral(n)
    elif name == "diagonals":
        m = np.zeros((n, n), np.uint8)
        idx = np.arange(n)
        m[idx, idx] = 255
        m[idx, (n - 1 - idx)] = 255
        m[idx[:-3], idx[:-3] + 3] = 255
    elif name == "comb":
        m = np.zeros((n, n), np.uint8)
        m[:, ::2] = 255                                   # 129 vertical teeth ...
        m[n - 1, :] = 255                                 # ... joined only by the last row
    elif name == "checker":
        m = (np.indices((n, n)).sum(axis=0) % 2 * 255).astype(np.uint8)   # one diagonal-connected component
    else:
        m = np.zeros((n, n), np.uint8)
        for k in range(0, n - 2, 2):
            m[k, k:k + 2] = 255
            m[k + 1, k + 1:k + 3] = 255
    for min_size in (1, 50, 5000, n * n):
        got = image_ops.apply_cc_analysis(m, min_size / m.size * (1 + 1e-9))
        assert np.array_equal(got, co.cc_filter(m, min_size)), (name, min_size)


def test_swt_line_features_on_adversarial_crop():
    from citlab_article_separation_new_amd import image_ops
    from oracle import classical_oracle as co
    swt = (_spiral(200) // 255 * 5).astype(np.uint8)
    swt[::7, ::5] = 9
    boxes = [[0, 0, 200, 200], [10, 10, 150, 60], [100, 0, 200, 200]]
    sw, hh = image_ops.swt_line_features(swt, boxes)
    for i, (x0, y0, x1, y1) in enumerate(boxes):
        assert (sw[i], hh[i]) == co.swt_features_textline(swt, (x0, y0, x1 - x0 - 1, y1 - y0 - 1))
