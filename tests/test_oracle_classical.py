"""Pins for oracle/classical_oracle.py (CPU only).

OpenCV / rasterio are not installed and the reference has no fixtures for these stages ("parity unpinned",
SURVEY.md section 8c): the oracle is pinned by hand-derived cases of the documented OpenCV semantics, by
definition-level brute force, and by scipy.ndimage where scipy implements the same operation."""
import numpy as np
import pytest
from scipy import ndimage

from oracle import classical_oracle as co


def test_bgr2gray_known_values():
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 30]]], np.uint8)
    g = co.bgr2gray(px)[0]
    # white stays white, pure channels follow the 15-bit coefficients B 3735, G 19235, R 9798
    assert g.tolist() == [255, 0, (255 * 3735 + 16384) >> 15, (255 * 19235 + 16384) >> 15,
                          (255 * 9798 + 16384) >> 15, (10 * 3735 + 200 * 19235 + 30 * 9798 + 16384) >> 15]


def test_cv_round_half_even():
    assert [co.cv_round(v) for v in (0.5, 1.5, 2.5, -0.5, 2.4999, 2.5001)] == [0, 2, 2, 0, 2, 3]


def test_resize_area_integer_scale_is_block_mean():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(12, 9, 3), dtype=np.uint8)
    out = co.resize_area(img, 1 / 3)
    assert out.shape == (4, 3, 3)
    ref = img.reshape(4, 3, 3, 3, 3).astype(np.float64).mean(axis=(1, 3))
    assert np.abs(out.astype(np.float64) - ref).max() <= 0.5 + 1e-6
    # 2x2 uses the integer form (s + 2) >> 2
    out2 = co.resize_area(img[:, :8], 0.5)
    s = img[:, :8].reshape(6, 2, 4, 2, 3).astype(np.int64).sum(axis=(1, 3))
    assert np.array_equal(out2, ((s + 2) >> 2).astype(np.uint8))


def test_resize_area_fractional_weights_sum_to_one_and_constant_image():
    for ssize, sc in ((100, 0.37), (4500, 1500 / 4501), (17, 0.9)):
        dsize = co.cv_round(ssize * sc)
        tab = co.area_table(ssize, dsize, 1.0 / sc)
        sums = np.zeros(dsize)
        for d, s, w in tab:
            assert 0 <= s < ssize
            sums[d] += w
        # OpenCV drops partial cells thinner than 1e-3 pixel, so a row may miss up to 1e-3/scale of weight
        assert np.allclose(sums, 1.0, atol=1.1e-3)
    img = np.full((40, 50), 173, np.uint8)
    assert np.all(co.resize_area(img, 0.37) == 173)
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(40, 50), dtype=np.uint8)
    out = co.resize_area(img, 0.4)            # scale 2.5: every output pixel averages a 2.5 x 2.5 window
    assert out.shape == (16, 20)
    # first pixel: window rows/cols 0,1 full and 2 half
    w = np.array([1, 1, 0.5]) / 2.5
    exp = (img[:3, :3].astype(np.float64) * np.outer(w, w)).sum()
    assert abs(float(out[0, 0]) - exp) <= 0.5 + 1e-4


def test_resize_cubic_constant_and_shape():
    img = np.full((10, 12), 99, np.uint8)
    out = co.resize_cubic(img, 1.5)
    assert out.shape == (15, 18) and np.all(out == 99)
    # the four taps always sum to 2048 +- rounding; A = -0.75 taps at fx = 0.5
    _, w = co.cubic_table(10, 20, 0.5)
    assert np.all(np.abs(w.sum(axis=1) - 2048) <= 2)
    c = co._cubic_coeffs(0.5)
    assert np.allclose(c, [-0.09375, 0.59375, 0.59375, -0.09375])


def test_cc_filter_hand_case():
    m = np.zeros((6, 8), np.uint8)
    m[0, 0] = 255
    m[1, 1] = 200                  # diagonal neighbour: 8-connected, any non-zero value is foreground
    m[3, 4:8] = 255                # area 4
    m[5, 0] = 1                    # area 1
    out = co.cc_filter(m, 2)
    exp = np.zeros_like(m)
    exp[0, 0] = exp[1, 1] = 255
    exp[3, 4:8] = 255
    assert np.array_equal(out, exp)
    assert np.array_equal(co.cc_filter(m, 3), np.where(np.arange(8)[None] >= 4, 255, 0).astype(np.uint8)
                          * (np.arange(6)[:, None] == 3))
    assert co.cc_filter(np.zeros((3, 3), np.uint8), 1).sum() == 0


def test_cc_min_size_expression():
    # base:244 with threshold 1/size*100: the double product may round below 100
    for size in (768 * 512, 1500 * 1000, 4500 * 3000, 900 * 600, 12345):
        assert co.cc_min_size(size, 1 / size * 100) in (99, 100)


@pytest.mark.parametrize("kw,kh", [(1, 1), (3, 1), (4, 1), (1, 5), (1, 6), (3, 2), (7, 4)])
def test_morphology_matches_definition(kw, kh):
    rng = np.random.default_rng(kw * 10 + kh)
    m = ((rng.random((13, 17)) < 0.7) * 255).astype(np.uint8)
    assert np.array_equal(co.erode_rect(m, kw, kh), co.erode_rect_bruteforce(m, kw, kh))
    assert np.array_equal(co.dilate_rect(m, kw, kh), co.erode_rect_bruteforce(m, kw, kh, dilate=True))


def test_morphology_odd_kernels_match_scipy():
    rng = np.random.default_rng(5)
    m = rng.random((40, 60)) < 0.8
    for kw, kh in ((5, 1), (1, 7), (3, 3)):
        st = np.ones((kh, kw), bool)
        assert np.array_equal(co.erode_rect(m * 255, kw, kh) > 0, ndimage.binary_erosion(m, st, border_value=1))
        assert np.array_equal(co.dilate_rect(m * 255, kw, kh) > 0, ndimage.binary_dilation(m, st, border_value=0))
        assert np.array_equal(co.open_rect(m * 255, kw, kh) > 0,
                              ndimage.binary_dilation(ndimage.binary_erosion(m, st, border_value=1), st))


def test_open_even_kernel_shifts_by_one_and_border_runs_survive():
    row = np.zeros((1, 30), np.uint8)
    row[0, 5:15] = 255                       # run of 10
    out = co.open_rect(row, 4, 1)[0]         # k=4: anchor 2, window [x-2, x+1]
    # erode -> [7, 13]; dilate with the same window -> [6, 15]: shifted one pixel to the right
    assert np.flatnonzero(out).tolist() == list(range(6, 16))
    out5 = co.open_rect(row, 5, 1)[0]
    assert np.flatnonzero(out5).tolist() == list(range(5, 15))
    assert co.open_rect(row, 11, 1).sum() == 0
    # a run touching the border is not eroded from outside (border = +inf): length 3 survives k = 5
    edge = np.zeros((1, 30), np.uint8)
    edge[0, 0:3] = 255
    assert np.flatnonzero(co.open_rect(edge, 5, 1)[0]).tolist() == [0, 1, 2]
    edge2 = np.zeros((1, 30), np.uint8)
    edge2[0, 0:2] = 255
    assert co.open_rect(edge2, 5, 1).sum() == 0


def test_separator_post_process_synthetic_page():
    H, W = 300, 400
    m = np.zeros((H, W, 2), np.uint8)
    m[50:53, 20:380, 0] = 255                # horizontal rule
    m[60:280, 200:203, 0] = 255              # vertical rule
    m[10:13, 10:13, 0] = 255                 # blob of 9 px: removed by the CC filter (min size 99/100)
    m[100:112, 300:312, 0] = 255             # 12x12 blob: passes the CC filter and both openings (k_h=6, k_v=6)
    m[:, :, 1] = 255 - m[:, :, 0]
    out = co.separator_post_process(m)
    k_h, k_v, k_c = co.separator_kernel_sizes(H, W)
    assert (k_h, k_v, k_c) == (6, 6, 4)
    hz, vt = out["horizontal"], out["vertical"]
    assert hz[10:13, 10:13].sum() == 0 and vt[10:13, 10:13].sum() == 0
    assert hz[51, 100] == 255 and vt[51, 100] == 0           # 3 rows high: not a vertical separator
    assert vt[150, 201] == 255 and hz[150, 201] == 0         # 3 px wide: not a horizontal separator
    assert np.all(hz[vt > 0] == 0)                           # horizontal minus vertical
    assert vt[105, 305] == 255 and hz[105, 305] == 0         # square blob ends up vertical only


def test_gaussian5_matches_scipy_mirror():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(23, 31), dtype=np.uint8)
    k = np.array([1, 4, 6, 4, 1], np.int64)
    ref = ndimage.convolve(img.astype(np.int64), np.outer(k, k), mode="mirror")
    assert np.array_equal(co.gaussian5(img), ((ref + 128) >> 8).astype(np.uint8))
    assert np.all(co.gaussian5(np.full((7, 9), 200, np.uint8)) == 200)


def test_otsu_separates_two_modes_and_matches_variance_argmax():
    rng = np.random.default_rng(3)
    img = np.concatenate([rng.normal(60, 8, 5000), rng.normal(190, 10, 3000)]).clip(0, 255).astype(np.uint8)
    t = co.otsu_threshold(img.reshape(100, 80))
    assert 90 < t < 160
    hist = np.bincount(img, minlength=256).astype(np.float64)
    p = hist / hist.sum()
    best, best_t = -1, 0
    for i in range(256):
        q1 = p[:i + 1].sum()
        q2 = 1 - q1
        if q1 < 1e-7 or q2 < 1e-7:
            continue
        m1 = (np.arange(i + 1) * p[:i + 1]).sum() / q1
        m2 = (np.arange(i + 1, 256) * p[i + 1:]).sum() / q2
        s = q1 * q2 * (m1 - m2) ** 2
        if s > best * (1 + 1e-12):
            best, best_t = s, i
    assert t == best_t


def test_edt_exact_against_bruteforce():
    rng = np.random.default_rng(4)
    b = (rng.random((20, 24)) < 0.85).astype(np.uint8) * 255
    d2 = co.edt_sq(b)
    zy, zx = np.nonzero(b == 0)
    for y in range(20):
        for x in range(24):
            exp = 0 if b[y, x] == 0 else int(((zy - y) ** 2 + (zx - x) ** 2).min())
            assert d2[y, x] == exp


def test_swt_distance_transform_stroke():
    g = np.full((40, 60), 230, np.uint8)
    g[10:30, 20:31] = 20                      # dark bar 11 px wide, 20 px high
    swt = co.swt_distance_transform(g)
    assert swt[5, 5] == 0                     # background
    assert swt[20, 25] in (5, 6, 7)           # half the stroke width (blur widens the bar slightly)
    assert swt.max() <= 8


def test_textline_features():
    swt = np.zeros((30, 80), np.uint8)
    swt[5:15, 10:16] = 2
    swt[8:12, 12:14] = 3                      # glyph 1: 6 wide, 10 high, max 3
    swt[5:25, 30:40] = 4                      # glyph 2: 10 wide, 20 high, max 4
    swt[6:8, 50:52] = 9                       # too small (2x2): rejected
    swt[20:23, 45:75] = 1                     # 30 x 3: aspect ratio 10 > 8 rejected
    sw, th = co.swt_features_textline(swt, (0, 0, 79, 29))
    assert sw == 3.5 and th == 20
    assert co.swt_features_textline(np.zeros((5, 5), np.uint8), (0, 0, 4, 4)) == (0.0, 0)
    prob = np.full((10, 10), 0.5)
    assert co.net_prob_textline(prob, (2, 2, 4, 3)) == 0.5
