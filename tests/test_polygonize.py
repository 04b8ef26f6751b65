"""Raster -> polygon rings (row a10): hand cases + rasterise-back / per-region properties (CPU only).

rasterio/GDAL is not installed; vertex order and start vertex are GDAL-specific, so polygons
are compared as covered pixel sets per region, plus exact rings for hand-checkable shapes."""
import numpy as np
import pytest
from scipy import ndimage

from citlab_article_separation_new_amd import polygonize as pz


def test_rectangle_ring_is_closed_and_on_pixel_corners():
    m = np.zeros((6, 9), np.uint8)
    m[1:4, 2:7] = 255
    polys = pz.shapes(m)
    assert polys == [[[(2.0, 1.0), (7.0, 1.0), (7.0, 4.0), (2.0, 4.0), (2.0, 1.0)]]]


def test_other_values_are_ignored_like_the_reference_filter():
    m = np.zeros((4, 4), np.uint8)
    m[0, 0] = 255
    m[2:4, 2:4] = 128                                       # p[1] == 255 filter (base:195)
    assert len(pz.shapes(m)) == 1
    assert pz.shapes(np.zeros((3, 3), np.uint8)) == []


def test_diagonal_pixels_share_one_polygon_with_connectivity_8():
    m = np.zeros((5, 5), np.uint8)
    m[1, 1] = m[2, 2] = m[3, 1] = 255
    p8 = pz.shapes(m, connectivity=8)
    p4 = pz.shapes(m, connectivity=4)
    assert len(p8) == 1 and len(p4) == 3
    ring = p8[0][0]
    # the ring passes twice through the corners where diagonal pixels touch
    assert ring[0] == ring[-1] and ring.count((2.0, 2.0)) == 2 and ring.count((2.0, 3.0)) == 2
    assert np.array_equal(pz.rasterize(p8, 5, 5), m)
    assert np.array_equal(pz.rasterize(p4, 5, 5), m)


def test_holes_and_island_inside_a_hole():
    m = np.zeros((11, 11), np.uint8)
    m[1:10, 1:10] = 255
    m[3:8, 3:8] = 0
    m[5, 5] = 255                                           # island inside the hole: its own polygon
    polys = pz.shapes(m)
    assert len(polys) == 2
    outer = max(polys, key=len)
    assert len(outer) == 2                                  # exterior + one hole
    assert outer[0] == [(1.0, 1.0), (10.0, 1.0), (10.0, 10.0), (1.0, 10.0), (1.0, 1.0)]
    assert sorted(outer[1][:-1]) == sorted([(3.0, 3.0), (3.0, 8.0), (8.0, 8.0), (8.0, 3.0)])
    assert np.array_equal(pz.rasterize(polys, 11, 11), m)


def test_two_holes_touching_at_a_corner_stay_separate():
    m = np.full((6, 6), 255, np.uint8)
    m[2, 2] = m[3, 3] = 0
    polys = pz.shapes(m)
    assert len(polys) == 1 and len(polys[0]) == 3
    assert np.array_equal(pz.rasterize(polys, 6, 6), m)


@pytest.mark.parametrize("density", [0.05, 0.3, 0.5, 0.62, 0.8, 0.97])
@pytest.mark.parametrize("connectivity", [4, 8])
def test_random_masks_region_by_region(density, connectivity):
    rng = np.random.default_rng(int(density * 100) + connectivity)
    H, W = 48, 57
    m = ((rng.random((H, W)) < density) * 255).astype(np.uint8)
    polys = pz.shapes(m, connectivity=connectivity)
    lab, n = ndimage.label(m > 0, structure=np.ones((3, 3)) if connectivity == 8 else None)
    assert len(polys) == n
    for i, poly in enumerate(polys):                        # raster first-touch order == scipy label order
        assert np.array_equal(pz.rasterize([poly], H, W) > 0, lab == i + 1)
        for ring in poly:
            assert ring[0] == ring[-1] and len(ring) >= 5
            for (x0, y0), (x1, y1) in zip(ring[:-1], ring[1:]):
                assert (x0 == x1) != (y0 == y1)             # axis-parallel, no zero-length or collinear pairs
            for a, b, c in zip(ring[:-2], ring[1:-1], ring[2:]):
                assert not (a[0] == b[0] == c[0]) and not (a[1] == b[1] == c[1])


def test_page_sized_mask_round_trip():
    m = np.zeros((1500, 1000), np.uint8)
    for i in range(14):
        m[100 * i + 50:100 * i + 53, 40:960] = 255
    for i in range(8):
        m[80:1400, 110 * i + 60:110 * i + 63] = 255
    polys = pz.shapes(m)
    assert len(polys) == 2                                  # the grid and the detached last rule
    assert np.array_equal(pz.rasterize(polys, 1500, 1000), m)
