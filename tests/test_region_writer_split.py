"""separator_region_to_page_writer.py:107-227,329-337 without GEOS: text lines cut at vertical separators, separator
polygons cut at large holes.  Expected PAGE-XML content is derived by hand for axis-parallel cases (what shapely's
``difference`` / ``intersection`` give there is unambiguous up to ring start / part order, which are normalised:
rings start top-left and run clockwise on screen, parts go left to right)."""
import numpy as np
import pytest

from citlab_article_separation_new_amd import rect_geometry as rg
from citlab_article_separation_new_amd.page_xml import Page
from citlab_article_separation_new_amd.region_to_page_writer import SeparatorRegionToPageWriter

PAGE = """<?xml version="1.0" encoding="UTF-8"?>
<PcGts xmlns="http://schema.primaresearch.org/PAGE/gts/pagecontent/2013-07-15">
  <Metadata><Creator>t</Creator><Created>2020-01-01T00:00:00</Created><LastChange>2020-01-01T00:00:00</LastChange></Metadata>
  <Page imageFilename="img.png" imageWidth="400" imageHeight="300">
    <TextRegion id="r1" type="paragraph"><Coords points="5,5 210,5 210,260 5,260"/>
      <TextLine id="A" custom="readingOrder {index:0;} structure {id:a1; type:article;}"><Coords points="10,10 200,10 200,40 10,40"/>
        <Baseline points="10,35 200,35"/>
        <Word id="w1"><Coords points="10,10 90,10 90,40 10,40"/><TextEquiv><Unicode>Hello</Unicode></TextEquiv></Word>
        <Word id="w2"><Coords points="110,10 200,10 200,40 110,40"/><TextEquiv><Unicode>World</Unicode></TextEquiv></Word>
        <TextEquiv><Unicode>Hello World</Unicode></TextEquiv></TextLine>
      <TextLine id="B"><Coords points="10,50 90,50 90,80 10,80"/><Baseline points="10,75 90,75"/>
        <TextEquiv><Unicode>left only</Unicode></TextEquiv></TextLine>
      <TextLine id="C"><Coords points="99,90 101,90 101,120 99,120"/><Baseline points="99,118 101,118"/></TextLine>
      <TextLine id="D"><Coords points="10,130 200,130 200,150 120,150 120,170 10,170"/><Baseline points="10,165 120,165 200,145"/>
        <TextEquiv><Unicode>no words here</Unicode></TextEquiv></TextLine>
      <TextLine id="E"><Coords points="10,180 200,185 200,215 10,210"/><Baseline points="10,205 200,210"/></TextLine>
      <TextLine id="F"><Coords points="102,220 200,220 200,250 102,250"/><Baseline points="102,245 200,245"/></TextLine>
      <TextEquiv><Unicode>region text</Unicode></TextEquiv>
    </TextRegion>
    <TextRegion id="r2"><Coords points="220,5 390,5 390,100 220,100"/>
      <TextLine id="G"><Coords points="230,10 380,10 380,40 230,40"/><Baseline points="230,35 380,35"/></TextLine>
    </TextRegion>
    <SeparatorRegion id="SeparatorRegion_7"><Coords points="1,1 5,1 5,5 1,5"/></SeparatorRegion>
  </Page>
</PcGts>
"""

VSEP = [(98, 0), (102, 0), (102, 300), (98, 300), (98, 0)]                    # x in [98, 102]
HSEP_HOLE = [[(220, 120), (320, 120), (320, 220), (220, 220), (220, 120)],       # 100 x 100 ring
             [(240, 140), (240, 190), (280, 190), (280, 140), (240, 140)]]       # hole 40 x 50 = 2000 px^2
HSEP_SMALL = [[(220, 230), (320, 230), (320, 260), (220, 260), (220, 230)],
              [(230, 240), (230, 250), (240, 250), (240, 240), (230, 240)]]       # hole 100 px^2: filled


def _run(tmp_path, region_dict, remove_holes=True):
    (tmp_path / "page").mkdir(exist_ok=True)
    src = tmp_path / "page" / "img.xml"
    src.write_text(PAGE)
    w = SeparatorRegionToPageWriter(str(src), None, None, None, region_dict)
    w.remove_separator_regions_from_page()
    w.merge_regions(remove_holes)
    out = tmp_path / "page" / "out.xml"
    w.save_page_xml(str(out))
    return Page(str(out))


def test_text_lines_are_cut_at_a_vertical_separator(tmp_path):
    page = _run(tmp_path, {"SeparatorRegion_vertical": [[VSEP]]})
    r1, r2 = page.get_text_regions()
    got = {tl.id: tl for tl in r1.text_lines}
    assert [tl.id for tl in r1.text_lines] == ["A_1", "A_2", "B", "D_1", "D_2", "E_1", "E_2", "F"]        # C is swallowed
    a1, a2 = got["A_1"], got["A_2"]
    assert a1.surr_p == [(10, 10), (98, 10), (98, 40), (10, 40), (10, 10)]
    assert a2.surr_p == [(102, 10), (200, 10), (200, 40), (102, 40), (102, 10)]
    assert a1.baseline == [(10, 35), (98, 35)] and a2.baseline == [(102, 35), (200, 35)]
    assert [w.id for w in a1.words] == ["w1"] and [w.id for w in a2.words] == ["w2"]
    assert a1.text == "Hello" and a2.text == "World"
    assert a1.get_article_id() == "a1" and a2.get_article_id() == "a1"                          # custom tags travel
    assert a1.custom["readingOrder"] == {"index": "0"}
    # untouched lines keep their nodes, text and coordinates
    assert got["B"].surr_p == [(10, 50), (90, 50), (90, 80), (10, 80)] and got["B"].text == "left only"
    # L-shaped line: the left part keeps the foot, the slanted baseline piece is cut at x = 102 (y = 165 - 20 * 18 / 80)
    d1, d2 = got["D_1"], got["D_2"]
    assert d1.surr_p == [(10, 130), (98, 130), (98, 170), (10, 170), (10, 130)]
    assert d2.surr_p == [(102, 130), (200, 130), (200, 150), (120, 150), (120, 170), (102, 170), (102, 130)]
    assert d1.baseline == [(10, 165), (98, 165)] and d2.baseline == [(102, 165), (120, 165), (200, 145)]
    assert d1.text == "no words here" and d2.text == "no words here"          # no Word elements: text is copied (:199-201)
    # a SLANTED line is cut too (general polygon clipping, poly_clip.py): top edge y = 180 + 5 (x - 10) / 190 -> 182.32 / 182.42 at
    # x = 98 / 102, bottom edge 30 lower; baseline y = 205 + 5 (x - 10) / 190 -> 207.32 / 207.42; PAGE coordinates are integers
    e1, e2 = got["E_1"], got["E_2"]
    assert e1.surr_p == [(10, 180), (98, 182), (98, 212), (10, 210), (10, 180)]
    assert e2.surr_p == [(102, 182), (200, 185), (200, 215), (102, 212), (102, 182)]
    assert e1.baseline == [(10, 205), (98, 207)] and e2.baseline == [(102, 207), (200, 210)]
    # a line that only touches the separator is not cut
    assert got["F"].surr_p == [(102, 220), (200, 220), (200, 250), (102, 250)]
    assert [tl.id for tl in r2.text_lines] == ["G"]
    seps = page.get_regions()["SeparatorRegion"]
    assert [(s.id, s.get_orientation(), s.points) for s in seps] == [("SeparatorRegion_1", "vertical", VSEP)]
    # XML layout of a split line: Coords, Baseline, Word*, TextEquiv
    tags = [c.tag.split("}")[1] for c in a1.node]
    assert tags == ["Coords", "Baseline", "Word", "TextEquiv"]
    # the region's own TextEquiv stays behind its lines
    assert [c.tag.split("}")[1] for c in r1.node][-1] == "TextEquiv"


def test_two_separators_cut_one_line_into_three(tmp_path):
    second = [(150, 0), (153, 0), (153, 300), (150, 300), (150, 0)]
    page = _run(tmp_path, {"SeparatorRegion_vertical": [[VSEP], [second]]})
    r1 = page.get_text_regions()[0]
    ids = [tl.id for tl in r1.text_lines]
    assert ids[:3] == ["A_1", "A_2_1", "A_2_2"]
    parts = {tl.id: tl for tl in r1.text_lines}
    assert parts["A_2_1"].surr_p[0] == (102, 10) and parts["A_2_1"].surr_p[1] == (150, 10)
    assert parts["A_2_2"].surr_p[0] == (153, 10) and parts["A_2_2"].baseline == [(153, 35), (200, 35)]
    # w2 spans x 110..200: 40 px of it lie in the middle piece, 47 px in the right one -> it goes right (:190-197)
    assert parts["A_2_1"].text == "" and parts["A_2_2"].text == "World"
    assert [w.id for w in parts["A_2_2"].words] == ["w2"] and parts["A_2_1"].words == []
    assert sum(rg.Region.from_rings([parts[i].surr_p]).area for i in ("A_1", "A_2_1", "A_2_2")) == 190 * 30 - 4 * 30 - 3 * 30


def test_horizontal_separators_do_not_cut_and_holes_are_handled(tmp_path):
    page = _run(tmp_path, {"SeparatorRegion_horizontal": [HSEP_HOLE, HSEP_SMALL], "SeparatorRegion": [[VSEP]]})
    assert [tl.id for tl in page.get_text_regions()[0].text_lines] == ["A", "B", "C", "D", "E", "F"]
    seps = page.get_regions()["SeparatorRegion"]
    # plain separators first, then horizontal (merge order :359); the ring with the 2000 px^2 hole is cut at the
    # hole's line centroid x = 260 into a left and a right part, the 100 px^2 hole is filled
    assert [s.get_orientation() for s in seps] == [None, "horizontal", "horizontal", "horizontal"]
    left, right, small = seps[1].points, seps[2].points, seps[3].points
    assert left == [(220, 120), (260, 120), (260, 140), (240, 140), (240, 190), (260, 190), (260, 220), (220, 220), (220, 120)]
    assert right == [(260, 120), (320, 120), (320, 220), (260, 220), (260, 190), (280, 190), (280, 140), (260, 140), (260, 120)]
    assert small == HSEP_SMALL[0]
    whole = rg.Region.from_rings(HSEP_HOLE)
    assert rg.Region.from_rings([left]).area + rg.Region.from_rings([right]).area == whole.area == 100 * 100 - 2000
    # without hole removal only the exterior ring is written (:349-351)
    page = _run(tmp_path, {"SeparatorRegion_horizontal": [HSEP_HOLE]}, remove_holes=False)
    assert [s.points for s in page.get_regions()["SeparatorRegion"]] == [HSEP_HOLE[0]]


def test_region_algebra_against_rasterisation():
    """difference / intersection / contains on random rectilinear polygons == the same on pixel masks"""
    from citlab_article_separation_new_amd import polygonize
    rng = np.random.default_rng(5)
    for _ in range(20):
        masks = []
        for _ in range(2):
            m = np.zeros((40, 50), bool)
            for _ in range(int(rng.integers(1, 5))):
                y0, x0 = int(rng.integers(0, 30)), int(rng.integers(0, 40))
                m[y0:y0 + int(rng.integers(2, 15)), x0:x0 + int(rng.integers(2, 15))] = True
            masks.append(m)
        regs = []
        for m in masks:
            polys = polygonize.shapes(m.astype(np.uint8) * 255, 255, connectivity=4)
            r = rg.Region.from_rings(polys[0])
            for p in polys[1:]:
                r = r.union(rg.Region.from_rings(p))
            regs.append(r)
            assert r.area == m.sum()
        a, b = regs
        for op, ref in ((a.difference(b), masks[0] & ~masks[1]), (a.intersection(b), masks[0] & masks[1]),
                        (a.union(b), masks[0] | masks[1])):
            assert op.area == ref.sum()
            back = polygonize.rasterize(op.polygons(), 40, 50) > 0
            assert np.array_equal(back, ref)
        assert a.contains(b) == bool((masks[1] & ~masks[0]).sum() == 0)
        assert a.overlaps(b) == bool((masks[0] & masks[1]).any())


def test_polyline_clipping():
    sep = rg.Region.from_rings([VSEP])
    assert rg.clip_polyline_outside([(0, 10), (300, 10)], sep) == [[(0, 10), (98, 10)], [(102, 10), (300, 10)]]
    assert rg.clip_polyline_outside([(0, 10), (50, 10), (50, 20)], sep) == [[(0, 10), (50, 10), (50, 20)]]
    assert rg.clip_polyline_outside([(99, 10), (101, 10)], sep) == []
    assert rg.clip_polyline_outside([(100, 10), (200, 10)], sep) == [[(102, 10), (200, 10)]]
    assert rg.clip_polyline_outside([(0, 0), (200, 100)], sep) == [[(0, 0), (98, 49)], [(102, 51), (200, 100)]]
    assert rg.ring_line_centroid([(0, 0), (4, 0), (4, 2), (0, 2)]) == (2.0, 1.0)
    assert not rg.is_rectilinear([(0, 0), (4, 1), (4, 2), (0, 2)]) and rg.is_rectilinear(VSEP)


def test_general_polygon_difference_hand_cases():
    """poly_clip.py: slanted outlines against rectilinear separators, expected rings and areas derived by hand"""
    from citlab_article_separation_new_amd import poly_clip as pc

    def same(got, want):                                               # cut points on slanted edges are one float interpolation
        return len(got) == len(want) and all(len(g) == len(w) and np.allclose(g, w, rtol=0, atol=1e-9) for g, w in zip(got, want))
    quad = [(0, 0), (100, 10), (100, 40), (0, 30)]                     # parallelogram: width 100, height 30, area 3000
    # (1) separator through the whole line: two parts; top edge y = x / 10, bottom edge y = 30 + x / 10
    parts = pc.difference_parts(quad, [(48.0, -10.0, 52.0, 100.0)])
    assert same(parts, [[(0.0, 0.0), (48.0, 4.8), (48.0, 34.8), (0.0, 30.0)], [(52.0, 5.2), (100.0, 10.0), (100.0, 40.0), (52.0, 35.2)]])
    assert abs(sum(pc.ring_area2(p) for p in parts) / 2 - (3000 - 4 * 30)) < 1e-9
    # (2) separator that ends inside the line: ONE part with a notch (the pieces left / right / below the rectangle are merged)
    parts = pc.difference_parts(quad, [(48.0, -10.0, 52.0, 20.0)])
    assert same(parts, [[(0.0, 0.0), (48.0, 4.8), (48.0, 20.0), (52.0, 20.0), (52.0, 5.2), (100.0, 10.0), (100.0, 40.0), (0.0, 30.0)]])
    assert abs(pc.ring_area2(parts[0]) / 2 - (3000 - 60)) < 1e-9        # notch: x in [48, 52], y from x / 10 to 20 -> 80 - 20
    # (3) two rectangles of one separator region (an L): the line is cut by the vertical bar only
    parts = pc.difference_parts(quad, [(48.0, -10.0, 52.0, 50.0), (52.0, 45.0, 90.0, 50.0)])
    assert len(parts) == 2 and parts[0][0] == (0.0, 0.0) and np.allclose(parts[1][0], (52.0, 5.2))
    # (4) swallowed / untouched
    assert pc.difference_parts(quad, [(-5.0, -5.0, 105.0, 45.0)]) == []
    assert same(pc.difference_parts(quad, [(200.0, 0.0, 210.0, 50.0)]), [[(0.0, 0.0), (100.0, 10.0), (100.0, 40.0), (0.0, 30.0)]])
    # (5) a concave outline (a "U") cut across both arms: three parts, not one polygon with zero-width bridges
    u = [(0, 0), (30, 0), (30, 60), (70, 60), (70, 0), (100, 0), (100, 100), (0, 100)]
    parts = pc.difference_parts(u, [(-10.0, 20.0, 110.0, 30.0)])
    assert sorted(round(pc.ring_area2(p) / 2) for p in parts) == [600, 600, 100 * 100 - 40 * 60 - 600 - 600 - 2 * 300]
    # intersection area with a convex (slanted) word box and the point / poly-line predicates
    word = [(40, 0), (60, 2), (60, 42), (40, 40)]
    left, right = pc.difference_parts(quad, [(48.0, -10.0, 52.0, 100.0)])
    assert abs(pc.intersection_area(left, word) - 8 * 30) < 1e-9 and abs(pc.intersection_area(right, word) - 8 * 30) < 1e-9
    assert pc.is_convex(word) and not pc.is_convex(u)
    assert pc.point_in_ring((48, 20), left) and pc.point_in_ring((10, 5), left) and not pc.point_in_ring((50, 20), left)
    assert pc.polyline_meets_ring([(-5, 15), (20, 18)], left) and not pc.polyline_meets_ring([(60, 20), (90, 25)], left)


def test_general_polygon_difference_conserves_area():
    """random convex and star-shaped polygons minus random disjoint rectangles: area(parts) + area(polygon within the rectangles)
    == area(polygon) (the rectangles are convex: their share is an exact Sutherland-Hodgman clip)"""
    from citlab_article_separation_new_amd import poly_clip as pc
    rng = np.random.default_rng(5)
    for trial in range(60):
        n = int(rng.integers(3, 9))
        ang = np.sort(rng.random(n) * 2 * np.pi)
        rad = (40 + 30 * rng.random(n)) if trial % 2 else np.full(n, 60.0)
        poly = [(100 + r * np.cos(a), 100 + r * np.sin(a)) for a, r in zip(ang, rad)]
        if abs(pc.ring_area2(poly)) < 1e-6:
            continue
        rects = []
        x = 40.0
        for _ in range(int(rng.integers(1, 4))):               # disjoint rectangles, left to right
            x0 = x + rng.random() * 30
            x1 = x0 + 2 + rng.random() * 15
            y0 = 30 + rng.random() * 100
            rects.append((x0, y0, x1, y0 + 5 + rng.random() * 120))
            x = x1 + 1
        parts, holes = pc.difference_parts(poly, rects, with_holes=True)       # (a rectangle inside the polygon leaves a hole)
        kept = (sum(pc.ring_area2(p) for p in parts) + sum(pc.ring_area2(h) for h in holes)) / 2
        cut = sum(pc.intersection_area(poly, [(a, b), (c, b), (c, d), (a, d)]) for a, b, c, d in rects)
        assert abs(kept + cut - abs(pc.ring_area2(poly)) / 2) < 1e-6, (trial, kept, cut)
        assert all(pc.ring_area2(p) > 0 for p in parts)


def test_vertices_exactly_on_a_cut_line():
    """a polygon vertex ON the cut line whose two neighbours lie on the lower side gave two coincident crossings
    that were paired with their neighbours along the line instead of with each other -- area was dropped or a corrupted ring came
    back.  PAGE coordinates and rectangle edges are both integers, so this is the common case, not an exotic one."""
    from citlab_article_separation_new_amd import poly_clip as pc
    area = lambda r: abs(pc.ring_area2(r)) / 2
    # repro 1: (-5, -2) touches x = -5 from the left
    ring = [(-6, -1), (-5, -2), (-6, -5), (-1, -3), (4, 5), (-1, 5), (-2, 2), (-4, 0)]
    lo, hi = pc.split_by_line(pc._open(ring), 0, -5.0)
    assert abs(sum(map(area, lo)) + sum(map(area, hi)) - area(ring)) < 1e-9 and area(ring) == 42.5
    assert len(lo) == 2 and len(hi) == 1                       # the touching vertex separates two lower pieces
    # repro 2: the rectangle [-5,-2] x [-6,-5] does not touch the polygon (its corner (-5,-3) lies on x = -5): untouched, same ring
    ring = [(-5, -3), (3, -5), (6, -5), (3, 1), (0, 6), (-5, 5), (-2, 1), (-3, 0)]
    parts = pc.difference_parts(ring, [(-5, -6, -2, -5)])
    assert len(parts) == 1 and abs(area(parts[0]) - area(ring)) < 1e-9 and area(ring) == 67.5
    assert sorted(parts[0]) == sorted((float(x), float(y)) for x, y in ring)
    # hand case: a diamond whose left and right corners lie on the two cut lines of a rectangle as wide as the diamond
    diamond = [(0, 5), (5, 0), (10, 5), (5, 10)]
    assert pc.difference_parts(diamond, [(0, 4, 10, 6)]) and abs(sum(map(area, pc.difference_parts(diamond, [(0, 4, 10, 6)]))) - (50 - 18)) < 1e-9
    # an edge that runs ALONG the cut line (x = 4 between y = 2 and y = 6): nothing is lost on either side
    poly = [(0, 0), (4, 2), (4, 6), (0, 8), (8, 8), (8, 0)]
    lo, hi = pc.split_by_line(pc._open(poly), 0, 4.0)
    assert abs(sum(map(area, lo)) + sum(map(area, hi)) - area(poly)) < 1e-9


def test_general_polygon_difference_on_an_integer_grid_conserves_area():
    """text-line-like outlines with INTEGER vertices against stacked separator rectangles with integer edges (vertices on cut lines
    all the time): the balance area(parts) - area(holes) + area inside the rectangles == area(polygon) holds, or the function
    refuses (ClipError, raised only for outlines that touch themselves) -- it never returns an unbalanced result."""
    import math
    import random
    from citlab_article_separation_new_amd import poly_clip as pc
    rng = random.Random(5)
    area = lambda r: abs(pc.ring_area2(r)) / 2
    refused = done = 0
    for _ in range(4000):
        R = rng.choice([6, 6, 20, 200])
        pts = set()
        n = rng.randint(4, 10)
        while len(pts) < n:
            pts.add((rng.randint(-R, R), rng.randint(-R, R)))
        pts = list(pts)
        cx, cy = sum(p[0] for p in pts) / n, sum(p[1] for p in pts) / n
        ring = sorted(pts, key=lambda p: math.atan2(p[1] - cy, p[0] - cx))        # star-shaped around the centroid
        if area(ring) == 0 or len(pc.repair_ring(ring)) != 1:
            continue
        rects, x, y = [], rng.randint(-R, R - 1), -R - 1
        for _k in range(rng.randint(1, 3)):                  # a column of disjoint rectangles
            h, w = rng.randint(1, max(1, R // 2)), rng.randint(1, 3)
            rects.append((x, y, x + w, y + h))
            y += h + rng.randint(0, 3)
        try:
            parts, holes = pc.difference_parts(ring, rects, with_holes=True)
        except pc.ClipError:
            refused += 1
            continue
        done += 1
        taken = sum(pc.intersection_area(ring, [(a, b), (c, b), (c, d), (a, d)]) for a, b, c, d in rects)
        assert abs(sum(map(area, parts)) - sum(map(area, holes)) + taken - area(ring)) < 1e-7 * max(1.0, area(ring))
    assert done > 3000 and refused <= 2, (done, refused)


def test_repair_ring_like_buffer0_hand_cases():
    """separator_region_to_page_writer.py:164,170,189 repair outlines with ``buffer(0)``.  Two hand-derived cases:
    (a) shapely's own documentation example -- a bow tie that TOUCHES itself at the vertex (1, 1): both triangles are kept;
    (b) a figure 8 whose edges CROSS at (1, 1): the two lobes are wound in opposite directions, the lobe through the ring's highest
    vertex gives the direction (GEOS Orientation.isCCW) and the other one is dropped (buffer(0)'s known area loss)."""
    from citlab_article_separation_new_amd import poly_clip as pc
    area = lambda r: abs(pc.ring_area2(r)) / 2
    a = pc.repair_ring([(0, 0), (0, 2), (1, 1), (2, 2), (2, 0), (1, 1), (0, 0)])
    assert sorted(sorted(r) for r in a) == [[(0.0, 0.0), (0.0, 2.0), (1.0, 1.0)], [(1.0, 1.0), (2.0, 0.0), (2.0, 2.0)]]
    b = pc.repair_ring([(0, 0), (2, 2), (2, 0), (0, 2)])
    assert len(b) == 1 and sorted(b[0]) == [(1.0, 1.0), (2.0, 0.0), (2.0, 2.0)] and area(b[0]) == 1.0
    # a simple ring comes back unchanged; a slanted text line with a self-crossing tail keeps its body
    quad = [(0, 0), (100, 10), (100, 40), (0, 30)]
    assert pc.repair_ring(quad) == [[(float(x), float(y)) for x, y in quad]]
    tail = [(0, 0), (100, 10), (100, 40), (0, 30), (-10, 10), (-20, 30), (-20, 10), (-10, 30)]      # the last four points cross: a small 8
    body = pc.repair_ring(tail)
    assert sum(map(area, body)) > 3000 and all(pc.ring_area2(r) * pc.ring_area2(body[0]) > 0 for r in body)


def test_slanted_line_that_cannot_be_clipped_is_left_uncut(tmp_path, caplog):
    """the writer never stores a ring that does not balance: a ClipError leaves the text line as it was and logs it"""
    import logging
    from citlab_article_separation_new_amd import poly_clip as pc
    from citlab_article_separation_new_amd.region_to_page_writer import SeparatorRegionToPageWriter

    class Line:
        id, words, text, baseline = "l1", [], "t", None
        surr_p = [(0, 0), (100, 10), (100, 40), (0, 30)]

        def split_copy(self, *a):
            raise AssertionError("an uncut line is not copied")
    real = pc.difference_parts
    pc.difference_parts = lambda *a, **k: (_ for _ in ()).throw(pc.ClipError("forced"))
    try:
        with caplog.at_level(logging.WARNING):
            out = SeparatorRegionToPageWriter._split_text_lines([Line()], [[(48, -10), (52, -10), (52, 100), (48, 100)]])
    finally:
        pc.difference_parts = real
    assert len(out) == 1 and out[0].surr_p == Line.surr_p and "left uncut" in caplog.text


def test_fuzz_integer_rings_against_a_rectangle_never_leave_a_line_uncut():
    """a fuzz of integer-coordinate star polygons against one rectangle was reported to raise ClipError in ~0.1 % of the cases
    (a text line then keeps its outline across the separator).  On the writer's path -- ``repair_ring`` first, as
    region_to_page_writer.py does, then ``difference_parts`` -- 9 000 seeded rings here, 52 000 when the finding was looked into (simple stars with every vertex on the integer grid,
    and rings with random radii that cross themselves) raise none, and the pieces conserve the area of the repaired loops.  (Rings that
    cross themselves handed to ``difference_parts`` WITHOUT the repair do fail its area check -- that is the guard working: 1.8 % of
    such rings in the same fuzz.)"""
    import math
    from citlab_article_separation_new_amd import poly_clip
    rect = (30, 30, 50, 45)
    n_cases = 0
    for seed in range(3):
        rng = np.random.default_rng(1000 + seed)
        for _ in range(1500):
            k = 2 * int(rng.integers(4, 9))
            star = [(int(round(40 + (rng.integers(10, 30) if i % 2 == 0 else rng.integers(3, 9)) * math.cos(2 * math.pi * i / k))),
                     int(round(40 + (rng.integers(10, 30) if i % 2 == 0 else rng.integers(3, 9)) * math.sin(2 * math.pi * i / k)))) for i in range(k)]
            m = int(rng.integers(5, 16))
            ang = np.sort(rng.uniform(0, 2 * math.pi, m))
            wild = [(int(round(40 + r * math.cos(a))), int(round(40 + r * math.sin(a)))) for a, r in zip(ang, rng.integers(2, 30, m))]
            for ring in (star, wild):
                ring = [p for i, p in enumerate(ring) if p != ring[i - 1]]
                if len(ring) < 3 or poly_clip.ring_area2(ring) == 0:
                    continue
                loops = poly_clip.repair_ring(ring)
                for lp in loops:
                    parts = poly_clip.difference_parts(lp, [rect])           # raises ClipError if the crossings do not pair / the area does not balance
                    left = sum(abs(poly_clip.ring_area2(q)) for q in parts) / 2.0
                    assert left <= abs(poly_clip.ring_area2(lp)) / 2.0 + 1e-6
                n_cases += 1
    assert n_cases > 8000
