"""Exact polygon algebra for RECTILINEAR polygons (all edges parallel to the axes) without GEOS.

The reference cuts text lines at vertical separators and separator polygons at their holes with shapely
(``separator_region_to_page_writer.py:107-227,329-337``: ``difference``, ``intersection``, ``contains``,
``intersects``, ``buffer(0)``).  The separators come out of the raster polygonizer and are rectilinear by
construction; text lines / words / regions in newspaper PAGE-XML are overwhelmingly boxes.  For that family the
operations are exact on a coordinate-compressed cell grid:

    region  = (xs, ys, occ): sorted breakpoints and a boolean matrix, cell (i, j) = [xs[j], xs[j+1]] x [ys[i], ys[i+1]]

Boolean operations merge the breakpoints of both operands; polygons are read back off the cell matrix with the
raster polygonizer (4-connectivity, the way GEOS separates polygons that only touch in a corner) and mapped to the
original coordinates.  Poly-lines (baselines) may have any direction: they are clipped against the region's rows of
rectangles segment by segment.  Anything that is not rectilinear is reported by ``is_rectilinear`` so that callers can
leave such objects alone instead of approximating them.
"""
import numpy as np

from . import polygonize


def is_rectilinear(points):
    pts = _open_ring(points)
    if len(pts) < 4:
        return False
    n = len(pts)
    for k in range(n):
        (x0, y0), (x1, y1) = pts[k], pts[(k + 1) % n]
        if x0 != x1 and y0 != y1:
            return False
    return True


def _open_ring(points):
    pts = [(p[0], p[1]) for p in points]
    if len(pts) > 1 and pts[0] == pts[-1]:
        pts = pts[:-1]
    return pts


class Region:
    """Finite union of axis-parallel rectangles with a common grid."""

    def __init__(self, xs, ys, occ):
        self.xs = np.asarray(xs, dtype=np.float64)
        self.ys = np.asarray(ys, dtype=np.float64)
        self.occ = np.asarray(occ, dtype=bool)

    # -- construction ----------------------------------------------------------------------------------------
    @classmethod
    def from_rings(cls, rings):
        """even-odd fill of rectilinear rings (first ring = exterior, further rings = holes; any orientation)"""
        rings = [_open_ring(r) for r in rings if len(_open_ring(r)) >= 4]
        for r in rings:
            if not is_rectilinear(r):
                raise ValueError("not a rectilinear ring")
        if not rings:
            return cls([0.0, 0.0], [0.0, 0.0], np.zeros((1, 1), bool))
        xs = np.unique(np.array([p[0] for r in rings for p in r], dtype=np.float64))
        ys = np.unique(np.array([p[1] for r in rings for p in r], dtype=np.float64))
        if len(xs) < 2 or len(ys) < 2:
            return cls([0.0, 0.0], [0.0, 0.0], np.zeros((1, 1), bool))
        delta = np.zeros((len(ys) - 1, len(xs)), dtype=np.int64)
        for r in rings:
            n = len(r)
            for k in range(n):
                (x0, y0), (x1, y1) = r[k], r[(k + 1) % n]
                if x0 == x1 and y0 != y1:                    # a vertical edge toggles the coverage to its right
                    ia, ib = np.searchsorted(ys, [min(y0, y1), max(y0, y1)])
                    delta[ia:ib, np.searchsorted(xs, x0)] += 1
        occ = (np.cumsum(delta, axis=1)[:, :-1] & 1).astype(bool)
        return cls(xs, ys, occ)

    @classmethod
    def box(cls, x0, y0, x1, y1):
        return cls([x0, x1], [y0, y1], np.ones((1, 1), bool))

    # -- boolean algebra -------------------------------------------------------------------------------------
    def _on_grid(self, xs, ys):
        """occupancy re-sampled on a finer grid that contains this region's breakpoints"""
        out = np.zeros((len(ys) - 1, len(xs) - 1), dtype=bool)
        if not self.occ.any():
            return out
        cx, cy = (xs[:-1] + xs[1:]) / 2, (ys[:-1] + ys[1:]) / 2
        jx = np.searchsorted(self.xs, cx, side="right") - 1
        iy = np.searchsorted(self.ys, cy, side="right") - 1
        okx = (jx >= 0) & (jx < len(self.xs) - 1)
        oky = (iy >= 0) & (iy < len(self.ys) - 1)
        sub = self.occ[np.clip(iy, 0, self.occ.shape[0] - 1)][:, np.clip(jx, 0, self.occ.shape[1] - 1)]
        return sub & oky[:, None] & okx[None, :]

    def _combine(self, other, op):
        xs = np.union1d(self.xs, other.xs)
        ys = np.union1d(self.ys, other.ys)
        return Region(xs, ys, op(self._on_grid(xs, ys), other._on_grid(xs, ys)))

    def difference(self, other):
        return self._combine(other, lambda a, b: a & ~b)

    def intersection(self, other):
        return self._combine(other, lambda a, b: a & b)

    def union(self, other):
        return self._combine(other, lambda a, b: a | b)

    # -- measures / predicates -------------------------------------------------------------------------------
    @property
    def area(self):
        return float((np.diff(self.ys)[:, None] * np.diff(self.xs)[None, :])[self.occ].sum())

    def is_empty(self):
        return not self.occ.any()

    def bounds(self):
        ii, jj = np.nonzero(self.occ)
        return (float(self.xs[jj.min()]), float(self.ys[ii.min()]), float(self.xs[jj.max() + 1]), float(self.ys[ii.max() + 1]))

    def overlaps(self, other):
        """interiors share area (shapely's ``intersects`` is also true for a mere boundary contact, whose difference is
        the unchanged polygon -- callers treat that as "nothing to cut")"""
        return not self.intersection(other).is_empty()

    def contains(self, other):
        return (not other.is_empty()) and other.difference(self).is_empty()

    def rectangles(self):
        """maximal horizontal runs of occupied cells: list of (x0, y0, x1, y1)"""
        out = []
        for i in range(self.occ.shape[0]):
            row = self.occ[i]
            j = 0
            while j < len(row):
                if row[j]:
                    k = j
                    while k + 1 < len(row) and row[k + 1]:
                        k += 1
                    out.append((float(self.xs[j]), float(self.ys[i]), float(self.xs[k + 1]), float(self.ys[i + 1])))
                    j = k + 1
                else:
                    j += 1
        return out

    # -- back to polygons ------------------------------------------------------------------------------------
    def polygons(self):
        """list of polygons, each [exterior, hole, ...] as closed rings of (x, y); 4-connected cell groups are separate
        polygons; ordered by (min x, min y) of their exterior"""
        if self.is_empty():
            return []
        polys = polygonize.shapes(self.occ.astype(np.uint8) * 255, 255, connectivity=4)
        out = []
        for poly in polys:
            out.append([[(float(self.xs[int(x)]), float(self.ys[int(y)])) for x, y in ring] for ring in poly])
        out.sort(key=lambda p: (min(x for x, _ in p[0]), min(y for _, y in p[0])))
        return out


def ring_line_centroid(ring):
    """centroid of a closed ring AS A LINE (length-weighted mean of the segment midpoints): what shapely returns for
    ``polygon.interiors[0].centroid`` (a LinearRing is a 1-dimensional geometry)"""
    pts = _open_ring(ring)
    n = len(pts)
    total = sx = sy = 0.0
    for k in range(n):
        (x0, y0), (x1, y1) = pts[k], pts[(k + 1) % n]
        length = ((x1 - x0) ** 2 + (y1 - y0) ** 2) ** 0.5
        total += length
        sx += length * (x0 + x1) / 2
        sy += length * (y0 + y1) / 2
    if total == 0:
        return pts[0]
    return sx / total, sy / total


def ring_area(ring):
    pts = _open_ring(ring)
    a = 0.0
    for k in range(len(pts)):
        (x0, y0), (x1, y1) = pts[k], pts[(k + 1) % len(pts)]
        a += x0 * y1 - x1 * y0
    return abs(a) / 2


def cut_at_holes(rings, min_hole_area=1000):
    """``convert_polygon_with_holes`` (separator_region_to_page_writer.py:30-70) after the hole filter of ``:329-333``:
    holes of at most ``min_hole_area`` px^2 are filled; while a part has a hole it is cut by the vertical line through
    the (line) centroid of its first hole into the part left and the part right of that line.  Returns exterior rings.
    GEOS' ring / part order is not reproducible without GEOS: holes are taken in raster order, parts left to right."""
    exterior, holes = rings[0], [h for h in rings[1:] if ring_area(h) > min_hole_area]
    todo = [Region.from_rings([exterior] + holes)]
    done = []
    while todo:
        region = todo.pop(0)
        requeue = []
        for poly in region.polygons():
            if len(poly) == 1:
                done.append(poly[0])
                continue
            cx = ring_line_centroid(poly[1])[0]
            part = Region.from_rings(poly)
            x0, y0, x1, y1 = part.bounds()
            if cx < x0 or cx > x1:                          # cannot happen for a hole of this polygon; mirrors :38-39
                done.append(poly[0])
                continue
            for half in (part.intersection(Region.box(x0, y0, cx, y1)), part.intersection(Region.box(cx, y0, x1, y1))):
                if not half.is_empty():
                    requeue.append(half)
        todo = requeue + todo
    return done


def clip_polyline_outside(points, region):
    """pieces of the poly-line that lie outside the region (``LineString.difference(polygon)``): list of point lists;
    cut points are exact for axis-parallel segments and floating point for slanted ones"""
    rects = region.rectangles()
    pts = [(float(x), float(y)) for x, y in points]
    subs = []                                                # (segment, t_from, t_to) outside the region, in order
    for k in range(len(pts) - 1):
        (x0, y0), (x1, y1) = pts[k], pts[k + 1]
        inside = []
        for rx0, ry0, rx1, ry1 in rects:                     # Liang-Barsky per rectangle
            t0, t1, ok = 0.0, 1.0, True
            for p, q in ((-(x1 - x0), x0 - rx0), (x1 - x0, rx1 - x0), (-(y1 - y0), y0 - ry0), (y1 - y0, ry1 - y0)):
                if p == 0:
                    if q < 0:
                        ok = False
                        break
                else:
                    t = q / p
                    if p < 0:
                        t0 = max(t0, t)
                    else:
                        t1 = min(t1, t)
            if ok and t0 < t1:
                inside.append((t0, t1))
        inside.sort()
        t = 0.0
        for a, b in inside:
            if a > t:
                subs.append((k, t, a))
            t = max(t, b)
        if t < 1.0:
            subs.append((k, t, 1.0))
    pieces = []
    for n, (k, a, b) in enumerate(subs):
        (x0, y0), (x1, y1) = pts[k], pts[k + 1]
        at = lambda t: pts[k] if t == 0.0 else (pts[k + 1] if t == 1.0 else
                                                (_snap(x0 + (x1 - x0) * t, region.xs), _snap(y0 + (y1 - y0) * t, region.ys)))
        joins = n > 0 and subs[n - 1][0] == k - 1 and subs[n - 1][2] == 1.0 and a == 0.0
        if joins:
            pieces[-1].append(at(b))
        else:
            pieces.append([at(a), at(b)])
    out = []
    for piece in pieces:
        dedup = [piece[0]]
        for q in piece[1:]:
            if q != dedup[-1]:
                dedup.append(q)
        if len(dedup) > 1:
            out.append(dedup)
    return out


def _snap(v, grid):
    """a cut lies on a grid line of the region: undo the rounding of ``p0 + (p1 - p0) * t``"""
    k = int(np.searchsorted(grid, v))
    for c in (k - 1, k):
        if 0 <= c < len(grid) and abs(grid[c] - v) <= 1e-9 * max(1.0, abs(v)):
            return float(grid[c])
    return v


def polyline_touches(points, region):
    """does any part of the poly-line run through the region's interior?"""
    pts = [(float(x), float(y)) for x, y in points]
    total = sum(((x1 - x0) ** 2 + (y1 - y0) ** 2) ** 0.5 for (x0, y0), (x1, y1) in zip(pts[:-1], pts[1:]))
    kept = 0.0
    for piece in clip_polyline_outside(pts, region):
        kept += sum(((x1 - x0) ** 2 + (y1 - y0) ** 2) ** 0.5 for (x0, y0), (x1, y1) in zip(piece[:-1], piece[1:]))
    return kept < total - 1e-9
