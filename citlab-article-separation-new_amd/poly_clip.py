"""General (slanted) polygons against rectilinear separator regions -- the part of shapely / GEOS that
``separator_region_to_page_writer.py:154-227`` needs for text lines whose outline is NOT axis-parallel:

    difference_parts(ring, rects)      ``Polygon(ring).difference(separator)`` -> the connected parts' exterior rings
    intersection_area(ring, convex)    ``a.intersection(b).area`` for a convex ``b`` (a word quadrilateral)
    point_in_ring(pt, ring)            inside or on the boundary

The separator is a rectilinear region, i.e. a union of disjoint axis-parallel rectangles (``rect_geometry.Region.rectangles``),
so every cut runs along a line x = c or y = c: a simple polygon is split by such a line exactly (crossings sorted along the
line and paired), the pieces outside a rectangle are collected (left of it, right of it, above and below its column), and at
the end pieces that share a boundary segment of positive length are merged again by cancelling opposite edges -- what remains
are the rings of the connected parts.  Coordinates on the cut lines are exact; the other coordinate of a cut point on a slanted
edge is one floating-point interpolation, computed once and shared by both sides.
"""
from collections import defaultdict


class ClipError(ValueError):
    """the splitter could not pair the crossings of a cut line, or a difference did not conserve area: the caller keeps the
    polygon as it is (and says so) instead of writing a corrupted outline"""


def ring_area2(ring):
    """twice the signed area (positive: counter-clockwise in a y-up frame)"""
    s = 0.0
    n = len(ring)
    for i in range(n):
        x0, y0 = ring[i]
        x1, y1 = ring[(i + 1) % n]
        s += x0 * y1 - x1 * y0
    return s


def _open(ring):
    pts = [(float(x), float(y)) for x, y in ring]
    if len(pts) > 1 and pts[0] == pts[-1]:
        pts = pts[:-1]
    out = []
    for p in pts:                                           # drop repeated points
        if not out or p != out[-1]:
            out.append(p)
    if len(out) > 1 and out[0] == out[-1]:
        out.pop()
    return out


def split_by_line(ring, axis, c):
    """Split the simple polygon ``ring`` (open list of points) along the line ``coordinate[axis] == c``.
    -> (polygons with coordinate <= c, polygons with coordinate >= c).  Points ON the line count as the upper side, so an edge
    that runs along the line belongs to the upper polygons.

    Vertices exactly on the line (PAGE coordinates and rectangle edges are both integers: not exotic) are handled as if the line
    lay an infinitesimal step BELOW them: a crossing next to such a vertex IS the vertex (no interpolation), and crossings at
    the same position are ordered by the direction their edge leaves the line towards the lower side -- a vertex that only
    touches the line from below gives two coincident crossings that pair with EACH OTHER (a zero-area spike on the upper side)
    instead of with their neighbours (``split_by_line([(-6,-1),(-5,-2),(-6,-5),(-1,-3),(4,5),(-1,5),(-2,2),(-4,0)],
    0, -5.0)`` lost 9.3 of 85 area units that way).  Raises ClipError when the crossings cannot be paired."""
    n = len(ring)
    if n < 3:
        return [], []
    side = [1 if p[axis] >= c else -1 for p in ring]
    if all(s > 0 for s in side):
        return [], [list(ring)]
    if all(s < 0 for s in side):
        return [list(ring)], []
    # the ring with crossing points inserted; a crossing belongs to both sides
    seq = []                                                # (point, side: -1 / +1 / 0 = crossing)
    slope = {}
    for i in range(n):
        p, q = ring[i], ring[(i + 1) % n]
        seq.append((p, side[i]))
        if side[i] != side[(i + 1) % n]:
            lo, hi = (p, q) if side[i] < 0 else (q, p)      # lo: the end point below the line
            if hi[axis] == c:                               # the upper end point lies ON the line: the crossing is that vertex
                o = hi[1 - axis]
            else:
                t = (c - p[axis]) / (q[axis] - p[axis])
                o = p[1 - axis] + (q[1 - axis] - p[1 - axis]) * t
            x = (c, o) if axis == 0 else (o, c)
            # tie-break of coincident crossings: where the edge meets a line an infinitesimal step below, relative to o
            slope[len(seq)] = (lo[1 - axis] - o) / (c - lo[axis])
            seq.append((x, 0))
    cross = [k for k, (_, s) in enumerate(seq) if s == 0]
    if len(cross) % 2:
        raise ClipError(f"odd number of crossings ({len(cross)}) with the line {'xy'[axis]} = {c}")
    order = sorted(cross, key=lambda k: (seq[k][0][1 - axis], slope[k]))
    partner = {}
    for a in range(0, len(order) - 1, 2):                   # consecutive crossings along the line bound one interior interval
        partner[order[a]] = order[a + 1]
        partner[order[a + 1]] = order[a]
    m = len(seq)

    def trace(want):
        polys, used = [], set()
        for start in range(m):
            if seq[start][1] != want or start in used:
                continue
            poly, k = [], start
            while True:
                pt, s = seq[k]
                if s == want:
                    if k in used:
                        break
                    used.add(k)
                    poly.append(pt)
                    k = (k + 1) % m
                elif s == 0:
                    poly.append(pt)
                    j = partner[k]
                    poly.append(seq[j][0])
                    k = (j + 1) % m
                else:                                       # ran onto the other side: the pairing is wrong
                    raise ClipError(f"crossings of the line {'xy'[axis]} = {c} do not pair up")
                if k == start:
                    break
            poly = _open(poly)
            if len(poly) >= 3 and abs(ring_area2(poly)) > 0.0:
                polys.append(poly)
        return polys

    return trace(-1), trace(1)


def subtract_rect(ring, rect):
    """pieces of the polygon outside the axis-parallel rectangle (x0, y0, x1, y1); their interiors are disjoint"""
    x0, y0, x1, y1 = rect
    out = []
    left, rest = split_by_line(ring, 0, x0)
    out += left
    for r in rest:
        mid, right = split_by_line(r, 0, x1)
        out += right
        for q in mid:
            top, low = split_by_line(q, 1, y0)
            out += top
            for w in low:
                _, below = split_by_line(w, 1, y1)
                out += below
    return out


def _merge(pieces):
    """union of pieces with disjoint interiors that may share boundary segments along axis-parallel lines: opposite edges
    cancel, the remaining edges are chained into rings; -> exterior rings of the connected parts (counter-clockwise)"""
    pieces = [p if ring_area2(p) > 0 else p[::-1] for p in pieces]
    xs, ys = defaultdict(set), defaultdict(set)             # vertices on every vertical / horizontal line
    for p in pieces:
        for x, y in p:
            xs[x].add(y)
            ys[y].add(x)
    edges = defaultdict(int)
    for p in pieces:
        n = len(p)
        for i in range(n):
            a, b = p[i], p[(i + 1) % n]
            pts = [a, b]
            if a[0] == b[0]:                                # vertical edge: split at every vertex that lies inside it
                lo, hi = sorted((a[1], b[1]))
                mids = sorted(v for v in xs[a[0]] if lo < v < hi)
                pts = [a] + [(a[0], v) for v in (mids if a[1] < b[1] else mids[::-1])] + [b]
            elif a[1] == b[1]:
                lo, hi = sorted((a[0], b[0]))
                mids = sorted(v for v in ys[a[1]] if lo < v < hi)
                pts = [a] + [(v, a[1]) for v in (mids if a[0] < b[0] else mids[::-1])] + [b]
            for u, v in zip(pts[:-1], pts[1:]):
                if edges.get((v, u), 0) > 0:
                    edges[(v, u)] -= 1
                else:
                    edges[(u, v)] += 1
    nxt = defaultdict(list)
    for (u, v), k in edges.items():
        for _ in range(k):
            nxt[u].append(v)
    rings = []
    while True:
        start = next((u for u, vs in nxt.items() if vs), None)
        if start is None:
            break
        ring, u = [], start
        while nxt[u]:
            ring.append(u)
            u = nxt[u].pop()
            if u == start:
                break
        if len(ring) >= 3:
            rings.append(ring)
    out, holes = [], []
    for r in rings:                                          # drop collinear points (cut points on straight edges)
        clean = []
        n = len(r)
        for i in range(n):
            a, b, c = r[i - 1], r[i], r[(i + 1) % n]
            cross = (b[0] - a[0]) * (c[1] - b[1]) - (b[1] - a[1]) * (c[0] - b[0])
            scale = (abs(b[0] - a[0]) + abs(b[1] - a[1])) * (abs(c[0] - b[0]) + abs(c[1] - b[1]))
            if abs(cross) > 1e-12 * scale:                   # (a cut point on a slanted edge is collinear up to one rounding)
                clean.append(b)
        if len(clean) >= 3 and ring_area2(clean) > 0:
            out.append(clean)
        elif len(clean) >= 3:                                # a clockwise ring is a hole of a part (the writers use exteriors only)
            holes.append(clean)
    return out, holes


def difference_parts(ring, rects, with_holes=False):
    """``Polygon(ring).difference(union of rects)`` -> exterior rings of its connected parts, ordered left to right (then top to
    bottom), each starting at its top-left-most vertex (``with_holes``: also the list of hole rings, clockwise).  ``rects`` are
    disjoint (``rect_geometry.Region.rectangles``).  Raises ClipError when the result does not balance (see below)."""
    poly = _open(ring)
    if len(poly) < 3:
        return ([], []) if with_holes else []
    pieces = [poly]
    for rect in rects:
        pieces = [q for p in pieces for q in subtract_rect(p, rect)]
        if not pieces:
            break
    parts, holes = _merge(pieces) if pieces else ([], [])
    # area conservation: what is left + what the (disjoint) rectangles took = the polygon.  A violated balance means a wrong
    # pairing or merge somewhere above -- never hand such a ring to the PAGE writer
    area = lambda r: abs(ring_area2(r)) / 2.0
    left = sum(area(r) for r in parts) - sum(area(r) for r in holes)
    taken = sum(intersection_area(poly, [(x0, y0), (x1, y0), (x1, y1), (x0, y1)]) for x0, y0, x1, y1 in rects)
    if abs(left + taken - area(poly)) > 1e-7 * max(1.0, area(poly)):
        raise ClipError(f"difference does not conserve area: {left:.6f} left + {taken:.6f} inside the rectangles != {area(poly):.6f}")
    if not parts:
        return ([], []) if with_holes else []
    out = []
    for p in parts:
        k = min(range(len(p)), key=lambda i: (p[i][1], p[i][0]))
        out.append(p[k:] + p[:k])
    out.sort(key=lambda p: (min(x for x, _ in p), min(y for _, y in p)))
    return (out, holes) if with_holes else out


def _seg_intersection(p, q, a, b):
    """proper crossing point of the open segments pq and ab (None if they do not cross in their interiors)"""
    d = (q[0] - p[0]) * (b[1] - a[1]) - (q[1] - p[1]) * (b[0] - a[0])
    if d == 0:
        return None
    t = ((a[0] - p[0]) * (b[1] - a[1]) - (a[1] - p[1]) * (b[0] - a[0])) / d
    u = ((a[0] - p[0]) * (q[1] - p[1]) - (a[1] - p[1]) * (q[0] - p[0])) / d
    if not (0.0 < t < 1.0 and 0.0 < u < 1.0):
        return None
    return (p[0] + (q[0] - p[0]) * t, p[1] + (q[1] - p[1]) * t)


def repair_ring(ring):
    """``Polygon(ring).buffer(0)`` for an outline that crosses or touches itself (the reference repairs every text-line and word
    outline that way before it clips: separator_region_to_page_writer.py:164,170,189): the ring is cut at its self-crossings and at
    repeated vertices into simple loops, and the loops that run in the ring's own direction are kept.  GEOS takes that direction
    from the turn at the ring's HIGHEST vertex (``Orientation.isCCW``), so the loop through that vertex is always kept and a loop
    wound the other way (the second lobe of a figure 8) is dropped -- the documented area loss of ``buffer(0)`` on bow ties.
    -> list of simple rings (open point lists), in the order their closing points are met; a simple ring comes back as it is."""
    pts = _open(ring)
    n = len(pts)
    if n < 3:
        return []
    cuts = defaultdict(list)                                 # edge index -> [(parameter along the edge, point)]
    for i in range(n):
        p, q = pts[i], pts[(i + 1) % n]
        for k in range(i + 2, n):
            if i == 0 and k == n - 1:
                continue                                     # neighbours across the closing point
            a, b = pts[k], pts[(k + 1) % n]
            x = _seg_intersection(p, q, a, b)
            if x is not None:
                cuts[i].append((abs(x[0] - p[0]) + abs(x[1] - p[1]), x))
                cuts[k].append((abs(x[0] - a[0]) + abs(x[1] - a[1]), x))
    seq = []
    for i in range(n):
        seq.append(pts[i])
        seq += [x for _, x in sorted(cuts[i])]
    if len(seq) == n and len(set(seq)) == n:
        return [pts]
    top = max(range(len(seq)), key=lambda i: (seq[i][1], -i))            # the highest vertex (first one met on ties)
    loops, stack, where = [], [], {}
    owner_of_top = None
    for idx, pt in enumerate(seq + [seq[0]]):
        if pt in where:                                      # back at a point of the walk: the points since then close a loop
            k = where[pt]
            loop = stack[k:]
            for q in loop[1:]:
                del where[q]
            del stack[k + 1:]
            if len(loop) >= 3 and ring_area2(loop) != 0.0:
                loops.append(loop)
                if seq[top] in loop and owner_of_top is None:
                    owner_of_top = len(loops) - 1
        else:
            where[pt] = len(stack)
            stack.append(pt)
    if not loops:
        return []
    ref = loops[owner_of_top if owner_of_top is not None else max(range(len(loops)), key=lambda i: abs(ring_area2(loops[i])))]
    sign = ring_area2(ref) > 0
    return [lp for lp in loops if (ring_area2(lp) > 0) == sign]


def is_convex(ring):
    pts = _open(ring)
    n = len(pts)
    if n < 3:
        return False
    sign = 0
    for i in range(n):
        a, b, c = pts[i], pts[(i + 1) % n], pts[(i + 2) % n]
        z = (b[0] - a[0]) * (c[1] - b[1]) - (b[1] - a[1]) * (c[0] - b[0])
        if z != 0:
            if sign and (z > 0) != (sign > 0):
                return False
            sign = 1 if z > 0 else -1
    return sign != 0


def intersection_area(ring, convex):
    """area of ``ring`` (any simple polygon) inside the CONVEX polygon ``convex`` (Sutherland-Hodgman: degenerate bridges of a
    concave subject have zero area)"""
    subj = _open(ring)
    clip = _open(convex)
    if ring_area2(clip) < 0:
        clip = clip[::-1]
    for i in range(len(clip)):
        a, b = clip[i], clip[(i + 1) % len(clip)]
        inside = lambda p: (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0]) >= 0
        out = []
        for k in range(len(subj)):
            p, q = subj[k], subj[(k + 1) % len(subj)]
            ip, iq = inside(p), inside(q)
            if ip:
                out.append(p)
            if ip != iq:
                d = (b[0] - a[0]) * (q[1] - p[1]) - (b[1] - a[1]) * (q[0] - p[0])
                t = ((b[0] - a[0]) * (a[1] - p[1]) - (b[1] - a[1]) * (a[0] - p[0])) / d
                out.append((p[0] + (q[0] - p[0]) * t, p[1] + (q[1] - p[1]) * t))
        subj = out
        if len(subj) < 3:
            return 0.0
    return abs(ring_area2(subj)) / 2.0


def point_in_ring(pt, ring):
    """inside or on the boundary"""
    x, y = float(pt[0]), float(pt[1])
    pts = _open(ring)
    inside = False
    n = len(pts)
    for i in range(n):
        (x0, y0), (x1, y1) = pts[i], pts[(i + 1) % n]
        cross = (x1 - x0) * (y - y0) - (y1 - y0) * (x - x0)
        if cross == 0 and min(x0, x1) <= x <= max(x0, x1) and min(y0, y1) <= y <= max(y0, y1):
            return True
        if (y0 > y) != (y1 > y) and x < x0 + (x1 - x0) * (y - y0) / (y1 - y0):
            inside = not inside
    return inside


def polyline_meets_ring(points, ring):
    """``LineString.intersects(Polygon)``: a vertex of the line inside the polygon, or a segment crossing its boundary"""
    pts = [(float(x), float(y)) for x, y in points]
    if any(point_in_ring(p, ring) for p in pts):
        return True
    poly = _open(ring)

    def seg_cross(p, q, a, b):
        def o(u, v, w):
            return (v[0] - u[0]) * (w[1] - u[1]) - (v[1] - u[1]) * (w[0] - u[0])
        d1, d2, d3, d4 = o(a, b, p), o(a, b, q), o(p, q, a), o(p, q, b)
        return (d1 > 0) != (d2 > 0) and (d3 > 0) != (d4 > 0) and d1 != 0 and d2 != 0 and d3 != 0 and d4 != 0
    for p, q in zip(pts[:-1], pts[1:]):
        for i in range(len(poly)):
            if seg_cross(p, q, poly[i], poly[(i + 1) % len(poly)]):
                return True
    return False
