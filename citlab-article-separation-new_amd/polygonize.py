"""Raster -> polygon rings on pixel-corner coordinates.

Stands in for ``rasterio.features.shapes(binary_image, connectivity=8)`` as used by
``region_net_post_processor_base.py:186-197`` (consumers keep the shapes whose value is 255): one polygon per
8-connected region of foreground pixels; every polygon is a list of closed rings ``[(x, y), ...]`` (first ring =
exterior, further rings = holes) whose vertices are pixel corners under the identity transform, collinear
vertices merged.  Vertex order / start vertex are GDAL implementation details that no consumer of the reference
depends on; here exteriors run clockwise on screen (y down) starting at their top-left corner.

Method: every foreground pixel contributes its sides that face background as unit edges directed so that the
foreground lies to the right; edges are chained into rings; where two rings could cross at a corner shared by two
diagonal foreground pixels the walk turns left, which keeps 8-connected pixels in one ring.
"""
import numpy as np

# headings: 0 = +x, 1 = +y, 2 = -x, 3 = -y  (clockwise on screen)
_DX = np.array([1, 0, -1, 0], dtype=np.int64)
_DY = np.array([0, 1, 0, -1], dtype=np.int64)


def _foreground(fg):
    """Sparse view of the mask: pixel coordinates (raster order) and a zero-padded copy for neighbour tests."""
    H, W = fg.shape
    p = np.zeros((H + 2, W + 2), dtype=bool)
    p[1:-1, 1:-1] = fg
    ys, xs = np.nonzero(fg)
    return p, ys.astype(np.int64), xs.astype(np.int64)


def _boundary_edges(p, ys, xs):
    """-> (vx, vy, heading, pixel index): start vertex, heading and owning pixel of every boundary edge."""
    out = []
    # neighbour offset (dy, dx), start-vertex offset (dx, dy), heading
    for ny, nx, ox, oy, hd in ((-1, 0, 0, 0, 0),           # top side:    (x, y)     -> (x+1, y)
                               (0, 1, 1, 0, 1),            # right side:  (x+1, y)   -> (x+1, y+1)
                               (1, 0, 1, 1, 2),            # bottom side: (x+1, y+1) -> (x, y+1)
                               (0, -1, 0, 1, 3)):          # left side:   (x, y+1)   -> (x, y)
        sel = np.flatnonzero(~p[ys + 1 + ny, xs + 1 + nx])
        out.append((xs[sel] + ox, ys[sel] + oy, np.full(sel.shape, hd, dtype=np.int64), sel))
    return [np.concatenate([o[i] for o in out]) for i in range(4)]


def shapes(mask, value=255, connectivity=8):
    """List of polygons (each a list of closed rings of ``(x, y)`` float tuples) of the regions where
    ``mask == value``; regions in raster first-touch order."""
    if connectivity not in (4, 8):
        raise ValueError("connectivity must be 4 or 8")
    mask = np.asarray(mask)
    if mask.ndim != 2:
        raise ValueError("shapes expects a 2-D array")
    fg = mask == value
    H, W = fg.shape
    if not fg.any():
        return []
    p, ys, xs = _foreground(fg)
    vx, vy, hd, owner = _boundary_edges(p, ys, xs)
    return _rings_from_edges(vx, vy, hd, ys[owner], xs[owner], W, connectivity)


def shapes_from_segments(starts, ends, H, W, connectivity=8):
    """Same result as :func:`shapes` from the maximal straight boundary segments extracted on the GPU
    (``asep_post_boundary_segments``: keys ((vy*(W+1)+vx)*4 + heading) of the start and end vertices, any order).
    The work here is proportional to the number of polygon corners, not to the image or boundary size."""
    if connectivity not in (4, 8):
        raise ValueError("connectivity must be 4 or 8")
    starts = np.asarray(starts, dtype=np.int64)
    ends = np.asarray(ends, dtype=np.int64)
    if starts.size != ends.size:
        raise ValueError("starts and ends must have the same length")
    if starts.size == 0:
        return []
    VW = W + 1
    parts = []
    for h in range(4):
        s = starts[(starts & 3) == h] >> 2
        e = ends[(ends & 3) == h] >> 2
        if s.size != e.size:
            raise ValueError("unbalanced boundary segments")
        sy, sx, ey, ex = s // VW, s % VW, e // VW, e % VW
        # segments of one heading on one grid line are disjoint: order both ends along the line and pair them
        so = np.lexsort((sx, sy)) if h in (0, 2) else np.lexsort((sy, sx))
        eo = np.lexsort((ex, ey)) if h in (0, 2) else np.lexsort((ey, ex))
        parts.append((sx[so], sy[so], ex[eo], ey[eo], np.full(s.size, h, dtype=np.int64)))
    svx, svy, evx, evy, hd = (np.concatenate([p[i] for p in parts]) for i in range(5))
    ns = svx.size
    skey = (svy * VW + svx) * 4 + hd
    order = np.argsort(skey, kind="stable")
    sorted_keys = skey[order]
    end_v = evy * VW + evx

    def lookup(heading):
        k = end_v * 4 + heading
        pos = np.minimum(np.searchsorted(sorted_keys, k), ns - 1)
        return np.where(sorted_keys[pos] == k, order[pos], -1)

    # every segment end is a corner: the ring turns left (8-connectivity keeps diagonal pixels together) or right
    first, second = ((hd + 3) % 4, (hd + 1) % 4) if connectivity == 8 else ((hd + 1) % 4, (hd + 3) % 4)
    nxt = lookup(first)
    nxt = np.where(nxt < 0, lookup(second), nxt)
    if (nxt < 0).any():                                    # pragma: no cover - would be a logic error
        raise RuntimeError("open boundary chain")
    cidx = np.argsort(skey[nxt], kind="stable")            # rings start at their top-left-most vertex
    nxt_l = nxt.tolist()
    ex_l, ey_l = evx.tolist(), evy.tolist()
    visited = bytearray(ns)
    ring_of = np.full(ns, -1, dtype=np.int64)
    rings, ring_area, ring_first = [], [], []
    for c0 in cidx.tolist():
        if visited[c0]:
            continue
        members = []
        pts = []
        c = c0
        while not visited[c]:
            visited[c] = 1
            members.append(c)
            pts.append((float(ex_l[c]), float(ey_l[c])))
            c = nxt_l[c]
        pts.append(pts[0])
        area2 = 0.0
        for (x0, y0), (x1, y1) in zip(pts[:-1], pts[1:]):
            area2 += x0 * y1 - x1 * y0
        ring_of[members] = len(rings)
        rings.append(pts)
        ring_area.append(area2)
        ring_first.append(c0)
    # holes: the foreground pixel that owns the last unit edge of the ring's first segment; walk left along its row
    # to the nearest left-facing (heading 3) boundary segment -- its ring is the exterior or a hole that starts higher
    left = np.flatnonzero(hd == 3)
    lx, ltop, lbot = svx[left], evy[left], svy[left]       # covers pixel rows ltop <= row < lbot, pixel column lx
    parent = {}

    def exterior_of(r):
        chain = []
        while ring_area[r] < 0:
            if r in parent:
                r = parent[r]
                continue
            chain.append(r)
            c = ring_first[r]
            h = int(hd[c])
            ux, uy = int(evx[c]) - int(_DX[h]), int(evy[c]) - int(_DY[h])      # start of the last unit edge
            oy, ox = uy - (1 if h >= 2 else 0), ux - (1 if h in (1, 2) else 0)
            cand = np.flatnonzero((ltop <= oy) & (oy < lbot) & (lx <= ox))
            r = int(ring_of[left[cand[np.argmax(lx[cand])]]])
        for hole in chain:
            parent[hole] = r
        return r

    polys = {}
    for r, pts in enumerate(rings):
        if ring_area[r] > 0:
            polys[r] = [pts]
    for r, pts in enumerate(rings):
        if ring_area[r] < 0:
            polys[exterior_of(r)].append(pts)
    return [polys[r] for r in sorted(polys)]


def _rings_from_edges(vx, vy, hd, oy, ox, W, connectivity):
    ne = vx.shape[0]
    VW = W + 1
    start_key = (vy * VW + vx) * 4 + hd
    order = np.argsort(start_key, kind="stable")
    sorted_keys = start_key[order]
    ex, ey = vx + _DX[hd], vy + _DY[hd]                    # end vertex
    end_v = ey * VW + ex

    def lookup(heading):
        k = end_v * 4 + heading
        pos = np.minimum(np.searchsorted(sorted_keys, k), ne - 1)
        return np.where(sorted_keys[pos] == k, order[pos], -1)

    # 8-connectivity: prefer the left turn; 4-connectivity: prefer the right turn (diagonal pixels separate)
    prefs = ((hd + 3) % 4, hd, (hd + 1) % 4) if connectivity == 8 else ((hd + 1) % 4, hd, (hd + 3) % 4)
    nxt = np.full(ne, -1, dtype=np.int64)
    for h in prefs:
        cand = lookup(h)
        nxt = np.where(nxt < 0, cand, nxt)
    if (nxt < 0).any():                                    # pragma: no cover - would be a logic error
        raise RuntimeError("open boundary chain")
    corner = hd[nxt] != hd                                 # the ring turns at the end vertex of this edge

    # contract straight runs: jump[e] = first corner edge strictly after e (pointer doubling, O(log run length))
    jump = nxt.copy()
    while True:
        todo = ~corner[jump]
        if not todo.any():
            break
        jump[todo] = jump[jump[todo]]
    # a ring's top-left-most vertex is the end vertex of a corner edge c whose successor nxt[c] has the ring's
    # smallest start key: visit corner edges in that order so every ring starts there and exteriors come out in
    # raster first-touch order of their regions
    cidx = np.flatnonzero(corner)
    cidx = cidx[np.argsort(start_key[nxt[cidx]], kind="stable")]
    jump_l = jump.tolist()
    ex_l, ey_l = ex.tolist(), ey.tolist()
    ring_of = np.full(ne, -1, dtype=np.int64)
    visited = bytearray(ne)
    rings, ring_area, ring_first = [], [], []
    for c0 in cidx.tolist():
        if visited[c0]:
            continue
        members = [c0]
        pts = [(float(ex_l[c0]), float(ey_l[c0]))]
        visited[c0] = 1
        c = jump_l[c0]
        while not visited[c]:
            visited[c] = 1
            members.append(c)
            pts.append((float(ex_l[c]), float(ey_l[c])))
            c = jump_l[c]
        pts.append(pts[0])
        area2 = 0.0
        for (x0, y0), (x1, y1) in zip(pts[:-1], pts[1:]):
            area2 += x0 * y1 - x1 * y0
        ring_of[members] = len(rings)
        rings.append(pts)
        ring_area.append(area2)
        ring_first.append(c0)
    # clockwise-on-screen rings (foreground to the right) have a positive shoelace sum with y down: exteriors.
    # A hole belongs to the region of the foreground pixel above its top-left corner: walk left along that pixel's
    # row to the first left-facing boundary edge; its ring is the exterior, or another hole whose own top-left
    # pixel lies strictly higher (so the walk terminates).
    ring_all = np.where(corner, ring_of, ring_of[jump])
    left = np.flatnonzero(hd == 3)
    left_keys = oy[left] * W + ox[left]
    lorder = np.argsort(left_keys, kind="stable")
    left_sorted = left_keys[lorder]
    parent = {}

    def exterior_of(r):
        chain = []
        while ring_area[r] < 0:
            if r in parent:
                r = parent[r]
                continue
            chain.append(r)
            e = ring_first[r]
            pos = np.searchsorted(left_sorted, oy[e] * W + ox[e], side="right") - 1
            r = int(ring_all[left[lorder[pos]]])
        for h in chain:
            parent[h] = r
        return r

    polys = {}
    for r, pts in enumerate(rings):
        if ring_area[r] > 0:
            polys[r] = [pts]
    for r, pts in enumerate(rings):
        if ring_area[r] < 0:
            polys[exterior_of(r)].append(pts)
    return [polys[r] for r in sorted(polys)]


def rasterize(polygons, H, W):
    """Inverse of :func:`shapes` for rings on pixel corners (even-odd rule over all rings): uint8 mask 0/255.
    Lets callers and tests compare polygon sets independent of vertex order."""
    delta = np.zeros((H + 1, W + 2), dtype=np.int64)
    for poly in polygons:
        for ring in poly:
            for (x0, y0), (x1, y1) in zip(ring[:-1], ring[1:]):
                if x0 == x1 and y0 != y1:                  # a vertical edge toggles the coverage to its right
                    ya, yb = (int(y0), int(y1)) if y0 < y1 else (int(y1), int(y0))
                    delta[ya:yb, int(x0)] += 1
    cover = np.cumsum(delta, axis=1)[:H, :W] & 1
    return (cover * 255).astype(np.uint8)
