"""Writers that put region information coming out of the ARU-Net post-processors into PAGE-XML.

    RegionToPageWriter             region_to_page_writer.py:13-46
    SeparatorRegionToPageWriter    separator_region_to_page_writer.py:11-25,107-386

``merge_regions`` does what the reference does with shapely, for the rectilinear case (``rect_geometry.py``; no GEOS in
this build): text lines that a VERTICAL separator runs through are cut into the parts left and right of it (``:156-227``),
words and text follow the part they overlap most, the baseline is cut with the line and parts without a baseline piece
are dropped; separator polygons with holes larger than 1000 px^2 are cut at the holes (``:30-70,329-337``).  A text line
whose outline is slanted goes through ``poly_clip.py`` (general simple polygon minus the separator's rectangles; the cut points on
slanted edges are rounded to integers like every PAGE coordinate).
Where GEOS' output order is an implementation detail (parts of a MultiPolygon, ring start vertex) parts are ordered left
to right and rings start at their top-left corner, running clockwise on screen.
"""
import logging
import os

from .image_io import get_image_dimensions
from .net_post_processing_helper import get_scaling_factor
from .page_xml import Page
from . import poly_clip
from .rect_geometry import Region, clip_polyline_outside, cut_at_holes, is_rectilinear, polyline_touches

SEPARATOR_REGION = "SeparatorRegion"


class RegionToPageWriter:
    """Owns the Page object a post-processor writes its regions into (``region_to_page_writer.py:13-46``)."""

    def __init__(self, path_to_page, path_to_image=None, fixed_height=None, scaling_factor=None, *args, **kwargs):
        self.path_to_page = path_to_page
        self.scaling_factor = None
        size = None
        if path_to_image is not None:
            size = get_image_dimensions(path_to_image)          # (width, height)
            self.scaling_factor = get_scaling_factor(size[1], size[0], scaling_factor, fixed_height)
        self.page_object = self.load_page_object(path_to_page, path_to_image, size)

    def load_page_object(self, path_to_page, path_to_image, size=None):
        """An existing PAGE file is parsed; a missing one becomes an empty page that names the image and carries the
        image size multiplied by the scaling factor and truncated (the reference writes the
        *scaled* size there, kept)."""
        if os.path.exists(path_to_page):
            return Page(path_to_page)
        width, height = size if size is not None else get_image_dimensions(path_to_image)
        return Page(img_filename=path_to_image, img_w=int(self.scaling_factor * width),
                    img_h=int(self.scaling_factor * height))

    def save_page_xml(self, save_path):
        folder = os.path.dirname(save_path)
        if folder and not os.path.isdir(folder):
            os.makedirs(folder, exist_ok=True)
        self.page_object.write_page_xml(save_path)


class SeparatorRegionToPageWriter(RegionToPageWriter):
    def __init__(self, path_to_page, path_to_image=None, fixed_height=None, scaling_factor=None, region_dict=None):
        super().__init__(path_to_page, path_to_image, fixed_height, scaling_factor)
        self.region_dict = region_dict or {}

    def remove_separator_regions_from_page(self):
        self.page_object.remove_regions(SEPARATOR_REGION)

    # -- text lines vs. vertical separators (separator_region_to_page_writer.py:156-227) --------------------------
    @staticmethod
    def _split_text_lines(lines, separator_rings):
        """``lines``: the current parts of ONE original text line; returns the parts after cutting at one separator."""
        sep = Region.from_rings(separator_rings)
        if sep.is_empty():
            return lines
        out = []
        rects = sep.rectangles()
        for line in lines:
            rectilinear = is_rectilinear(line.surr_p)
            if rectilinear:
                outline = Region.from_rings([line.surr_p])
                if sep.contains(outline):                    # swallowed by the separator: the line disappears (:172-174)
                    continue
                if not outline.overlaps(sep):
                    out.append(line)
                    continue
                parts = [p[0] for p in outline.difference(sep).polygons()]
                part_regions = [Region.from_rings([p]) for p in parts]
                part_area = lambda k, ring: (Region.from_rings([ring]) if is_rectilinear(ring) else Region.box(*_bbox(ring))) \
                    .intersection(part_regions[k]).area
                meets = lambda piece, k: _polyline_in(piece, part_regions[k])
            else:
                # slanted outline: general polygon minus the separator's rectangles (poly_clip.py)
                if len(line.surr_p) < 3:
                    out.append(line)
                    continue
                # an outline that crosses itself is repaired like ``Polygon(points).buffer(0)`` (:170): cut at its crossings, the loops
                # wound like the ring are kept and clipped one by one (a MultiPolygon's parts)
                loops = poly_clip.repair_ring(line.surr_p)
                try:
                    parts = [q for lp in loops for q in poly_clip.difference_parts(lp, rects)]
                except poly_clip.ClipError as e:             # never write a ring that does not balance: the line stays uncut
                    logging.warning("text line %s left uncut (%s)", line.id, e)
                    out.append(line)
                    continue
                parts.sort(key=lambda q: (min(x for x, _ in q), min(y for _, y in q)))
                area0 = sum(abs(poly_clip.ring_area2(lp)) for lp in loops) / 2.0
                area1 = sum(abs(poly_clip.ring_area2(p)) for p in parts) / 2.0
                if not parts:                                # swallowed by the separator
                    continue
                if area1 >= area0 * (1.0 - 1e-12) and len(loops) == 1:   # no area lost: the separator does not run through the line
                    out.append(line)
                    continue
                # a word goes to the part it overlaps most: exact for convex words (quadrilaterals), bounding box otherwise
                part_area = lambda k, ring: poly_clip.intersection_area(parts[k], ring if poly_clip.is_convex(ring) else _box_ring(ring))
                meets = lambda piece, k: poly_clip.polyline_meets_ring(piece, parts[k])
            words = [[] for _ in parts]
            if len(parts) != 1:
                for word in line.words:                      # a word goes to the part it overlaps most (:190-197)
                    if len(word.surr_p) < 3:                 # a word without an outline overlaps nothing: argmax of zeros = part 0
                        words[0].append(word)
                        continue
                    areas = [part_area(k, word.surr_p) for k in range(len(parts))]
                    words[max(range(len(parts)), key=lambda k: areas[k])].append(word)
            else:
                words[0] = list(line.words)
            # every baseline piece goes to the FIRST part it meets, a later piece of the same part replaces an earlier one;
            # parts without a piece are dropped (:203-224, _get_parent_region :146-152)
            pieces = clip_polyline_outside(line.baseline, sep) if line.baseline else []
            owner = {}
            for piece in pieces:
                for k in range(len(parts)):
                    if meets(piece, k):
                        owner[k] = piece
                        break
            if not line.baseline:
                owner = {k: None for k in range(len(parts))}  # (the reference reads an unset variable here: kept all)
            for k in sorted(owner):
                text = line.text
                if len(parts) != 1 and line.words:
                    text = " ".join(w.text for w in words[k])
                new_id = line.id if len(parts) == 1 else f"{line.id}_{k + 1}"
                ring = parts[k] if rectilinear else [(_num(x), _num(y)) for x, y in parts[k] + parts[k][:1]]      # closed like the others
                out.append(line.split_copy(new_id, ring, owner[k], words[k], text))
        return out

    def merge_regions(self, remove_holes=True):
        """:359-386 order: plain, horizontal, vertical separators; vertical ones first cut the text lines they cross;
        every polygon (cut at its large holes) becomes one SeparatorRegion whose custom tag carries the orientation."""
        for separator_type in (SEPARATOR_REGION, SEPARATOR_REGION + "_horizontal", SEPARATOR_REGION + "_vertical"):
            polygons = self.region_dict.get(separator_type)
            if polygons is None:
                continue
            polygons = [_as_rings(p) for p in polygons]
            orientation = separator_type[len(SEPARATOR_REGION) + 1:] or None
            if orientation == "vertical":
                for region in self.page_object.get_text_regions():
                    if not region.text_lines:
                        continue
                    parts_of = {id(tl): [tl] for tl in region.text_lines}
                    for rings in polygons:
                        if not rings or not all(is_rectilinear(r) for r in rings):
                            continue
                        for key in parts_of:
                            parts_of[key] = self._split_text_lines(parts_of[key], rings)
                    final = [tl for tl0 in region.text_lines for tl in parts_of[id(tl0)]]
                    if [id(t) for t in final] != [id(t) for t in region.text_lines]:
                        region.replace_text_lines(final)
            for rings in polygons:
                if not rings:
                    continue
                if remove_holes and len(rings) > 1 and all(is_rectilinear(r) for r in rings):
                    for exterior in cut_at_holes(rings, min_hole_area=1000):
                        self.page_object.add_separator_region(exterior, orientation)
                else:
                    self.page_object.add_separator_region(rings[0], orientation)


def _as_rings(polygon):
    """a polygon is a list of rings (exterior first) or, for convenience, a bare ring"""
    if polygon and isinstance(polygon[0], (list, tuple)) and polygon[0] and isinstance(polygon[0][0], (list, tuple)):
        return [list(r) for r in polygon]
    return [list(polygon)]


def _box_ring(points):
    x0, y0, x1, y1 = _bbox(points)
    return [(x0, y0), (x1, y0), (x1, y1), (x0, y1)]


def _num(v):
    """PAGE coordinates are integers: a cut point on a slanted edge is rounded (half away from zero like ``int(round(v))`` of
    polygon.py), a value that is integral stays as it is"""
    r = round(v)
    return int(r) if abs(v - r) < 1e-9 else int(v + 0.5) if v >= 0 else -int(-v + 0.5)


def _bbox(points):
    xs, ys = [p[0] for p in points], [p[1] for p in points]
    return min(xs), min(ys), max(xs), max(ys)


def _polyline_in(points, region):
    """does the poly-line touch the (closed) region?  -- ``intersects`` of :109-113"""
    if polyline_touches(points, region):
        return True
    x0, y0, x1, y1 = region.bounds() if not region.is_empty() else (0, 0, -1, -1)
    rects = region.rectangles()
    return any(rx0 <= px <= rx1 and ry0 <= py <= ry1 for px, py in points for rx0, ry0, rx1, ry1 in rects)
