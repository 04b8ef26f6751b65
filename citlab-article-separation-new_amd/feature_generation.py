"""PAGE-XML -> graph json for the GNN: the step immediately before ``run_gnn_clustering``.

Re-states ``article_separation/gnn/input/feature_generation.py`` on this package's PAGE-XML classes, without
shapely / OpenCV:

    15 node features  = region size + centre (4, :18-45), top / bottom baseline size + centre (8, :48-82),
                        stroke width (1, :162-181), text height (1, :184-203), heading flag (1, :206-216)
    edge set          = Delaunay neighbours of the region centres rounded to 50 px (:512-538) or fully connected
                        (:494-509; also used for < 4 nodes, :666-667)
    2 edge features   = (horizontally, vertically) separated, bounding-box rules (:319-398) or centre-line
                        intersection (:219-290; segment tests replace shapely's LineString.intersects)
    visual regions    = bounding box per node (:472-479), convex hull per edge (:482-491)
    gt relations      = same majority article id (:762-790)

The stroke-width / text-height features use the GPU distance transform (``image_ops.swt_distance_transform``); the
word-vector similarity features (``TextblockSimilarity``, gensim models) are outside this build.
"""
import functools
import json
import logging
import os
import re
import time

import numpy as np

from .page_xml import Page
from .path_util import get_img_from_page_path


# ---------------------------------------------------------------------------------------------------------------
# small geometry helpers (python_util/geometry/util.py:508-570, python_util/math/rounding.py:36-44)
# ---------------------------------------------------------------------------------------------------------------
def get_bounding_box(points):
    """feature_generation.py:293-297 -> (min_x, max_x, min_y, max_y)."""
    points = np.asarray(points)
    return np.min(points[:, 0]), np.max(points[:, 0]), np.min(points[:, 1]), np.max(points[:, 1])


def bounding_box(points):
    xs, ys = zip(*points)
    return [(min(xs), min(ys)), (max(xs), min(ys)), (max(xs), max(ys)), (min(xs), max(ys))]


def convex_hull(points):
    """Andrew's monotone chain, strict left turns (collinear points dropped), util.py:523-570."""

    def turn_left(p, q, r):
        return (q[0] - p[0]) * (r[1] - p[1]) - (r[0] - p[0]) * (q[1] - p[1]) > 0

    pts = sorted(points, key=functools.cmp_to_key(
        lambda a, b: -1 if (a[0] < b[0] or (a[0] == b[0] and a[1] < b[1])) else (0 if a == b else 1)))
    lower = []
    for pt in pts:
        while len(lower) > 1 and not turn_left(lower[-2], lower[-1], pt):
            lower.pop()
        lower.append(pt)
    upper = []
    for pt in reversed(pts):
        while len(upper) > 1 and not turn_left(upper[-2], upper[-1], pt):
            upper.pop()
        upper.append(pt)
    return lower[:-1] + upper[:-1]


def round_by_precision_and_base(x, prec=2, base=1.0):
    return (base * (np.array(x) / base).round()).round(prec)


def _orient(p, q, r):
    v = (float(q[0]) - float(p[0])) * (float(r[1]) - float(p[1])) - (float(q[1]) - float(p[1])) * (float(r[0]) - float(p[0]))
    return int(v > 0) - int(v < 0)


def _on_segment(p, q, r):
    return min(p[0], q[0]) <= r[0] <= max(p[0], q[0]) and min(p[1], q[1]) <= r[1] <= max(p[1], q[1])


def segments_intersect(a1, a2, b1, b2):
    """Closed segments share at least one point (== shapely ``LineString.intersects`` for two segments)."""
    o1, o2 = _orient(a1, a2, b1), _orient(a1, a2, b2)
    o3, o4 = _orient(b1, b2, a1), _orient(b1, b2, a2)
    if o1 != o2 and o3 != o4:
        return True
    return ((o1 == 0 and _on_segment(a1, a2, b1)) or (o2 == 0 and _on_segment(a1, a2, b2)) or
            (o3 == 0 and _on_segment(b1, b2, a1)) or (o4 == 0 and _on_segment(b1, b2, a2)))


def line_poly_intersection(line, polygon):
    """:300-312: the segment ``line`` = (p, q) meets the boundary of ``polygon`` (closed on the fly)."""
    polygon = list(polygon)
    if polygon[0] != polygon[-1]:
        polygon.append(polygon[0])
    return any(segments_intersect(line[0], line[1], polygon[i], polygon[i + 1]) for i in range(len(polygon) - 1))


def line_in_bounding_box(line, min_x, max_x, min_y, max_y):
    """:315-320."""
    x1, x2 = min(line[0][0], line[1][0]), max(line[0][0], line[1][0])
    y1, y2 = min(line[0][1], line[1][1]), max(line[0][1], line[1][1])
    return x1 > min_x and x2 < max_x and y1 > min_y and y2 < max_y


# ---------------------------------------------------------------------------------------------------------------
# node features
# ---------------------------------------------------------------------------------------------------------------
def _size_center(points, norm_x, norm_y):
    min_x, max_x, min_y, max_y = get_bounding_box(np.asarray(points, dtype=np.int32))
    width = float(max_x) - float(min_x)
    height = float(max_y) - float(min_y)
    return [width / norm_x, height / norm_y, (min_x + max_x) / (2 * norm_x), (min_y + max_y) / (2 * norm_y)]


def get_text_region_geometric_features(text_region, norm_x, norm_y):
    """:18-45."""
    return _size_center(text_region.points, norm_x, norm_y)


def get_text_region_baseline_features(text_region, norm_x, norm_y):
    """:48-82: first and last text line of the region."""
    feature = []
    for tl in (text_region.text_lines[0], text_region.text_lines[-1]):
        feature.extend(_size_center(tl.baseline, norm_x, norm_y))
    return feature


def _swt_components(crop):
    from .heading_net_post_processor import StrokeWidthDistanceTransform
    swt = StrokeWidthDistanceTransform.__new__(StrokeWidthDistanceTransform)
    swt._clean_ccs = 2
    return swt.clean_connected_components(swt.connected_components_cv(crop))


def get_textline_stroke_widths_heights_dist_trafo(page_path, text_lines, img_path=None, swt_img=None, device=0):
    """:106-159.  ``swt_img`` may be supplied by the caller (tests); otherwise the page image is decoded and the
    distance transform runs on the GPU."""
    if swt_img is None:
        from . import image_ops
        from .heading_net_post_processor import bgr_to_gray_u8
        from .image_io import load_image_bgr
        if img_path is None:
            img_path = get_img_from_page_path(page_path)
        if not img_path:
            raise ValueError(f"Could not find corresponding image file to pagexml '{page_path}'")
        d_swt = image_ops.swt_distance_transform_device(bgr_to_gray_u8(load_image_bgr(img_path)), device)
        boxes = []
        for text_line in text_lines:
            min_x, max_x, min_y, max_y = get_bounding_box(np.asarray(text_line.surr_p, dtype=np.int32))
            boxes.append([min_x, min_y, max_x + 1, max_y + 1])
        sws, hts = image_ops.swt_line_features(d_swt, boxes, device=device)
        return ({tl.id: sws[i] for i, tl in enumerate(text_lines)},
                {tl.id: int(hts[i]) for i, tl in enumerate(text_lines)})
    stroke_widths, heights = {}, {}
    for text_line in text_lines:
        min_x, max_x, min_y, max_y = get_bounding_box(np.asarray(text_line.surr_p, dtype=np.int32))
        crop = swt_img[min_y:max_y + 1, min_x:max_x + 1]
        vals, height = [], 0
        for (cx, cy, cw, ch) in _swt_components(crop):
            vals.append(np.max(crop[cy:cy + ch, cx:cx + cw]))
            height = max(height, ch)
        stroke_widths[text_line.id] = np.median(vals) if vals else 0.0
        heights[text_line.id] = height
    return stroke_widths, heights


def _region_max_feature(text_region, per_line, norm):
    """:162-203: maximum over the region's lines that carry text; empty regions get 0."""
    if all(not line.text for line in text_region.text_lines):
        return [0.0]
    return [np.max([per_line[line.id] for line in text_region.text_lines if line.text]) / norm]


def get_text_region_stroke_width_feature(text_region, textline_stroke_widths, norm=1.0):
    return _region_max_feature(text_region, textline_stroke_widths, norm)


def get_text_region_text_height_feature(text_region, textline_heights, norm=1.0):
    return _region_max_feature(text_region, textline_heights, norm)


def get_text_region_heading_feature(text_region):
    """:206-216 (a region without type attribute counts as paragraph, page.py:479-506)."""
    return [float((text_region.region_type or "paragraph").lower() == 'heading')]


# ---------------------------------------------------------------------------------------------------------------
# separator rules
# ---------------------------------------------------------------------------------------------------------------
def _sep_orientation(separator_region, bb_sep):
    orientation = separator_region.get_orientation()
    if orientation is None:
        width = max(bb_sep[1] - bb_sep[0], 1)
        height = max(bb_sep[3] - bb_sep[2], 1)
        orientation = "horizontal" if float(height) / float(width) < 5 else "vertical"
    return orientation


def is_vertically_separated(min_x_a, max_x_a, min_y_a, max_y_a, min_x_b, max_x_b, min_y_b, max_y_b,
                            min_x_sep, max_x_sep, min_y_sep, max_y_sep):
    """:352-366."""
    mean_x_sep = (min_x_sep + max_x_sep) / 2
    if not ((max_x_a <= mean_x_sep <= min_x_b) or (max_x_b <= mean_x_sep <= min_x_a)):
        return False
    if not ((max_y_a >= min_y_sep and min_y_a <= max_y_sep) or (max_y_b >= min_y_sep and min_y_b <= max_y_sep)):
        return False
    return True


def is_horizontally_separated(min_x_a, max_x_a, min_y_a, max_y_a, min_x_b, max_x_b, min_y_b, max_y_b,
                              min_x_sep, max_x_sep, min_y_sep, max_y_sep):
    """:369-384."""
    mean_y_sep = (min_y_sep + max_y_sep) / 2
    if not ((min_y_a <= mean_y_sep <= max_y_b) or (min_y_b <= mean_y_sep <= max_y_a)):
        return False
    if ((max_x_a <= min_x_sep and max_x_b <= min_x_sep) or (min_x_a >= max_x_sep and min_x_b >= max_x_sep)):
        return False
    return True


def get_edge_separator_feature_bb(text_region_a, text_region_b, separator_regions):
    """:319-349."""
    bb_a = get_bounding_box(np.asarray(text_region_a.points, dtype=np.int32))
    bb_b = get_bounding_box(np.asarray(text_region_b.points, dtype=np.int32))
    horizontally_separated = vertically_separated = False
    for separator_region in separator_regions:
        bb_sep = get_bounding_box(np.asarray(separator_region.points, dtype=np.int32))
        if _sep_orientation(separator_region, bb_sep) == "vertical":
            if is_vertically_separated(*bb_a, *bb_b, *bb_sep):
                vertically_separated = True
        else:
            if is_horizontally_separated(*bb_a, *bb_b, *bb_sep):
                horizontally_separated = True
        if horizontally_separated and vertically_separated:
            break
    return [float(horizontally_separated), float(vertically_separated)]


def get_edge_separator_feature_line(text_region_a, text_region_b, separator_regions):
    """:219-290.  Note the reference compares the *region object* with 'vertical' (``:272``), so a separator tagged
    vertical falls through to the ratio check; kept as is."""
    min_x_a, max_x_a, min_y_a, max_y_a = get_bounding_box(np.asarray(text_region_a.points, dtype=np.int32))
    min_x_b, max_x_b, min_y_b, max_y_b = get_bounding_box(np.asarray(text_region_b.points, dtype=np.int32))
    segment = (((min_x_a + max_x_a) / 2, (min_y_a + max_y_a) / 2), ((min_x_b + max_x_b) / 2, (min_y_b + max_y_b) / 2))
    horizontally_separated = vertically_separated = False
    for separator_region in separator_regions:
        min_x_s, max_x_s, min_y_s, max_y_s = get_bounding_box(np.asarray(separator_region.points, dtype=np.int32))
        width = max(max_x_s - min_x_s, 1)
        height = max(max_y_s - min_y_s, 1)
        ratio = float(height) / float(width)
        corners = [(min_x_s, min_y_s), (max_x_s, min_y_s), (min_x_s, max_y_s), (max_x_s, max_y_s)]
        if line_poly_intersection(segment, corners) or line_in_bounding_box(segment, min_x_s, max_x_s, min_y_s, max_y_s):
            if line_poly_intersection(segment, separator_region.points):
                if separator_region.get_orientation() == 'horizontal':
                    horizontally_separated = True
                elif ratio < 5:
                    horizontally_separated = True
                else:
                    vertically_separated = True
                if horizontally_separated and vertically_separated:
                    break
    return [float(horizontally_separated), float(vertically_separated)]


def is_aligned_horizontally_separated(text_region_a, text_region_b, separator_regions):
    """:387-437 (confidence masking, run_gnn_clustering.py:180)."""
    min_x_a, max_x_a, min_y_a, max_y_a = get_bounding_box(np.asarray(text_region_a.points, dtype=np.int32))
    min_x_b, max_x_b, min_y_b, max_y_b = get_bounding_box(np.asarray(text_region_b.points, dtype=np.int32))
    for separator_region in separator_regions:
        bb = get_bounding_box(np.asarray(separator_region.points, dtype=np.int32))
        min_x_s, max_x_s, min_y_s, max_y_s = bb
        if _sep_orientation(separator_region, bb) == 'vertical':
            continue
        mean_y_sep = (min_y_s + max_y_s) / 2
        if not ((min_y_a <= mean_y_sep <= max_y_b) or (min_y_b <= mean_y_sep <= max_y_a)):
            continue
        if not ((max_x_a >= min_x_s and max_x_b >= min_x_s) and (min_x_a <= max_x_s and min_x_b <= max_x_s)):
            continue
        return True
    return None


def is_aligned_heading_separated(text_region_a, text_region_b):
    """:440-470."""
    heading_a = (text_region_a.region_type or "").lower() == 'heading'
    heading_b = (text_region_b.region_type or "").lower() == 'heading'
    if heading_a and heading_b:
        return False
    if not (heading_a or heading_b):
        return False
    min_x_a, max_x_a, min_y_a, max_y_a = get_bounding_box(np.asarray(text_region_a.points, dtype=np.int32))
    min_x_b, max_x_b, min_y_b, max_y_b = get_bounding_box(np.asarray(text_region_b.points, dtype=np.int32))
    if not (min_x_a <= max_x_b and min_x_b <= max_x_a):
        return False
    if heading_a and not (min_y_a >= max_y_b):
        return False
    if heading_b and not (min_y_b >= max_y_a):
        return False
    return True


def mask_horizontally_separated_confs(confs, page_path, mask_heading=True, mask_horizontal=True):
    """run_gnn_clustering.py:151-187: zero the confidences of region pairs of one column that are split by a heading
    or by a horizontal separator."""
    regions = Page(page_path).get_regions()
    if mask_horizontal and 'SeparatorRegion' not in regions:
        logging.warning("No separators found for confidence masking.")
        return confs
    text_regions = regions['TextRegion']
    separator_regions = regions.get('SeparatorRegion', [])
    n = len(text_regions)
    masked = np.ones_like(confs, dtype=np.int32)
    for i in range(n):
        for j in range(i + 1, n):
            if mask_heading and is_aligned_heading_separated(text_regions[i], text_regions[j]):
                masked[i, j] = masked[j, i] = 0
                continue
            if mask_horizontal and is_aligned_horizontally_separated(text_regions[i], text_regions[j],
                                                                     separator_regions):
                masked[i, j] = masked[j, i] = 0
    return masked * confs


# ---------------------------------------------------------------------------------------------------------------
# edges, visual regions
# ---------------------------------------------------------------------------------------------------------------
def fully_connected_edges(num_nodes):
    """:494-509: all ordered pairs without self loops, row major."""
    idx = np.tile(np.arange(num_nodes, dtype=np.int32), [num_nodes, 1])
    pairs = np.stack([idx.T, idx], axis=2).reshape([-1, 2])
    return np.delete(pairs, np.arange(num_nodes) * (num_nodes + 1), axis=0)


def delaunay_edges(num_nodes, node_positions):
    """:512-538: neighbours in the Delaunay triangulation of the positions rounded to 50 px."""
    from scipy.spatial import Delaunay
    try:
        from scipy.spatial import QhullError
    except ImportError:  # pragma: no cover - older scipy
        from scipy.spatial.qhull import QhullError
    smooth = round_by_precision_and_base(node_positions, base=50)
    try:
        delaunay = Delaunay(smooth)
    except QhullError:
        logging.warning("Delaunay input has the same x-coords. Defaulting to unsmoothed data.")
        delaunay = Delaunay(node_positions)
    indptr, indices = delaunay.vertex_neighbor_vertices
    out = []
    for v in range(num_nodes):
        nb = indices[indptr[v]:indptr[v + 1]]
        out.append(np.stack(np.broadcast_arrays(v, nb), axis=1))
    return np.concatenate(out, axis=0)


def get_node_visual_region(text_region):
    return bounding_box(text_region.points)


def get_edge_visual_region(text_region_a, text_region_b):
    return convex_hull(list(text_region_a.points) + list(text_region_b.points))


# ---------------------------------------------------------------------------------------------------------------
# page -> arrays -> json
# ---------------------------------------------------------------------------------------------------------------
_NONE11 = (None,) * 11


def build_input_and_target(page_path, interaction='delaunay', visual_regions=False, external_data=None,
                           sim_feat_extractor=None, separators='bb', swt_img=None, device=0):
    """:593-813."""
    assert interaction in ('fully', 'delaunay'), \
        f"Interaction setup {interaction} is not supported. Choose from ('fully', 'delaunay') instead."
    if sim_feat_extractor is not None:
        raise NotImplementedError("word-vector text block similarities are not part of this build")
    page = Page(page_path)
    regions = page.get_regions()
    text_lines = page.get_textlines()
    resolution = page.get_image_resolution()
    norm_x, norm_y = float(resolution[0]), float(resolution[1])
    if 'TextRegion' not in regions:
        logging.warning(f'No TextRegions found in {page_path}. Returning None.')
        return _NONE11
    text_regions = regions['TextRegion']
    num_nodes = len(text_regions)
    if num_nodes <= 1:
        logging.warning(f'Less than two nodes found in {page_path}. Returning None.')
        return _NONE11

    stroke_widths, heights = get_textline_stroke_widths_heights_dist_trafo(page_path, text_lines, swt_img=swt_img,
                                                                           device=device)
    sw_max = np.max(list(stroke_widths.values()))
    th_max = np.max(list(heights.values()))

    node_features = []
    for text_region in text_regions:
        f = []
        f.extend(get_text_region_geometric_features(text_region, norm_x, norm_y))
        f.extend(get_text_region_baseline_features(text_region, norm_x, norm_y))
        f.extend(get_text_region_stroke_width_feature(text_region, stroke_widths, norm=sw_max))
        f.extend(get_text_region_text_height_feature(text_region, heights, norm=th_max))
        f.extend(get_text_region_heading_feature(text_region))
        if external_data:
            for ext in external_data:
                ext_page = ext.get(os.path.basename(page_path))
                if ext_page is None:
                    logging.warning(f'Could not find key {os.path.basename(page_path)} in external data json.')
                    continue
                if 'node_features' in ext_page:
                    nf = ext_page['node_features']
                    if text_region.id in nf:
                        f.extend(nf[text_region.id])
                    elif 'default' in nf:
                        f.extend([nf['default']])
                    else:
                        f.extend([0.0])
        node_features.append(f)

    if interaction == 'fully' or num_nodes < 4:
        interacting_nodes = fully_connected_edges(num_nodes)
    else:
        node_centers = np.array(node_features, dtype=np.float32)[:, 2:4] * [norm_x, norm_y]
        interacting_nodes = delaunay_edges(num_nodes, node_centers)
    num_interacting_nodes = interacting_nodes.shape[0]

    separator_regions = regions.get('SeparatorRegion')
    edge_features = []
    for i in range(num_interacting_nodes):
        a, b = text_regions[interacting_nodes[i, 0]], text_regions[interacting_nodes[i, 1]]
        ef = []
        if separator_regions:
            if separators == 'line':
                ef.extend(get_edge_separator_feature_line(a, b, separator_regions))
            else:
                ef.extend(get_edge_separator_feature_bb(a, b, separator_regions))
        else:
            ef.extend([0.0, 0.0])
        if external_data:
            for ext in external_data:
                ext_page = ext.get(os.path.basename(page_path))
                if ext_page is None:
                    continue
                if 'edge_features' in ext_page:
                    try:
                        ef.extend(ext_page['edge_features'][a.id][b.id])
                    except (KeyError, TypeError):
                        ef.extend(ext_page['edge_features'].get('default', [0.5]))
        edge_features.append(ef)

    vr_nodes, np_nodes, np_edges, vr_edges_array = [], [], [], None
    if visual_regions:
        for text_region in text_regions:
            vr = get_node_visual_region(text_region)
            vr_nodes.append(vr)
            np_nodes.append(len(vr))
        vr_edges = []
        for i in range(num_interacting_nodes):
            vr = get_edge_visual_region(text_regions[interacting_nodes[i, 0]], text_regions[interacting_nodes[i, 1]])
            vr_edges.append(vr)
            np_edges.append(len(vr))
        vr_edges_array = np.zeros((num_interacting_nodes, np.max(np_edges), 2))
        for i, vr in enumerate(vr_edges):
            vr_edges_array[i, :len(vr), :] = vr

    # ground truth: majority article id per region (:762-790); ties -> first of list(set(...)) like the reference
    tr_ids = []
    for text_region in text_regions:
        ids = [tl.get_article_id() for tl in text_region.text_lines]
        uniq = list(set(ids))
        occ = np.array([ids.count(a) for a in uniq], dtype=np.int32)
        tr_ids.append(uniq[int(np.argmax(occ))] if occ.shape[0] > 1 else uniq[0])
    gt_relations = [[1, i, j] for i, a in enumerate(tr_ids) for j, b in enumerate(tr_ids) if a == b]

    return (np.array(num_nodes, dtype=np.int32), interacting_nodes.astype(np.int32),
            np.array(num_interacting_nodes, dtype=np.int32), np.array(node_features, dtype=np.float32),
            np.array(edge_features, dtype=np.float32) if edge_features else None,
            np.transpose(np.array(vr_nodes, dtype=np.float32), axes=(0, 2, 1)) if visual_regions else None,
            np.array(np_nodes, dtype=np.int32) if visual_regions else None,
            np.transpose(vr_edges_array, axes=(0, 2, 1)) if visual_regions else None,
            np.array(np_edges, dtype=np.int32) if visual_regions else None,
            np.array(gt_relations, dtype=np.int32), np.array(len(gt_relations), dtype=np.int32))


def generate_feature_jsons(page_paths, out_path=None, interaction="delaunay", visual_regions=True, json_list=None,
                           tb_similarity_setup=(None, None), separators='line', device=0):
    """:816-911: one ``<name>.json`` per PAGE-XML; default folder ``json<nodeDim><i><edgeDim>[v]<separators>`` next to
    the ``page`` folder."""
    json_data = []
    for json_path in json_list or []:
        with open(json_path) as f:
            json_data.append(json.load(f))
    if tb_similarity_setup[0] and tb_similarity_setup[1]:
        raise NotImplementedError("word-vector text block similarities are not part of this build")
    create_default_dir = not out_path
    skipped, written = [], []
    t0 = time.time()
    for page_path in page_paths:
        logging.info(f"Processing... {page_path}")
        (num_nodes, interacting_nodes, num_interacting_nodes, node_features, edge_features, vr_nodes, np_nodes,
         vr_edges, np_edges, gt_relations, gt_num_relations) = build_input_and_target(
            page_path=page_path, interaction=interaction, visual_regions=visual_regions, external_data=json_data,
            separators=separators, device=device)
        if num_nodes is None:
            skipped.append(page_path)
            continue
        out = {"num_nodes": num_nodes.tolist(), "interacting_nodes": interacting_nodes.tolist(),
               "num_interacting_nodes": num_interacting_nodes.tolist(), "node_features": node_features.tolist(),
               "edge_features": edge_features.tolist()}
        if vr_nodes is not None and np_nodes is not None:
            out["visual_regions_nodes"] = vr_nodes.tolist()
            out["num_points_visual_regions_nodes"] = np_nodes.tolist()
        if vr_edges is not None and np_edges is not None:
            out["visual_regions_edges"] = vr_edges.tolist()
            out["num_points_visual_regions_edges"] = np_edges.tolist()
        out["gt_relations"] = gt_relations.tolist()
        out["gt_num_relations"] = gt_num_relations.tolist()
        if create_default_dir:
            visual = 'v' if visual_regions else ''
            out_path = re.sub(r'page$', f'json{node_features.shape[1]}{interaction[0]}{edge_features.shape[1]}'
                                        f'{visual}{separators}', os.path.dirname(page_path))
        os.makedirs(out_path, exist_ok=True)
        target = os.path.join(out_path, os.path.splitext(os.path.basename(page_path))[0] + ".json")
        with open(target, "w") as f:
            json.dump(out, f)
        written.append(target)
    logging.info(f"Time (feature generation): {time.time() - t0:.2f} seconds")
    logging.info(f"Wrote {len(written)}/{len(page_paths)} files; skipped {len(skipped)}.")
    return written
