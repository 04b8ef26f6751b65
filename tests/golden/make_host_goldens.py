"""Generates tests/golden/host_goldens.json: the reference's numpy-only HOST functions around the two nets, run on seeded duck-typed pages
(VERDICT r5 next #5).  The reference is imported in the build container with placeholder modules for TensorFlow / OpenCV / lxml / shapely /
gensim ... (ref_import.install_stubs); every function below touches numpy and plain Python only.

Pinned here (reference file:line -> product function that must reproduce every case exactly, tests/test_host_goldens.py):
  gnn/input/feature_generation.py:18-81    get_text_region_geometric_features / _baseline_features     -> feature_generation.py (f4)
  :162-218                                  stroke-width / text-height / heading features of a region   -> feature_generation.py (f4)
  :319-398                                  get_edge_separator_feature_bb, is_vertically / is_horizontally_separated (f4)
  :401-471                                  is_aligned_horizontally_separated, is_aligned_heading_separated (a20 masking, f4)
  :474-491                                  get_node_visual_region, get_edge_visual_region (absolute pixels: the reference's quirk)
  :494-535                                  fully_connected_edges, delaunay_edges (scipy)
  gnn/run_gnn_clustering.py:151-186         mask_horizontally_separated_confs (a20)  -> feature_generation.mask_horizontally_separated_confs
  gnn/input/input_dataset.py:343-375,444-457  get_input_and_target_from_json, build_full_relations (a13)  -> gnn_input.py
  image_segmentation/net_post_processing/region_net_post_processor_base.py:253-268  rescale_polygons (a10) -> polygon helpers
  .../heading_net_post_processor.py:50-63,65-200,247-270  scale_to_new_interval, the fusion rule of to_page_xml (driven with recorded per-line
                                            measurements and a recording PAGE writer), get_net_prob_for_text_line (a11)

Run:  python tests/golden/make_host_goldens.py
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

ref_import.install_stubs()

from article_separation.gnn.input import feature_generation as rfg  # noqa: E402
from article_separation.gnn import run_gnn_clustering as rrun  # noqa: E402
from article_separation.gnn.input import input_dataset as rin  # noqa: E402
from article_separation.image_segmentation.net_post_processing import heading_net_post_processor as rhead  # noqa: E402
from article_separation.image_segmentation.net_post_processing.region_net_post_processor_base import RegionNetPostProcessor  # noqa: E402
from python_util.parser.xml.page.page_objects import Points  # noqa: E402


# ---- duck-typed page objects (what the functions read: points.points_list, text_lines, baseline, text, id, region_type, get_orientation) ----
class Pts:
    def __init__(self, pts):
        self.points_list = [tuple(int(v) for v in p) for p in pts]


class Line:
    def __init__(self, lid, text, baseline, surr=None):
        self.id, self.text, self.baseline = lid, text, Pts(baseline)
        self.surr_p = Points([tuple(p) for p in surr]) if surr is not None else None
        self.custom = {}


class Region:
    def __init__(self, rid, pts, lines, region_type):
        self.id, self.points, self.text_lines, self.region_type = rid, Pts(pts), lines, region_type


class Sep:
    def __init__(self, pts, orientation):
        self.points, self._o = Pts(pts), orientation

    def get_orientation(self):
        return self._o


def rect(x0, y0, x1, y1):
    return [(x0, y0), (x1, y0), (x1, y1), (x0, y1)]


def make_page(rng, n_regions, n_seps, W=3000, H=4500):
    """a column layout with jitter: text blocks in 2-5 columns, some headings, horizontal / vertical / untyped separators, a few polygons"""
    ncol = int(rng.integers(2, 6))
    colw = W // ncol
    regions, spec = [], []
    for i in range(n_regions):
        col = int(rng.integers(0, ncol))
        x0 = col * colw + int(rng.integers(5, 60))
        x1 = min(W - 1, x0 + int(rng.integers(colw // 3, colw - 40)))
        y0 = int(rng.integers(20, H - 400))
        y1 = y0 + int(rng.integers(30, 380))
        if rng.random() < 0.25:                               # a six-point outline instead of a rectangle
            xm = (x0 + x1) // 2
            pts = [(x0, y0), (xm, y0 - int(rng.integers(0, 15))), (x1, y0), (x1, y1), (xm, y1 + int(rng.integers(0, 15))), (x0, y1)]
        else:
            pts = rect(x0, y0, x1, y1)
        nl = int(rng.integers(1, 5))
        lines = []
        for k in range(nl):
            by = y0 + (k + 1) * (y1 - y0) // (nl + 1)
            text = "" if rng.random() < 0.15 else f"t{i}_{k}"
            bl = [(x0 + 3, by), ((x0 + x1) // 2, by + int(rng.integers(-3, 4))), (x1 - 3, by)]
            lines.append((f"r{i}l{k}", text, bl))
        rtype = "heading" if rng.random() < 0.3 else ("paragraph" if rng.random() < 0.8 else "Heading")
        spec.append({"id": f"r{i}", "points": [list(p) for p in pts], "lines": [[lid, t, [list(p) for p in bl]] for lid, t, bl in lines], "type": rtype})
        regions.append(Region(f"r{i}", pts, [Line(lid, t, bl) for lid, t, bl in lines], rtype))
    seps, sspec = [], []
    for s in range(n_seps):
        kind = rng.random()
        if kind < 0.45:                                       # horizontal rule
            x0 = int(rng.integers(0, W - 300)); x1 = x0 + int(rng.integers(200, W - x0)); y0 = int(rng.integers(10, H - 20)); y1 = y0 + int(rng.integers(2, 12))
        elif kind < 0.9:                                      # vertical rule
            x0 = int(rng.integers(10, W - 20)); x1 = x0 + int(rng.integers(2, 12)); y0 = int(rng.integers(0, H - 400)); y1 = y0 + int(rng.integers(200, H - y0))
        else:                                                 # a blob: the ratio rule decides
            x0 = int(rng.integers(10, W - 200)); x1 = x0 + int(rng.integers(5, 150)); y0 = int(rng.integers(10, H - 500)); y1 = y0 + int(rng.integers(5, 450))
        o = [None, "horizontal", "vertical"][int(rng.integers(0, 3))]
        pts = rect(x0, y0, x1, y1)
        seps.append(Sep(pts, o))
        sspec.append({"points": [list(p) for p in pts], "orientation": o})
    return regions, seps, spec, sspec


def floats(v):
    return [float(x) for x in v]


def main():
    out = {"pages": [], "bbox_rules": [], "edges": [], "json_feed": [], "rescale": [], "heading": [], "net_prob": [], "scale_interval": []}
    rng = np.random.default_rng(20260)

    # ---- pages: node features, pair rules, masks ----
    for pidx, (n, ns) in enumerate([(6, 3), (12, 8), (25, 14), (9, 0), (40, 25), (3, 5)]):
        regions, seps, spec, sspec = make_page(rng, n, ns)
        W, H = 3000, 4500
        sw = {ln.id: float(rng.integers(1, 9)) + (0.5 if rng.random() < 0.3 else 0.0) for r in regions for ln in r.text_lines}
        th = {ln.id: int(rng.integers(8, 60)) for r in regions for ln in r.text_lines}
        rec = {"regions": spec, "separators": sspec, "norm": [W, H], "stroke_widths": sw, "heights": th, "nodes": [], "pairs": []}
        for r in regions:
            rec["nodes"].append({
                "geometric": floats(rfg.get_text_region_geometric_features(r, W, H)),
                "baseline": floats(rfg.get_text_region_baseline_features(r, W, H)),
                "stroke_width": floats(rfg.get_text_region_stroke_width_feature(r, sw, norm=7.0)),
                "text_height": floats(rfg.get_text_region_text_height_feature(r, th, norm=55.0)),
                "heading": floats(rfg.get_text_region_heading_feature(r)),
                "visual_region": np.asarray(rfg.get_node_visual_region(r)).tolist(),
            })
        pair_idx = [(i, j) for i in range(n) for j in range(n) if i != j]
        if len(pair_idx) > 220:
            pair_idx = [pair_idx[k] for k in rng.choice(len(pair_idx), 220, replace=False)]
        for i, j in pair_idx:
            a, b = regions[i], regions[j]
            rec["pairs"].append({
                "i": i, "j": j,
                "separator_bb": floats(rfg.get_edge_separator_feature_bb(a, b, seps)),
                "aligned_horizontally_separated": bool(rfg.is_aligned_horizontally_separated(a, b, seps)),
                "aligned_heading_separated": bool(rfg.is_aligned_heading_separated(a, b)),
                "edge_visual_region": np.asarray(rfg.get_edge_visual_region(a, b)).tolist(),
            })
        # mask_horizontally_separated_confs: the reference reads the page through Page(path); here its two region lists are handed over directly
        confs = rng.random((n, n)).astype(np.float32)

        class FakePage:
            def __init__(self, path):
                pass

            def get_regions(self, _r=regions, _s=seps):
                d = {"TextRegion": _r}
                if _s:
                    d["SeparatorRegion"] = _s
                return d
        rrun.Page = FakePage
        masks = {}
        for mh, ms in ((True, True), (True, False), (False, True)):
            rrun.FLAGS = type("F", (), {"mask_heading_separated_confs": mh, "mask_horizontally_separated_confs": ms})()
            key = f"heading={int(mh)},horizontal={int(ms)}"
            try:                                              # (a page without separators: early return with the flag, KeyError at :160 without it)
                res = rrun.mask_horizontally_separated_confs(confs.copy(), "unused")
                masks[key] = {"dtype": str(res.dtype), "values": np.asarray(res, np.float64).tolist()}
            except KeyError as e:
                masks[key] = {"raises": "KeyError", "arg": str(e.args[0])}
        rec["confs"] = confs.astype(np.float64).tolist()
        rec["masked"] = masks
        out["pages"].append(rec)

    # ---- the two bounding-box rules on raw boxes (incl. touching / equal coordinates) ----
    for _ in range(400):
        v = rng.integers(0, 12, size=12) * 10
        a = [int(min(v[0], v[1])), int(max(v[0], v[1])), int(min(v[2], v[3])), int(max(v[2], v[3]))]
        b = [int(min(v[4], v[5])), int(max(v[4], v[5])), int(min(v[6], v[7])), int(max(v[6], v[7]))]
        s = [int(min(v[8], v[9])), int(max(v[8], v[9])), int(min(v[10], v[11])), int(max(v[10], v[11]))]
        out["bbox_rules"].append({"a": a, "b": b, "s": s, "vertical": bool(rfg.is_vertically_separated(*a, *b, *s)),
                                  "horizontal": bool(rfg.is_horizontally_separated(*a, *b, *s))})

    # ---- edge sets ----
    for n in (1, 2, 3, 7, 20):
        out["edges"].append({"kind": "full", "n": n, "edges": rfg.fully_connected_edges(n).tolist()})
    for n in (4, 9, 30, 120):
        pos = np.stack([rng.integers(0, 3000, n), rng.integers(0, 4500, n)], axis=1).astype(np.float64)
        e = rfg.delaunay_edges(n, pos)
        out["edges"].append({"kind": "delaunay", "n": n, "positions": pos.tolist(), "edges": np.asarray(e).tolist()})

    # ---- json -> feed arrays, full relation list ----
    with tempfile.TemporaryDirectory() as tmp:
        for k, (n, e, vis) in enumerate([(5, 8, False), (11, 40, True), (2, 1, True)]):
            data = {"num_nodes": n, "interacting_nodes": rng.integers(0, n, (e, 2)).tolist(), "num_interacting_nodes": e,
                    "node_features": rng.random((n, 15)).round(6).tolist(), "edge_features": rng.random((e, 2)).round(6).tolist(),
                    "gt_relations": [[1, int(a), int(b)] for a, b in rng.integers(0, n, (max(1, n // 2), 2))], "gt_num_relations": max(1, n // 2)}
            if vis:
                data["visual_regions_nodes"] = rng.integers(0, 3000, (n, 2, 4)).tolist()
                data["num_points_visual_regions_nodes"] = [4] * n
                data["visual_regions_edges"] = rng.integers(0, 3000, (e, 2, 6)).tolist()
                data["num_points_visual_regions_edges"] = [6] * e
            p = os.path.join(tmp, f"g{k}.json")
            json.dump(data, open(p, "w"))
            got = rin.get_input_and_target_from_json(p)
            rel, nrel, relgt = rin.build_full_relations(n, got["gt_relations"])
            out["json_feed"].append({"json": data, "arrays": {kk: {"dtype": str(v.dtype), "shape": list(v.shape), "values": v.tolist()} for kk, v in got.items()},
                                     "relations": rel.tolist(), "num_relations": int(nrel), "relations_gt": relgt.tolist()})

    # ---- rescale_polygons (region_net_post_processor_base.py:253-268) ----
    for sf in (1.0, 0.5, 2.0, 1.0 / 3.0, 1.5, 0.37):
        polys = {"SeparatorRegion_horizontal": [[[[int(a), int(b)] for a, b in rng.integers(0, 4000, (5, 2))], [[int(a), int(b)] for a, b in rng.integers(0, 4000, (4, 2))]]
                                                for _ in range(3)],
                 "SeparatorRegion_vertical": [[[[int(a), int(b)] for a, b in rng.integers(0, 4000, (7, 2))]]]}
        src = json.loads(json.dumps(polys))
        res = RegionNetPostProcessor.rescale_polygons(None, polys, sf)
        out["rescale"].append({"scaling_factor": sf, "polygons": src, "rescaled": json.loads(json.dumps(res, default=lambda o: [int(v) for v in o]))})

    # ---- heading post-processor: scale_to_new_interval, get_net_prob_for_text_line, the fusion rule of to_page_xml ----
    hp = object.__new__(rhead.HeadingNetPostProcessor)
    for d, lo, hi in [(5, 0, 10), (3.5, 3.5, 3.5), (-2, -4, 6), (0.25, 0, 1), (7, 7, 9)]:
        out["scale_interval"].append({"data": d, "old_min": lo, "old_max": hi, "value": float(hp.scale_to_new_interval(d, lo, hi))})
    net = np.round(rng.random((150, 110)) * (rng.random((150, 110)) < 0.3), 3).astype(np.float64)   # (lines reach beyond it: slices clip)
    for sf in (1.0, 0.5, 0.25):
        for _ in range(12):
            x0, y0 = int(rng.integers(0, 160)), int(rng.integers(0, 260))
            surr = [(x0, y0), (x0 + int(rng.integers(5, 200)), y0 + int(rng.integers(-4, 5))), (x0 + int(rng.integers(5, 200)), y0 + int(rng.integers(6, 60))),
                    (x0 - int(rng.integers(0, 4)), y0 + int(rng.integers(6, 60)))]
            surr = [(max(0, a), max(0, b)) for a, b in surr]
            ln = Line("x", "t", [(0, 0), (1, 1)], surr)
            v = hp.get_net_prob_for_text_line(net, ln, sf)
            out["net_prob"].append({"scaling_factor": sf, "surr_p": [list(p) for p in surr], "value": None if not np.isfinite(v) else float(v)})
    out["net_prob_map"] = net.tolist()
    out["net_prob"].append({"scaling_factor": 1.0, "surr_p": None, "value": float(hp.get_net_prob_for_text_line(net, Line("n", "t", [(0, 0), (1, 1)], None), 1.0))})

    # the fusion rule: to_page_xml (:65-200) with recorded measurements in place of the image stages and a PAGE writer that records what is set
    for case in range(10):
        n_reg = int(rng.integers(1, 7))
        reg_lines, all_lines = [], []
        for r in range(n_reg):
            nl = int(rng.integers(0, 5))
            ls = [Line(f"r{r}l{k}", "t", [(0, 0), (1, 1)], rect(10, 10, 60, 30) if rng.random() < 0.9 else None) for k in range(nl)]
            reg_lines.append(ls)
            all_lines.extend(ls)
        meas = {ln.id: (float(rng.integers(1, 8)) + (0.5 if rng.random() < 0.2 else 0.0), int(rng.integers(10, 70)), float(rng.random())) for ln in all_lines}
        weight = [{"net": 0.33, "stroke_width": 0.33, "text_height": 0.33}, {"net": 0.8, "stroke_width": 0.0, "text_height": 0.2}, {"net": 0.0, "stroke_width": 0.5, "text_height": 0.5}][case % 3]
        thresh = [{"net_thresh": 0.9, "stroke_width_thresh": 0.9, "text_height_thresh": 0.9, "sw_th_thresh": 0.8},
                  {"net_thresh": 1.0, "stroke_width_thresh": 1.0, "text_height_thresh": 0.9, "sw_th_thresh": 0.9}][case % 2]
        threshold, tlp = [0.5, 0.4, 0.7][case % 3], [1.0, 0.8, 0.5][case % 3]

        class FakeNode:
            def __init__(self, obj):
                self.obj, self.attrs = obj, {}

            def set(self, k, v):
                self.attrs[k] = v

        class FakePageObject:
            page_doc = None

            def __init__(self):
                self.regions = [type("R", (), {"id": f"r{r}", "text_lines": ls})() for r, ls in enumerate(reg_lines)]
                self.nodes = {}

            def get_textlines(self):
                return all_lines

            def get_text_regions(self):
                return self.regions

            def get_child_by_id(self, doc, cid):
                obj = next((x for x in all_lines + self.regions if x.id == cid))
                return [self.nodes.setdefault(cid, FakeNode(obj))]

            def set_custom_attr(self, node, key, sub, value):
                node.obj.custom.setdefault(key, {})[sub] = value          # (what page.py does to the TextLine the region keeps)

        class FakeWriter:
            def __init__(self, *a, **k):
                self.page_object, self.scaling_factor = FakePageObject(), 1.0

            def save_page_xml(self, path):
                pass
        rhead.RegionToPageWriter = FakeWriter
        hp = object.__new__(rhead.HeadingNetPostProcessor)
        hp.fixed_height, hp.scaling_factor, hp.weight_dict, hp.thresh_dict, hp.threshold, hp.text_line_percentage = 0, 1.0, weight, thresh, threshold, tlp
        hp.get_swt_features_image = lambda image_path: "swt"
        hp.get_swt_features_textline = lambda swt, tl: meas[tl.id][:2]
        hp.get_net_prob_for_text_line = lambda netp, tl, sf: meas[tl.id][2]
        for ln in all_lines:
            ln.custom = {}
        page = hp.to_page_xml("p.xml", image_path="i.png", net_output_post="net")
        heads = sorted(ln.id for ln in all_lines if ln.custom.get("structure", {}).get("semantic_type") == "heading")
        rtypes = {rid: nd.attrs.get("type") for rid, nd in page.nodes.items() if rid.startswith("r") and "l" not in rid}
        out["heading"].append({"regions": [[ln.id for ln in ls] for ls in reg_lines], "has_outline": {ln.id: ln.surr_p is not None for ln in all_lines},
                               "measurements": {k: list(v) for k, v in meas.items()}, "weight_dict": weight, "thresh_dict": thresh, "threshold": threshold,
                               "text_line_percentage": tlp, "heading_lines": heads, "region_types": rtypes})

    path = os.path.join(HERE, "host_goldens.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes;", {k: len(v) for k, v in out.items() if isinstance(v, list)})


if __name__ == "__main__":
    main()
