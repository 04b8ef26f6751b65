"""The host functions around the two nets against what the REFERENCE computes: tests/golden/host_goldens.json was written by
tests/golden/make_host_goldens.py, which imports /root/reference in the build container (placeholder modules for TensorFlow / OpenCV / lxml /
shapely ...) and runs its numpy-only functions on seeded duck-typed pages.  The product's restatements must reproduce EVERY case exactly --
rows a10 (rescale_polygons), a11 (heading fusion rule, per-line net confidence), a13 (json -> feed arrays, relations), a20 (confidence masks)
and f4 (node / edge features, edge sets) of SURVEY section 8.  CPU only; nothing of the reference is needed to run it."""
import json
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

G = json.load(open(os.path.join(ROOT, "tests", "golden", "host_goldens.json")))


class _Line:
    def __init__(self, lid, text, baseline):
        self.id, self.text, self.baseline = lid, text, [tuple(p) for p in baseline]


class _Region:
    def __init__(self, spec):
        self.id, self.points, self.region_type = spec["id"], [tuple(p) for p in spec["points"]], spec["type"]
        self.text_lines = [_Line(*ln) for ln in spec["lines"]]


class _Sep:
    def __init__(self, spec):
        self.points, self._o = [tuple(p) for p in spec["points"]], spec["orientation"]

    def get_orientation(self):
        return self._o


def _page(rec):
    return [_Region(s) for s in rec["regions"]], [_Sep(s) for s in rec["separators"]]


@pytest.mark.parametrize("k", range(len(G["pages"])))
def test_node_and_pair_features_of_a_page(k):
    """feature_generation.py:18-81,162-218,319-491 of the reference on the same regions: exact floats (same operations in the same order)"""
    from citlab_article_separation_new_amd import feature_generation as fg
    rec = G["pages"][k]
    regions, seps = _page(rec)
    W, H = rec["norm"]
    for r, want in zip(regions, rec["nodes"]):
        assert [float(v) for v in fg.get_text_region_geometric_features(r, W, H)] == want["geometric"]
        assert [float(v) for v in fg.get_text_region_baseline_features(r, W, H)] == want["baseline"]
        assert [float(v) for v in fg.get_text_region_stroke_width_feature(r, rec["stroke_widths"], norm=7.0)] == want["stroke_width"]
        assert [float(v) for v in fg.get_text_region_text_height_feature(r, rec["heights"], norm=55.0)] == want["text_height"]
        assert [float(v) for v in fg.get_text_region_heading_feature(r)] == want["heading"]
        assert np.asarray(fg.get_node_visual_region(r)).tolist() == want["visual_region"]        # absolute pixels, as the reference writes them
    for pr in rec["pairs"]:
        a, b = regions[pr["i"]], regions[pr["j"]]
        assert fg.get_edge_separator_feature_bb(a, b, seps) == pr["separator_bb"], (pr["i"], pr["j"])
        assert bool(fg.is_aligned_horizontally_separated(a, b, seps)) == pr["aligned_horizontally_separated"]
        assert bool(fg.is_aligned_heading_separated(a, b)) == pr["aligned_heading_separated"]
        assert np.asarray(fg.get_edge_visual_region(a, b)).tolist() == pr["edge_visual_region"], (pr["i"], pr["j"])


@pytest.mark.parametrize("k", range(len(G["pages"])))
def test_confidence_masks_of_a_page(k, monkeypatch):
    """run_gnn_clustering.py:151-186: values AND dtype (the int32 mask times the float32 confidences is a float64 matrix: what the clustering
    then sees).  The reference fails with a KeyError on a page without separators when only the heading mask is asked for (:160 reads
    regions['SeparatorRegion'] unconditionally); the product masks the headings there -- checked against the pair rule the reference pins."""
    from citlab_article_separation_new_amd import feature_generation as fg
    rec = G["pages"][k]
    regions, seps = _page(rec)
    confs = np.asarray(rec["confs"], np.float64).astype(np.float32)

    class FakePage:
        def __init__(self, path):
            pass

        def get_regions(self):
            d = {"TextRegion": regions}
            if seps:
                d["SeparatorRegion"] = seps
            return d
    monkeypatch.setattr(fg, "Page", FakePage)
    for key, want in rec["masked"].items():
        mh, ms = (int(t.split("=")[1]) for t in key.split(","))
        got = fg.mask_horizontally_separated_confs(confs.copy(), "unused", mask_heading=bool(mh), mask_horizontal=bool(ms))
        if "raises" in want:
            n = len(regions)
            exp = confs.astype(np.float64).copy()
            for pr in rec["pairs"]:
                if pr["aligned_heading_separated"]:
                    exp[pr["i"], pr["j"]] = exp[pr["j"], pr["i"]] = 0.0
            pairs_seen = {(pr["i"], pr["j"]) for pr in rec["pairs"]} | {(pr["j"], pr["i"]) for pr in rec["pairs"]}
            assert all((i, j) in pairs_seen for i in range(n) for j in range(n) if i != j)     # (small pages record every pair)
            assert np.array_equal(np.asarray(got, np.float64), exp)
            continue
        assert str(got.dtype) == want["dtype"], key
        assert np.array_equal(np.asarray(got, np.float64), np.asarray(want["values"], np.float64)), key


def test_bounding_box_rules_on_raw_boxes():
    from citlab_article_separation_new_amd import feature_generation as fg
    assert len(G["bbox_rules"]) == 400
    for c in G["bbox_rules"]:
        assert bool(fg.is_vertically_separated(*c["a"], *c["b"], *c["s"])) == c["vertical"], c
        assert bool(fg.is_horizontally_separated(*c["a"], *c["b"], *c["s"])) == c["horizontal"], c
    assert any(c["vertical"] for c in G["bbox_rules"]) and any(c["horizontal"] for c in G["bbox_rules"])


def test_edge_sets():
    from citlab_article_separation_new_amd import feature_generation as fg
    for c in G["edges"]:
        if c["kind"] == "full":
            assert fg.fully_connected_edges(c["n"]).tolist() == c["edges"]
        else:
            assert np.asarray(fg.delaunay_edges(c["n"], np.asarray(c["positions"], np.float64))).tolist() == c["edges"], c["n"]


def test_json_to_feed_arrays_and_full_relations(tmp_path):
    """input_dataset.py:343-375,444-457: every array with its dtype and shape; all N^2 ordered pairs row major + their ground-truth vector"""
    from citlab_article_separation_new_amd import gnn_input
    for k, c in enumerate(G["json_feed"]):
        p = tmp_path / f"g{k}.json"
        p.write_text(json.dumps(c["json"]))
        got = gnn_input.get_input_and_target_from_json(str(p))
        assert set(got) == set(c["arrays"])
        for name, want in c["arrays"].items():
            assert str(got[name].dtype) == want["dtype"] and list(got[name].shape) == want["shape"], name
            assert got[name].tolist() == want["values"], name
        rel, nrel, relgt = gnn_input.build_full_relations(c["json"]["num_nodes"], got["gt_relations"])
        assert rel.tolist() == c["relations"] and int(nrel) == c["num_relations"] and relgt.tolist() == c["relations_gt"]
        assert rel.dtype == np.int32 and relgt.dtype == np.int32


def test_rescale_polygons():
    """region_net_post_processor_base.py:253-268 (int() truncation of every coordinate times the factor)"""
    from citlab_article_separation_new_amd.separator_net_post_processor import SeparatorNetPostProcessor
    for c in G["rescale"]:
        got = SeparatorNetPostProcessor.rescale_polygons(None, json.loads(json.dumps(c["polygons"])), c["scaling_factor"])
        assert json.loads(json.dumps(got)) == c["rescaled"], c["scaling_factor"]


def test_heading_scale_interval_and_net_confidence_of_a_line():
    """heading_net_post_processor.py:50-63,247-270: the mean net confidence over the rescaled bounding box of a line (the slice clips at the
    map's border, the divisor is the box's nominal size)"""
    from citlab_article_separation_new_amd.heading_net_post_processor import HeadingNetPostProcessor, LineGeometry, _scale_to_new_interval
    for c in G["scale_interval"]:
        assert float(_scale_to_new_interval(c["data"], c["old_min"], c["old_max"])) == c["value"]
    hp = object.__new__(HeadingNetPostProcessor)
    net = np.asarray(G["net_prob_map"], np.float64)
    for c in G["net_prob"]:
        line = LineGeometry("x", [tuple(p) for p in c["surr_p"]] if c["surr_p"] is not None else [])
        got = hp.get_net_prob_for_text_line(net, line, c["scaling_factor"])
        if c["value"] is None:
            assert not np.isfinite(got)
        else:
            assert float(got) == c["value"], c


@pytest.mark.parametrize("k", range(len(G["heading"])))
def test_heading_fusion_rule(k):
    """heading_net_post_processor.py:65-200 of the reference, driven with recorded per-line measurements and a recording PAGE writer: which lines
    get the heading tag and which type every region ends with -- for three weight sets, two threshold sets, regions without lines and lines
    without outline"""
    from citlab_article_separation_new_amd import heading_net_post_processor as hnp
    c = G["heading"][k]

    class Line:
        def __init__(self, lid):
            self.id, self.sem = lid, None

        def set_structure_attribute(self, key, value):
            assert key == "semantic_type"
            self.sem = value

        def flush(self):
            pass

        def get_semantic_type(self):
            return self.sem

    class Region:
        def __init__(self, rid, lines):
            self.id, self.text_lines, self.region_type = rid, lines, None
            self.node = types.SimpleNamespace(set=lambda k, v: None)
    regions = [Region(f"r{r}", [Line(lid) for lid in ids]) for r, ids in enumerate(c["regions"])]
    lines = [ln for r in regions for ln in r.text_lines]
    page_object = types.SimpleNamespace(get_text_regions=lambda: regions)
    writer = types.SimpleNamespace(page_object=page_object, save_page_xml=lambda path: None)
    m = c["measurements"]
    # (a line without outline measures 0 / 0, :96-99; its net confidence is whatever get_net_prob_for_text_line returns -- recorded)
    sw = {ln.id: (m[ln.id][0] if c["has_outline"][ln.id] else 0) for ln in lines}
    th = {ln.id: (m[ln.id][1] if c["has_outline"][ln.id] else 0) for ln in lines}
    net = {ln.id: (m[ln.id][2] if c["weight_dict"]["net"] != 0 else 0) for ln in lines}
    hnp.apply_heading_values(writer, lines, (sw, th, net), c["weight_dict"], c["threshold"], c["thresh_dict"], c["text_line_percentage"], "out.xml")
    assert sorted(ln.id for ln in lines if ln.sem == "heading") == c["heading_lines"]
    assert {r.id: r.region_type for r in regions} == c["region_types"]
