"""Host-side pieces of the two ARU-Net pipelines that need no GPU: image decode conventions, region writers,
sub-list fan-out of the CLI, and the heading fusion rule driven with hand-made feature images."""
import os

import numpy as np
import pytest
from PIL import Image

PAGE = """<?xml version="1.0" encoding="UTF-8"?>
<PcGts xmlns="http://schema.primaresearch.org/PAGE/gts/pagecontent/2013-07-15">
  <Metadata><Creator>t</Creator><Created>2020-01-01T00:00:00</Created><LastChange>2020-01-01T00:00:00</LastChange></Metadata>
  <Page imageFilename="img.png" imageWidth="400" imageHeight="300">
    <TextRegion id="r1"><Coords points="10,10 200,10 200,140 10,140"/>
      <TextLine id="h1"><Coords points="10,10 200,10 200,60 10,60"/></TextLine>
      <TextLine id="b1"><Coords points="10,80 200,80 200,100 10,100"/></TextLine>
    </TextRegion>
    <TextRegion id="r2" type="heading"><Coords points="10,150 200,150 200,290 10,290"/>
      <TextLine id="b2"><Coords points="10,150 200,150 200,170 10,170"/></TextLine>
      <TextLine id="b3"><Coords points="10,180 200,180 200,200 10,200"/></TextLine>
      <TextLine id="b4"><Coords points="10,210 200,210 200,230 10,230"/></TextLine>
    </TextRegion>
    <TextRegion id="r3"><Coords points="210,10 390,10 390,60 210,60"/>
      <TextLine id="nocoords"/>
    </TextRegion>
    <SeparatorRegion id="SeparatorRegion_1"><Coords points="205,0 207,0 207,300 205,300"/></SeparatorRegion>
  </Page>
</PcGts>
"""


def _workdir(tmp_path, with_page=True):
    img = tmp_path / "img.png"
    Image.fromarray(np.full((300, 400), 230, np.uint8)).save(img)
    (tmp_path / "page").mkdir()
    if with_page:
        (tmp_path / "page" / "img.xml").write_text(PAGE)
    return str(img)


def test_image_decode_conventions(tmp_path):
    from citlab_article_separation_new_amd import image_io
    rgb = np.zeros((4, 5, 3), np.uint8)
    rgb[..., 0], rgb[..., 1], rgb[..., 2] = 200, 100, 50            # R, G, B
    Image.fromarray(rgb).save(tmp_path / "c.png")
    Image.fromarray(rgb[..., 1]).save(tmp_path / "g.png")
    bgr = image_io.load_image_bgr(str(tmp_path / "c.png"))
    assert bgr.shape == (4, 5, 3) and bgr[0, 0].tolist() == [50, 100, 200]
    assert image_io.load_image_bgr(str(tmp_path / "g.png")).shape == (4, 5)
    assert image_io.get_image_dimensions(str(tmp_path / "c.png")) == (5, 4)
    g = image_io.load_image_gray(str(tmp_path / "c.png"))
    assert g[0, 0] == (50 * 3735 + 100 * 19235 + 200 * 9798 + 16384) >> 15


def test_image_decode_deep_palette_alpha_and_exif(tmp_path):
    """cv2.imread conventions Pillow does not follow by itself: 16-bit samples keep their high byte (not clipped to
    white), palette / alpha files become 3-channel BGR, the EXIF orientation is applied."""
    from citlab_article_separation_new_amd import image_io
    deep = (np.arange(20, dtype=np.uint16).reshape(4, 5) * 3000 + 17)
    Image.fromarray(deep).save(tmp_path / "d16.png")
    Image.fromarray(deep).save(tmp_path / "d16.tif")
    for name in ("d16.png", "d16.tif"):
        got = image_io.load_image_bgr(str(tmp_path / name))
        assert got.dtype == np.uint8 and np.array_equal(got, (deep >> 8).astype(np.uint8)), name
    assert image_io.load_image_bgr(str(tmp_path / "d16.png")).max() < 255          # nothing saturates
    pal = Image.fromarray(np.array([[0, 1], [1, 0]], np.uint8), mode="P")
    pal.putpalette([10, 20, 30, 200, 150, 100] + [0] * 756)
    pal.save(tmp_path / "p.png")
    got = image_io.load_image_bgr(str(tmp_path / "p.png"))
    assert got.shape == (2, 2, 3) and got[0, 0].tolist() == [30, 20, 10] and got[0, 1].tolist() == [100, 150, 200]
    rgba = np.zeros((2, 3, 4), np.uint8)
    rgba[..., 0], rgba[..., 3] = 90, 7
    Image.fromarray(rgba).save(tmp_path / "a.png")
    assert image_io.load_image_bgr(str(tmp_path / "a.png"))[0, 0].tolist() == [0, 0, 90]
    Image.fromarray(np.array([[0, 255]], np.uint8)).convert("1").save(tmp_path / "b.png")
    assert image_io.load_image_bgr(str(tmp_path / "b.png")).tolist() == [[0, 255]]
    # EXIF orientation 6 = "rotate 90 clockwise to display": a 2x3 file decodes to 3x2
    img = Image.fromarray(np.arange(6, dtype=np.uint8).reshape(2, 3) * 40)
    exif = Image.Exif()
    exif[0x0112] = 6
    img.save(tmp_path / "r.png", exif=exif)
    got = image_io.load_image_bgr(str(tmp_path / "r.png"))
    assert got.shape == (3, 2) and np.array_equal(got, np.rot90(np.arange(6, dtype=np.uint8).reshape(2, 3) * 40, -1))


def test_worker_that_dies_without_result_fails_the_run_instead_of_hanging():
    import multiprocessing as mp
    from citlab_article_separation_new_amd import run_gnn_clustering as cli
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    procs = [ctx.Process(target=_post_ok, args=(q,)), ctx.Process(target=os._exit, args=(3,))]
    for pr in procs:
        pr.start()
    out, errors = cli._collect_results(procs, q)
    assert out == ["fine"] and len(errors) == 1 and "exit code 3" in errors[0] and "worker 1" in errors[0]
    # a worker that POSTED its result and then exits non-zero (a crash at teardown) is not a failure, and it must
    # not get the healthy worker that is still running terminated
    q = ctx.Queue()
    procs = [ctx.Process(target=_post_ok_then_crash, args=(q,)), ctx.Process(target=_post_ok_late, args=(q,))]
    for pr in procs:
        pr.start()
    out, errors = cli._collect_results(procs, q)
    assert sorted(out) == ["early", "late"] and errors == []


def _post_ok(q):
    q.put((0, "ok", ["fine"]))


def _post_ok_then_crash(q):
    q.put((0, "ok", ["early"]))
    q.close()
    q.join_thread()
    os._exit(7)


def _post_ok_late(q):
    import time
    time.sleep(2.5)
    q.put((1, "ok", ["late"]))


def test_cli_sub_lists_match_reference_arithmetic():
    from citlab_article_separation_new_amd.run_net_post_processing import build_parser, build_sub_lists
    imgs = [f"i{k}" for k in range(23)]
    subs = build_sub_lists(imgs, 8)                                  # 23 // 8 = 2 per sub-list
    assert [len(s) for s in subs] == [2] * 11 + [1] and sum(subs, []) == imgs
    assert [len(s) for s in build_sub_lists(imgs[:3], 8)] == [1, 1, 1]
    assert max(len(s) for s in build_sub_lists([str(i) for i in range(1000)], 4)) == 50
    a = build_parser().parse_args(["--path_to_image_list", "l", "--path_to_pb", "m.pb", "--mode", "separator"])
    assert a.fixed_height is None and a.scaling_factor == 1.0 and a.threshold == 0.05 and a.num_processes == 8
    with pytest.raises(SystemExit):
        build_parser().parse_args(["--path_to_image_list", "l", "--path_to_pb", "m", "--mode", "textblock"])


def test_separator_writer_replaces_regions_and_names_files(tmp_path):
    from citlab_article_separation_new_amd.page_xml import Page
    from citlab_article_separation_new_amd.region_to_page_writer import SeparatorRegionToPageWriter
    img = _workdir(tmp_path)
    page_path = str(tmp_path / "page" / "img.xml")
    polys = {"SeparatorRegion_horizontal": [[[(10, 70), (200, 70), (200, 73), (10, 73), (10, 70)]]],
             "SeparatorRegion_vertical": [[[(205, 0), (208, 0), (208, 300), (205, 300), (205, 0)],
                                           [(206, 10), (206, 20), (207, 20), (207, 10), (206, 10)]]]}
    w = SeparatorRegionToPageWriter(page_path, img, 150, 1.0, polys)
    assert w.scaling_factor == 0.5
    w.remove_separator_regions_from_page()
    w.merge_regions()
    w.save_page_xml(page_path + ".xml")
    out = Page(page_path + ".xml")
    seps = out.get_regions()["SeparatorRegion"]
    assert [s.id for s in seps] == ["SeparatorRegion_1", "SeparatorRegion_2"]
    assert [s.get_orientation() for s in seps] == ["horizontal", "vertical"]
    assert seps[0].points == [(10, 70), (200, 70), (200, 73), (10, 73), (10, 70)]
    assert seps[1].points[0] == (205, 0) and len(seps[1].points) == 5    # exterior ring only
    assert len(out.get_textlines()) == 6                              # text content untouched


def test_writer_creates_page_with_scaled_size_when_missing(tmp_path):
    from citlab_article_separation_new_amd.region_to_page_writer import RegionToPageWriter
    img = _workdir(tmp_path, with_page=False)
    w = RegionToPageWriter(str(tmp_path / "page" / "img.xml"), img, fixed_height=150, scaling_factor=1.0)
    assert w.page_object.get_image_resolution() == (200, 150)         # region_to_page_writer.py:35-37 quirk


def _heading_processor(tmp_path, **kw):
    from citlab_article_separation_new_amd.config import AruConfig
    from citlab_article_separation_new_amd.heading_net_post_processor import HeadingNetPostProcessor
    from citlab_article_separation_new_amd.weights import init_aru_weights, save_weights
    cfg = AruConfig()
    path = str(tmp_path / "aru.asepw")
    save_weights(path, init_aru_weights(cfg, 1), {"aru_cfg": cfg.__dict__ if hasattr(cfg, "__dict__") else {}})
    return HeadingNetPostProcessor([], path, 150, 1.0, **kw)


def _glyphs(swt, y0, y1, x0, n, gw, value):
    for k in range(n):
        swt[y0:y1, x0 + k * (gw + 4): x0 + k * (gw + 4) + gw] = value


def test_heading_fusion_rule(tmp_path):
    from citlab_article_separation_new_amd.page_xml import Page
    img = _workdir(tmp_path)
    proc = _heading_processor(tmp_path, weight_dict={'net': 0.8, 'stroke_width': 0.0, 'text_height': 0.2},
                              threshold=0.4,
                              thresh_dict={'net_thresh': 1.0, 'stroke_width_thresh': 1.0, 'text_height_thresh': 0.9,
                                           'sw_th_thresh': 0.9}, text_line_percentage=0.8)
    swt = np.zeros((300, 400), np.uint8)
    _glyphs(swt, 15, 55, 14, 6, 20, 6)                               # h1: tall glyphs (40 px), stroke value 6
    for y in (82, 152, 182, 212):
        _glyphs(swt, y, y + 14, 14, 10, 8, 2)                        # body lines: 14 px high, stroke value 2
    prob = np.zeros((150, 200))                                      # heading net output at the scaled size (sc = 0.5)
    prob[40:50, 5:100] = 0.9                                         # b1 (y 80..100 -> 40..50): strong net evidence
    page = proc.to_page_xml(str(tmp_path / "page" / "img.xml"), img, prob, swt)
    out = Page(str(tmp_path / "page" / "img.xml.xml"))
    sem = {tl.id: tl.get_semantic_type() for tl in out.get_textlines()}
    # h1: text height conf = 1.0 >= 0.9 -> heading; b1: 0.8 * (0.9*95*10 / (96*11)) ~ 0.65 > 0.4 -> heading
    assert sem["h1"] == "heading" and sem["b1"] == "heading"
    assert sem["b2"] is None and sem["b3"] is None and sem["b4"] is None and sem["nocoords"] is None
    types = {r.id: r.region_type for r in out.get_text_regions()}
    assert types == {"r1": "heading", "r2": "paragraph", "r3": "paragraph"}   # r2 loses its stale heading type
    # the features behind the decision
    tl = {t.id: t for t in page.get_textlines()}
    sw, th = proc.get_swt_features_textline(swt, tl["h1"])
    assert (sw, th) == (6.0, 40)
    assert proc.get_swt_features_textline(swt, tl["b1"]) == (2.0, 14)
    p = proc.get_net_prob_for_text_line(prob, tl["b1"], 0.5)
    assert p == np.sum(prob[40:51, 5:101]) / (96 * 11)


def test_heading_without_net_weight_uses_only_swt(tmp_path):
    from citlab_article_separation_new_amd.page_xml import Page
    img = _workdir(tmp_path)
    proc = _heading_processor(tmp_path, weight_dict={'net': 0.0, 'stroke_width': 0.5, 'text_height': 0.5},
                              threshold=0.5)
    swt = np.zeros((300, 400), np.uint8)
    proc.to_page_xml(str(tmp_path / "page" / "img.xml"), img, None, swt)
    out = Page(str(tmp_path / "page" / "img.xml.xml"))
    # all features equal: differences are 0, scale_to_new_interval returns the raw 0 -> nothing is a heading
    assert all(tl.get_semantic_type() is None for tl in out.get_textlines())
    assert {r.region_type for r in out.get_text_regions()} == {"paragraph"}


def test_swt_component_cleaning():
    from citlab_article_separation_new_amd.heading_net_post_processor import StrokeWidthDistanceTransform
    s = StrokeWidthDistanceTransform()
    img = np.zeros((30, 80), np.uint8)
    img[5:15, 10:16] = 2
    img[6:8, 50:52] = 9
    img[20:23, 45:75] = 1
    boxes = s.connected_components_cv(img)
    assert sorted(boxes) == [(10, 5, 6, 10), (45, 20, 30, 3), (50, 6, 2, 2)]
    assert s.clean_connected_components(boxes) == [(10, 5, 6, 10)]


def test_line_box_arithmetic_equals_the_per_point_form():
    """The heading owner measures from the four extremes of a text line (heading_net_post_processor.line_boxes, array arithmetic)
    instead of from its points (heading_net_post_processor.py:218-270 of the reference): the box of the rescaled outline is the
    rescaled box (int() truncation is monotone, also for negative coordinates), and the clipped crop is what numpy makes of
    net_output[ya:ya+height, xa:xa+width] -- negative starts counting from the end included."""
    from citlab_article_separation_new_amd.heading_net_post_processor import LineGeometry, _slice_bounds, line_boxes
    from citlab_article_separation_new_amd.host_util import rescale_points
    rng = np.random.default_rng(4)
    lines = []
    for k in range(300):
        n = int(rng.integers(1, 9))
        lines.append(LineGeometry(f"l{k}", [(int(x), int(y)) for x, y in rng.integers(-40, 700, (n, 2))]))
    lines.insert(7, LineGeometry("empty", []))
    ids, boxes, has = line_boxes(lines)
    assert ids == [l.id for l in lines] and not has[7] and has.sum() == 300 and boxes.dtype == np.int64
    for sc in (0.2, 1 / 3, 0.75, 1.0, 1.37):
        lo = (boxes[:, :2] * sc).astype(np.int64)
        hi = (boxes[:, 2:] * sc).astype(np.int64)
        size = hi - lo + 1
        for h, w in ((120, 90), (500, 400), (1000, 1000)):
            x0, x1 = _slice_bounds(lo[:, 0], size[:, 0], w)
            y0, y1 = _slice_bounds(lo[:, 1], size[:, 1], h)
            for i, l in enumerate(lines):
                if not l.surr_p:
                    continue
                x, y, bw, bh = l.get_bounding_box()
                assert (x, y, x + bw + 1, y + bh + 1) == (boxes[i, 0], boxes[i, 1], boxes[i, 2] + 2, boxes[i, 3] + 2)
                pts = rescale_points(l.surr_p, sc)
                xs, ys = [p[0] for p in pts], [p[1] for p in pts]
                xa, ya, width, height = min(xs), min(ys), max(xs) - min(xs) + 1, max(ys) - min(ys) + 1
                assert (xa, ya, width, height) == (lo[i, 0], lo[i, 1], size[i, 0], size[i, 1])
                ys0, ys1, _ = slice(ya, ya + height).indices(h)
                xs0, xs1, _ = slice(xa, xa + width).indices(w)
                assert (xs0, max(xs0, xs1), ys0, max(ys0, ys1)) == (x0[i], x1[i], y0[i], y1[i])


def test_separator_regions_written_as_records_give_the_bytes_of_the_element_tree_writer(tmp_path):
    """round 6: add_separator_region keeps the regions as (id, custom, points) records and write_page_xml splices them
    into the serialised text -- byte for byte what ElementTree writes for real nodes, on an empty page, behind text regions, with and without
    orientation, after ids that are already taken; a reader of the DOM in between finds real nodes"""
    from citlab_article_separation_new_amd import page_xml
    rng = np.random.default_rng(3)

    def build(fast, with_text, taken):
        if with_text:
            src = tmp_path / "in.xml"
            src.write_text('<?xml version="1.0" encoding="UTF-8"?>\n<PcGts xmlns="http://schema.primaresearch.org/PAGE/gts/pagecontent/2013-07-15">'
                           '<Metadata><Creator>x</Creator><Created>2020-01-01T00:00:00</Created><LastChange>2020-01-01T00:00:00</LastChange></Metadata>'
                           '<Page imageFilename="a.png" imageWidth="300" imageHeight="400"><TextRegion id="r1"><Coords points="1,2 30,2 30,40 1,40"/>'
                           '<TextLine id="l1"><Coords points="1,2 30,2 30,10 1,10"/><Baseline points="1,9 30,9"/><TextEquiv><Unicode>a &amp; b</Unicode></TextEquiv>'
                           '</TextLine></TextRegion>' + ('<SeparatorRegion id="SeparatorRegion_2"><Coords points="5,5 6,6 7,7"/></SeparatorRegion>' if taken else '')
                           + '</Page></PcGts>')
            page = page_xml.Page(str(src))
        else:
            page = page_xml.Page(creator_name="t", img_filename="a.png", img_w=300, img_h=400)
        r = np.random.default_rng(5)
        ids = []
        for k in range(40):
            pts = [(int(a), int(b)) for a, b in r.integers(0, 300, (int(r.integers(3, 7)), 2))]
            ids.append(page.add_separator_region(pts, [None, "horizontal", "vertical"][k % 3]))
        if not fast:
            page._materialize()                              # real ElementTree nodes: the writer's plain path
            assert not page.__dict__.get("_sep_fast")
        out = tmp_path / ("fast.xml" if fast else "slow.xml")
        page.write_page_xml(str(out))
        data = re.sub(rb"<LastChange>[^<]*</LastChange>", b"", out.read_bytes())
        data = re.sub(rb"<Created>[^<]*</Created>", b"", data)
        return data, ids, page
    import re
    for with_text in (False, True):
        for taken in ((False, True) if with_text else (False,)):
            fast, ids_f, page = build(True, with_text, taken)
            slow, ids_s, _ = build(False, with_text, taken)
            assert fast == slow and ids_f == ids_s and fast.count(b"<SeparatorRegion ") == 40 + (1 if taken else 0)
            assert ("SeparatorRegion_2" not in ids_f) == taken and ids_f[0] == "SeparatorRegion_1"
            # a second write gives the same file; reading the DOM afterwards finds the regions as nodes, and a third write still agrees
            again = tmp_path / "again.xml"
            page.write_page_xml(str(again))
            strip = lambda b: re.sub(rb"<Created>[^<]*</Created>", b"", re.sub(rb"<LastChange>[^<]*</LastChange>", b"", b))
            assert strip(again.read_bytes()) == fast
            assert len(page.get_regions()["SeparatorRegion"]) == 40 + (1 if taken else 0)
            page.write_page_xml(str(again))
            assert strip(again.read_bytes()) == fast
