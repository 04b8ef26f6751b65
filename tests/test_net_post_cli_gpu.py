"""End to end on the GPU through the reference's CLI surface (run_net_post_processing.py --mode separator|heading):
image file + frozen graph (.pb) -> PAGE-XML.  The fused device path (uint8 upload -> resize/gray -> ARU-Net -> uint8
+ threshold epilogue -> CC filter / openings) must give exactly what the reference's step-by-step sequence gives
when EVERY step is evaluated by the CPU oracle (net: oracle/aru_oracle.py on the weights of the .pb; classical steps:
oracle/classical_oracle.py; heading fusion rule: restated in this file from heading_net_post_processor.py:94-195).
Nothing on the expected side comes from the engine."""
import numpy as np
import pytest
from PIL import Image

pytestmark = pytest.mark.gpu

import os  # noqa: E402
import sys  # noqa: E402
sys.path.insert(0, os.path.dirname(__file__))
import tf_aru_graph  # noqa: E402


def _page_xml(path, W, H, lines):
    regs = []
    for i, (x0, y0, x1, y1) in enumerate(lines):
        regs.append(f'<TextRegion id="r{i}"><Coords points="{x0},{y0} {x1},{y0} {x1},{y1} {x0},{y1}"/>'
                    f'<TextLine id="r{i}l0"><Coords points="{x0},{y0} {x1},{y0} {x1},{y1} {x0},{y1}"/>'
                    f'<Baseline points="{x0},{y1 - 3} {x1},{y1 - 3}"/></TextLine></TextRegion>')
    path.write_text('<?xml version="1.0" encoding="UTF-8"?>\n<PcGts xmlns="http://schema.primaresearch.org/PAGE/gts/'
                    'pagecontent/2013-07-15"><Metadata><Creator>t</Creator><Created>2020-01-01T00:00:00</Created>'
                    '<LastChange>2020-01-01T00:00:00</LastChange></Metadata>'
                    f'<Page imageFilename="x.png" imageWidth="{W}" imageHeight="{H}">' + "".join(regs)
                    + '<SeparatorRegion id="old"><Coords points="1,1 5,1 5,5 1,5"/></SeparatorRegion></Page></PcGts>')


def _setup(tmp_path, W=600, H=900, color=False):
    from citlab_article_separation_new_amd import pb_import, synth
    from citlab_article_separation_new_amd.config import AruConfig
    from citlab_article_separation_new_amd.weights import init_aru_weights
    cfg = AruConfig()
    w = init_aru_weights(cfg, 77, bias_jitter=0.05, logit_scale=0.05)
    pb = tmp_path / "separator_aru.pb"
    pb.write_bytes(tf_aru_graph.build_aru_pb(w, cfg))      # laid out like a TF1 freeze, serialised by protobuf
    data = tmp_path / "data"
    (data / "page").mkdir(parents=True)
    gray = synth.synth_page(3, W=W, H=H)
    if color:
        rng = np.random.default_rng(0)
        rgb = np.stack([gray, np.clip(gray.astype(int) - 10, 0, 255).astype(np.uint8),
                        np.clip(gray.astype(int) + rng.integers(-5, 6, gray.shape), 0, 255).astype(np.uint8)], axis=-1)
        Image.fromarray(rgb).save(data / "p0.png")
    else:
        Image.fromarray(gray).save(data / "p0.png")
    lines = [(60, 70 + 60 * i, 300, 110 + 60 * i) for i in range(6)] + [(320, 80, 560, 170)]
    _page_xml(data / "page" / "p0.xml", W, H, lines)
    lst = tmp_path / "images.lst"
    lst.write_text(str(data / "p0.png") + "\n")
    return str(pb), str(lst), data


@pytest.mark.parametrize("fixed_height,color", [(300, False), (400, True)])
def test_separator_cli_matches_stepwise_reference_sequence(tmp_path, fixed_height, color):
    from citlab_article_separation_new_amd import image_io, net_post_processing_helper as helper, polygonize
    from citlab_article_separation_new_amd import run_net_post_processing as cli
    from citlab_article_separation_new_amd.host_util import rescale_points
    from citlab_article_separation_new_amd.page_xml import Page
    from oracle import classical_oracle as co
    pb, lst, data = _setup(tmp_path, color=color)
    img = image_io.load_image_bgr(str(data / "p0.png"))
    # step-by-step sequence of separator_net_post_processor.py:141-151 (oracle for the classical steps)
    _, grey, sc = co.scale_and_gray(img, fixed_height, 1.0)
    graph = helper.load_graph(pb)
    from oracle import aru_oracle
    prob = aru_oracle.forward_torch(grey.astype(np.float32), graph.tensors, graph.cfg)      # the ORACLE's net output
    thr = round(float(np.median(prob[:, :, 0])), 3)                 # random weights: put the threshold mid-range
    net_u8 = aru_oracle.to_uint8(prob)
    mask = aru_oracle.apply_threshold(net_u8, thr)
    # (the engine's probabilities differ from the oracle's by ~1e-7, so a uint8 value can flip by one where p * 255 sits on an
    # integer: the engine's uint8 map may differ in at most a handful of pixels, none of them at the threshold)
    eng_u8 = np.array(helper.get_net_output(grey, graph, "0") * 255, dtype=np.uint8)
    assert (eng_u8 != net_u8).mean() <= 1e-4 and np.array_equal(helper.apply_threshold(eng_u8, thr), mask)
    post = co.separator_post_process(mask)
    assert 0.02 < (mask[:, :, 0] > 0).mean() < 0.98
    # expected PAGE-XML: the writer fed with the ORACLE-side polygons on a copy of the input page (text lines a
    # vertical separator runs through are cut, polygons with large holes are cut at the holes: test_region_writer_split.py)
    from citlab_article_separation_new_amd.region_to_page_writer import SeparatorRegionToPageWriter
    polygons = {f"SeparatorRegion_{o}": [[rescale_points(r, 1 / sc) for r in poly] for poly in polygonize.shapes(post[o])]
                for o in ("horizontal", "vertical")}
    assert polygons["SeparatorRegion_horizontal"] or polygons["SeparatorRegion_vertical"], "no separators; adjust the threshold"
    ref_xml = tmp_path / "expected_in.xml"
    ref_xml.write_text((data / "page" / "p0.xml").read_text())
    writer = SeparatorRegionToPageWriter(str(ref_xml), str(data / "p0.png"), fixed_height, 1.0, polygons)
    writer.remove_separator_regions_from_page()
    writer.merge_regions()
    writer.save_page_xml(str(tmp_path / "expected_out.xml"))
    want = Page(str(tmp_path / "expected_out.xml"))

    rc = cli.main(["--path_to_image_list", lst, "--path_to_pb", pb, "--mode", "separator",
                   "--fixed_height", str(fixed_height), "--threshold", str(thr), "--num_processes", "1"])
    assert rc == 0
    out = Page(str(data / "page" / "p0.xml.xml"))
    seps = out.get_regions()["SeparatorRegion"]
    assert "old" not in [s.id for s in seps]
    assert [(s.id, s.get_orientation(), s.points) for s in seps] == \
           [(s.id, s.get_orientation(), s.points) for s in want.get_regions()["SeparatorRegion"]]
    assert [(t.id, t.surr_p, t.baseline) for t in out.get_textlines()] == \
           [(t.id, t.surr_p, t.baseline) for t in want.get_textlines()]
    assert len(out.get_textlines()) >= 1


def test_heading_cli_matches_stepwise_reference_sequence(tmp_path):
    from citlab_article_separation_new_amd import image_io, net_post_processing_helper as helper
    from citlab_article_separation_new_amd import run_net_post_processing as cli
    from citlab_article_separation_new_amd.heading_net_post_processor import HeadingNetPostProcessor
    from citlab_article_separation_new_amd.page_xml import Page
    from oracle import classical_oracle as co
    pb, lst, data = _setup(tmp_path)
    img = image_io.load_image_bgr(str(data / "p0.png"))
    _, grey, sc = co.scale_and_gray(img, 450, 1.0)
    graph = helper.load_graph(pb)
    from oracle import aru_oracle
    net_u8 = aru_oracle.to_uint8(aru_oracle.forward_torch(grey.astype(np.float32), graph.tensors, graph.cfg))
    swt = co.swt_distance_transform(img)
    # expected tags: the fusion rule of heading_net_post_processor.py:94-195 restated HERE on oracle-side per-line measurements
    # (CLI constants: weights net 0.8 / stroke width 0.0 / text height 0.2; OR-thresholds 1.0 / 1.0 / 0.9 / 0.9; heading iff
    # confidence > 0.4; region heading iff >= 0.8 of its lines; run_net_post_processing.py:15-23)
    from collections import Counter
    page_in = Page(str(data / "page" / "p0.xml"))
    lines_in = page_in.get_textlines()
    net_map = net_u8[:, :, 0] / 255
    sw, th, netp = {}, {}, {}
    for tl in lines_in:
        sw[tl.id], th[tl.id] = co.swt_features_textline(swt, tl.get_bounding_box())
        xs = [int(sc * x) for x, _ in tl.surr_p]
        ys = [int(sc * y) for _, y in tl.surr_p]
        netp[tl.id] = co.net_prob_textline(net_map, (min(xs), min(ys), max(xs) - min(xs) + 1, max(ys) - min(ys) + 1))
    sw_mode = Counter(sw.values()).most_common(1)[0][0]
    th_mode = Counter(th.values()).most_common(1)[0][0]
    dsw = {k: v - sw_mode for k, v in sw.items()}
    dth = {k: v - th_mode for k, v in th.items()}

    def unit(v, lo, hi):                                             # scale_to_new_interval(.., 0, 1), :50-63
        return v if hi - lo == 0 else (v - lo) / (hi - lo)
    want_lines = {}
    for tl in lines_in:
        c_sw = unit(dsw[tl.id], min(dsw.values()), max(dsw.values()))
        c_th = unit(dth[tl.id], min(dth.values()), max(dth.values()))
        if c_sw >= 1.0 or c_th >= 0.9 or (c_sw + c_th) / 2 >= 0.9 or netp[tl.id] >= 1.0:
            conf = 1.0
        else:
            conf = 0.8 * netp[tl.id] + 0.0 * c_sw + 0.2 * c_th
        want_lines[tl.id] = "heading" if conf > 0.4 else None
    want_regions = {}
    for r in page_in.get_text_regions():
        n_head = sum(1 for tl in r.text_lines if want_lines[tl.id] == "heading")
        want_regions[r.id] = "heading" if r.text_lines and n_head / len(r.text_lines) >= 0.8 else "paragraph"
    assert "heading" in want_lines.values() and None in want_lines.values()        # the case discriminates

    rc = cli.main(["--path_to_image_list", lst, "--path_to_pb", pb, "--mode", "heading", "--fixed_height", "450",
                   "--num_processes", "1"])
    assert rc == 0
    out = Page(str(data / "page" / "p0.xml.xml"))
    assert {t.id: t.get_semantic_type() for t in out.get_textlines()} == want_lines
    assert {r.id: r.region_type for r in out.get_text_regions()} == want_regions
    # the two device-side inputs of the fusion against the oracle's: the distance transform bit for bit, the uint8 net map up to
    # isolated +-1 flips where p * 255 sits on an integer (the probabilities agree to ~1e-7)
    hp = HeadingNetPostProcessor([str(data / "p0.png")], pb, 450, 1.0)
    eng = hp.heading_probability(img)
    assert (eng != net_u8).mean() <= 1e-4 and int(np.abs(eng.astype(int) - net_u8.astype(int)).max()) <= 1
    assert np.array_equal(hp.SWT.distance_transform(img), swt)


@pytest.mark.parametrize("mode", ["separator", "heading"])
def test_host_workers_write_the_same_files_as_the_inline_run(tmp_path, mode):
    """--num_processes = host workers around the GPU owner (host_pipeline.py): images decoded ahead into page-locked
    shared-memory slots, PAGE-XML parsed / written by worker processes.  The files must equal those of the inline run."""
    import re
    import shutil
    from citlab_article_separation_new_amd import run_net_post_processing as cli
    from citlab_article_separation_new_amd import synth
    pb, _, data = _setup(tmp_path)
    names = []
    for k in range(5):                                       # five pages, different content
        name = f"q{k}"
        Image.fromarray(synth.synth_page(10 + k, W=600, H=900)).save(data / f"{name}.png")
        shutil.copy(data / "page" / "p0.xml", data / "page" / f"{name}.xml")
        names.append(name)
    lst = tmp_path / "five.lst"
    lst.write_text("\n".join(str(data / f"{n}.png") for n in names) + "\n")
    # random weights: a threshold in the middle of the net's output range, so that separators exist
    from citlab_article_separation_new_amd import image_io, net_post_processing_helper as helper
    from oracle import classical_oracle as co
    _, grey, _ = co.scale_and_gray(image_io.load_image_bgr(str(data / "q0.png")), 450, 1.0)
    thr = round(float(np.median(helper.get_net_output(grey, helper.load_graph(pb), "0")[:, :, 0])), 3)
    outs = {}
    for workers in (1, 6):
        assert cli.main(["--path_to_image_list", str(lst), "--path_to_pb", pb, "--mode", mode, "--fixed_height", "450",
                         "--threshold", str(thr), "--num_processes", str(workers)]) == 0
        outs[workers] = {}
        for n in names:
            f = data / "page" / f"{n}.xml.xml"
            outs[workers][n] = re.sub(r"<LastChange>[^<]*</LastChange>", "", f.read_text())
            f.unlink()
    assert outs[1] == outs[6]
    assert all(("SeparatorRegion" in v) if mode == "separator" else ("TextLine" in v) for v in outs[6].values())


def test_two_gpu_owner_processes_write_the_same_files_as_one(tmp_path, monkeypatch):
    """run_net_post_processing.py:94-110 with more than one GPU owner: the image list is split over spawned
    owner processes, each with its own model instance and its share of the host workers.  The test box has one GPU, so both
    owners are put on device 0 (ASEP_GPU_OWNERS); files must equal those of the single-owner run."""
    import re
    import shutil
    from citlab_article_separation_new_amd import run_net_post_processing as cli
    from citlab_article_separation_new_amd import synth
    pb, _, data = _setup(tmp_path)
    names = []
    for k in range(6):
        name = f"m{k}"
        Image.fromarray(synth.synth_page(40 + k, W=600, H=900)).save(data / f"{name}.png")
        shutil.copy(data / "page" / "p0.xml", data / "page" / f"{name}.xml")
        names.append(name)
    lst = tmp_path / "six.lst"
    lst.write_text("\n".join(str(data / f"{n}.png") for n in names) + "\n")
    outs = {}
    for owners in ("0", "0,0"):
        monkeypatch.setenv("ASEP_GPU_OWNERS", owners)
        assert cli.main(["--path_to_image_list", str(lst), "--path_to_pb", pb, "--mode", "separator", "--fixed_height", "450",
                         "--threshold", "0.5", "--num_processes", "4"]) == 0
        outs[owners] = {}
        for n in names:
            f = data / "page" / f"{n}.xml.xml"
            outs[owners][n] = re.sub(r"<LastChange>[^<]*</LastChange>", "", f.read_text())
            f.unlink()
    assert outs["0"] == outs["0,0"] and len(outs["0,0"]) == 6
    monkeypatch.setenv("ASEP_GPU_OWNERS", "3")
    from citlab_article_separation_new_amd import _lib
    with pytest.raises(_lib.AsepError, match="ASEP_GPU_OWNERS"):
        cli.main(["--path_to_image_list", str(lst), "--path_to_pb", pb, "--mode", "separator", "--num_processes", "2"])


def test_compute_dtype_switch_for_models_loaded_from_files(tmp_path, monkeypatch):
    """BASELINE configs[4] runs the reference's command lines with "bf16 convs": ASEP_COMPUTE_DTYPE=bf16 makes load_graph hand
    the bf16 engine path the same file; the separator CLI then writes PAGE-XML from it, the probabilities stay within the bf16
    gate (2e-2) of the fp32 ones and are not identical to them (the switch did something)."""
    from citlab_article_separation_new_amd import image_io, net_post_processing_helper as helper, run_net_post_processing as cli
    from oracle import classical_oracle as co
    pb, lst, data = _setup(tmp_path)
    _, grey, _ = co.scale_and_gray(image_io.load_image_bgr(str(data / "p0.png")), 450, 1.0)
    monkeypatch.setenv("ASEP_COMPUTE_DTYPE", "f32")        # the plain fp32 kernels (unset, a model file runs the engine's default: f32s)
    g32 = helper.load_graph(pb)
    assert g32.cfg.compute_dtype == "f32"
    p32 = helper.get_net_output(grey, g32, "0")
    g32.close()
    monkeypatch.delenv("ASEP_COMPUTE_DTYPE")
    gd = helper.load_graph(pb)
    assert gd.cfg.compute_dtype == "f32s"                  # the default arithmetic of a model loaded from a file
    gd.close()
    monkeypatch.setenv("ASEP_COMPUTE_DTYPE", "bf16")
    g16 = helper.load_graph(pb)
    assert g16.cfg.compute_dtype == "bf16"
    p16 = helper.get_net_output(grey, g16, "0")
    g16.close()
    d = float(np.abs(p16 - p32).max())
    assert 0.0 < d <= 2e-2, d
    assert cli.main(["--path_to_image_list", str(lst), "--path_to_pb", pb, "--mode", "separator", "--fixed_height", "450",
                     "--threshold", "0.5", "--num_processes", "1"]) == 0
    assert (data / "page" / "p0.xml.xml").exists()
    monkeypatch.setenv("ASEP_COMPUTE_DTYPE", "fp16")
    with pytest.raises(ValueError, match="ASEP_COMPUTE_DTYPE"):
        helper.load_graph(pb)
    # fp32 with split products: the same file, fp32 results (the fp32 tolerance against the plain fp32 path), the same command line
    monkeypatch.setenv("ASEP_COMPUTE_DTYPE", "f32s")
    gs = helper.load_graph(pb)
    assert gs.cfg.compute_dtype == "f32s"
    ps = helper.get_net_output(grey, gs, "0")
    gs.close()
    ds = float(np.abs(ps - p32).max())
    assert 0.0 < ds <= 1e-5, ds
    (data / "page" / "p0.xml.xml").unlink()
    assert cli.main(["--path_to_image_list", str(lst), "--path_to_pb", pb, "--mode", "separator", "--fixed_height", "450",
                     "--threshold", "0.5", "--num_processes", "1"]) == 0
    assert (data / "page" / "p0.xml.xml").exists()


# what the bf16 command line holds against the fp32 oracle's regions on a mask cut through the MIDDLE of the net's output (no margin)
BF16_REGION_IOU = 0.88           # measured 0.93 (horizontal, 2892 regions) / 0.90 (vertical, 325 regions): 0.7 % of the mask pixels flip where
BF16_REGION_COUNT_TOL = 0.06     # the margin is below the bf16 error, thin regions and the CC / opening stages amplify that; counts 2.6 % / 4.3 % apart


def _raster(polys, H, W):
    """union of simple polygons (lists of (x, y) vertices on the pixel grid) as a boolean image; both sides of a comparison go
    through this same rasteriser, so its edge convention cancels"""
    from PIL import ImageDraw
    img = Image.new("1", (W, H), 0)
    d = ImageDraw.Draw(img)
    for pts in polys:
        if len(pts) >= 3:
            d.polygon([(float(x), float(y)) for x, y in pts], fill=1)
    return np.array(img, dtype=bool)


def test_bf16_separator_cli_polygons_against_the_oracle(tmp_path, monkeypatch):
    """the bf16 command line was checked on probabilities only.  Here its PRODUCT: the separator regions the bf16
    CLI writes for a page, against the regions of the reference's step sequence (separator_net_post_processor.py:141-157)
    evaluated by the fp32 oracle at every step -- same number of regions per orientation, and the rasterised regions overlap
    with IoU >= 0.99.  Weights with unit logit scale (a saturating class softmax, like a trained net's), threshold 0.5."""
    from citlab_article_separation_new_amd import image_io, polygonize, synth
    from citlab_article_separation_new_amd import run_net_post_processing as cli
    from citlab_article_separation_new_amd.config import AruConfig
    from citlab_article_separation_new_amd.host_util import rescale_points
    from citlab_article_separation_new_amd.page_xml import Page
    from citlab_article_separation_new_amd.region_to_page_writer import SeparatorRegionToPageWriter
    from citlab_article_separation_new_amd.weights import init_aru_weights
    from oracle import aru_oracle, classical_oracle as co
    Wp, Hp = 1000, 1500
    cfg = AruConfig()
    w = init_aru_weights(cfg, 4321, bias_jitter=0.05, logit_scale=1.0)
    pb = tmp_path / "separator_aru_unit.pb"
    data = tmp_path / "data"
    (data / "page").mkdir(parents=True)
    Image.fromarray(synth.synth_page(5, W=Wp, H=Hp)).save(data / "q0.png")
    lst = tmp_path / "q.lst"
    lst.write_text(str(data / "q0.png") + "\n")
    _, grey, sc = co.scale_and_gray(image_io.load_image_bgr(str(data / "q0.png")), Hp, 1.0)
    # random unit-scale weights saturate: shift the separator class's bias by the median logit margin of this page, so that the
    # threshold 0.5 cuts through the middle of the net's output (the most sensitive mask; a trained net has far more margin)
    _, inter = aru_oracle.forward_torch(grey.astype(np.float32), w, cfg, return_intermediates=True)
    b = w["aru_net/logit/class/biases"].copy()
    b[0] -= np.float32(np.median(inter["logits"][:, :, 0] - inter["logits"][:, :, 1]))
    w["aru_net/logit/class/biases"] = b
    pb.write_bytes(tf_aru_graph.build_aru_pb(w, cfg))
    prob = aru_oracle.forward_torch(grey.astype(np.float32), w, cfg)
    assert 0.4 < float((prob[:, :, 0] > 0.5).mean()) < 0.6
    post = co.separator_post_process(aru_oracle.apply_threshold(aru_oracle.to_uint8(prob), 0.5))
    polygons = {f"SeparatorRegion_{o}": [[rescale_points(r, 1 / sc) for r in poly] for poly in polygonize.shapes(post[o])]
                for o in ("horizontal", "vertical")}
    writer = SeparatorRegionToPageWriter(str(tmp_path / "none.xml"), str(data / "q0.png"), Hp, 1.0, polygons)
    writer.merge_regions()
    want = [(r.get_orientation(), r.points) for r in writer.page_object.get_regions()["SeparatorRegion"]]
    monkeypatch.setenv("ASEP_COMPUTE_DTYPE", "bf16")
    assert cli.main(["--path_to_image_list", str(lst), "--path_to_pb", str(pb), "--mode", "separator", "--fixed_height", str(Hp),
                     "--threshold", "0.5", "--num_processes", "1"]) == 0
    got = [(r.get_orientation(), r.points) for r in Page(str(data / "page" / "q0.xml.xml")).get_regions()["SeparatorRegion"]]
    report = {}
    for o in ("horizontal", "vertical"):
        a, b = [p for k, p in want if k == o], [p for k, p in got if k == o]
        ra, rb = _raster(a, Hp, Wp), _raster(b, Hp, Wp)
        iou = float((ra & rb).sum()) / max(1, int((ra | rb).sum()))
        report[o] = (len(a), len(b), round(iou, 5), int(ra.sum()))
    print("\nbf16 separator CLI vs oracle regions (count oracle, count bf16, IoU, oracle pixels):", report)
    assert sum(v[0] for v in report.values()) >= 20, "the page must hold a meaningful number of separator regions"
    for o, (na, nb, iou, px) in report.items():
        assert abs(na - nb) <= max(1, round(BF16_REGION_COUNT_TOL * na)) and iou >= BF16_REGION_IOU, (o, na, nb, iou)


def test_pages_in_flight_give_the_masks_of_the_page_by_page_form(tmp_path, monkeypatch):
    """SeparatorNetPostProcessor.enqueue_page / collect_page (the owner runs one page behind the GPU): three pages queued
    before the first is collected give the segments of the synchronous separator_masks; with a segment capacity of 4 every
    mask takes the ask-again branch and still gives them"""
    from citlab_article_separation_new_amd import polygonize, synth
    from citlab_article_separation_new_amd.separator_net_post_processor import SeparatorNetPostProcessor
    from citlab_article_separation_new_amd import image_io, net_post_processing_helper as helper
    from oracle import classical_oracle as co
    pb, lst, data = _setup(tmp_path)
    pages = [synth.synth_page(30 + k, W=500 + 40 * k, H=700 + 30 * k) for k in range(3)]
    _, grey, _ = co.scale_and_gray(image_io.load_image_bgr(str(data / "p0.png")), 450, 1.0)
    thr = round(float(np.median(helper.get_net_output(grey, helper.load_graph(pb), "0")[:, :, 0])), 3)
    proc = SeparatorNetPostProcessor([], pb, 450, 1.0, thr, "0")
    want = []
    for p in pages:
        masks, sc, extras = proc.separator_masks(p, edges_only=False)
        want.append(({k: polygonize.shapes(m, value=255, connectivity=8) for k, m in masks.items()}, sc, extras["size"]))
    assert any(v for w, _, _ in want for v in w.values())
    for cap in (SeparatorNetPostProcessor.SEGMENT_CAPACITY, 4):
        monkeypatch.setattr(SeparatorNetPostProcessor, "SEGMENT_CAPACITY", cap)
        tickets = [proc.enqueue_page(p, edges_only=True, lane=k % 2) for k, p in enumerate(pages)]      # two lanes, as run() does
        for t, (polys, sc, size) in zip(tickets, want):
            masks, sc2, extras = proc.collect_page(t)
            assert sc2 == sc and extras["size"] == size
            for k, (starts, ends) in masks.items():
                assert polygonize.shapes_from_segments(starts, ends, size[0], size[1], connectivity=8) == polys[k]


def test_a_group_of_pages_of_different_sizes_in_one_net_call_gives_the_masks_of_the_page_by_page_form(tmp_path):
    """round 6: SeparatorNetPostProcessor.enqueue_group -- the pipelined owner runs PAGE_GROUP decoded pages, whatever their sizes after
    --fixed_height, through ONE batched net call (asep_aru_forward_batch_dev2) with the classical stages per page behind it: the segments
    of every page are those of the synchronous page-by-page form"""
    from citlab_article_separation_new_amd import polygonize, synth
    from citlab_article_separation_new_amd.separator_net_post_processor import SeparatorNetPostProcessor
    from citlab_article_separation_new_amd import image_io, net_post_processing_helper as helper
    from oracle import classical_oracle as co
    pb, lst, data = _setup(tmp_path)
    pages = [synth.synth_page(40 + k, W=[500, 610, 380, 540][k], H=[700, 660, 720, 700][k]) for k in range(4)]   # widths differ at height 450
    _, grey, _ = co.scale_and_gray(image_io.load_image_bgr(str(data / "p0.png")), 450, 1.0)
    thr = round(float(np.median(helper.get_net_output(grey, helper.load_graph(pb), "0")[:, :, 0])), 3)
    proc = SeparatorNetPostProcessor([], pb, 450, 1.0, thr, "0")
    want = []
    for p in pages:
        masks, sc, extras = proc.separator_masks(p, edges_only=False)
        want.append(({k: polygonize.shapes(m, value=255, connectivity=8) for k, m in masks.items()}, sc, extras["size"]))
    assert len({w[2] for w in want}) >= 3 and any(v for w, _, _ in want for v in w.values())
    tickets = proc.enqueue_group(pages, edges_only=True, lane=1)
    for t, (polys, sc, size) in zip(tickets, want):
        masks, sc2, extras = proc.collect_page(t)
        assert sc2 == sc and extras["size"] == size
        for k, (starts, ends) in masks.items():
            assert polygonize.shapes_from_segments(starts, ends, size[0], size[1], connectivity=8) == polys[k]


@pytest.mark.parametrize("dtype", ["f32s", "f32"])
@pytest.mark.parametrize("color", [False, True])
def test_heading_pages_in_flight_give_the_measurements_of_the_page_by_page_form(tmp_path, color, dtype, monkeypatch):
    """HeadingNetPostProcessor.enqueue_page / collect_page (one page behind the GPU; net output and distance transform stay in
    HBM, gray conversion and box sums on the device) against the step-by-step form on the host -- oracle net output,
    oracle distance transform, numpy slice sums: stroke widths and heights identical, net confidences to 1e-12 (integer sum / 255
    instead of a float64 sum of uint8 / 255).  Lines beyond the page border and a line without outline included."""
    from citlab_article_separation_new_amd import image_io, net_post_processing_helper as helper
    from citlab_article_separation_new_amd.heading_net_post_processor import HeadingNetPostProcessor, LineGeometry
    from oracle import aru_oracle, classical_oracle as co
    monkeypatch.setenv("ASEP_COMPUTE_DTYPE", dtype)        # both fp32 arithmetics of the engine against the same oracle, the same gate
    pb, lst, data = _setup(tmp_path, color=color)
    graph = helper.load_graph(pb)
    assert graph.cfg.compute_dtype == dtype
    base = image_io.load_image_bgr(str(data / "p0.png"))
    pages = [base, np.ascontiguousarray(base[:700, :500]), np.ascontiguousarray(base[::-1])]
    worst_steps, worst_rate, total_steps, total_area = 0.0, 0.0, 0.0, 0
    lines = [LineGeometry(f"l{i}", [(x0, y0), (x1, y0), (x1, y1), (x0, y1)])
             for i, (x0, y0, x1, y1) in enumerate([(60, 70 + 60 * k, 300, 110 + 60 * k) for k in range(6)]
                                                  + [(320, 80, 560, 170), (450, 600, 640, 720), (0, 0, 30, 12), (10, 20, 11, 21)])]
    lines.append(LineGeometry("none", []))
    hp = HeadingNetPostProcessor([], pb, 450, 1.0, weight_dict={"net": 0.8, "stroke_width": 0.0, "text_height": 0.2})
    hp.gpu_devices = "0"
    tickets = [hp.enqueue_page(p, lane=k % 2) for k, p in enumerate(pages)]   # three pages queued (two lanes) before the first is measured
    for img, t in zip(pages, tickets):
        sw, th, netp = hp.collect_page(t, lines)
        _, grey, sc = co.scale_and_gray(img, 450, 1.0)
        net_map = aru_oracle.to_uint8(aru_oracle.forward_torch(grey.astype(np.float32), graph.tensors, graph.cfg))[:, :, 0] / 255
        swt = co.swt_distance_transform(img)
        assert set(sw) == set(th) == set(netp) == {l.id for l in lines}
        assert (sw["none"], th["none"], netp["none"]) == (0, 0, 0)
        n_pos = 0
        for l in lines[:-1]:
            w_sw, w_th = co.swt_features_textline(swt, l.get_bounding_box())
            assert (sw[l.id], th[l.id]) == (w_sw, w_th), l.id
            xs = [int(sc * x) for x, _ in l.surr_p]
            ys = [int(sc * y) for _, y in l.surr_p]
            bw, bh = max(xs) - min(xs) + 1, max(ys) - min(ys) + 1
            want = co.net_prob_textline(net_map, (min(xs), min(ys), bw, bh))
            # The engine's uint8 map differs from the oracle's by isolated +-1 steps where p * 255 sits on an integer: 1.3e-5 (f32) / 1.4e-5
            # (f32s) of the pixels of a whole 3000 x 4500 frame (tests/test_full_frame_gpu.py).  A text-line box has 1e3 .. 2e4 pixels, so the
            # MEAN of a box moves by whole pixel steps: one step in a 10^4-pixel box is already 1e-4 of its pixels.  Rounds 3-4 gated every box
            # at 1e-4, i.e. at ONE step -- a gate both fp32 arithmetics sit on (round 4: one box of the f32s run at two steps = 1.7e-4, the f32
            # run at one): it is sized in steps now -- at most three per box (Poisson tail of a 1.4e-5 rate over 2e4 pixels: 3e-3 per box for
            # two, 3e-4 for three) for the split-product arithmetic, ONE for the plain fp32 kernels (where rounds 3-5 measured at most one: a
            # +-1..2 LSB regression of combine_kernel must not hide behind the f32s allowance) -- and the sum over all boxes of a
            # run at the whole-frame rate times ten (asserted below).
            area = max(1, min(bw, net_map.shape[1]) * min(bh, net_map.shape[0]))
            steps = abs(netp[l.id] - want) * 255 * area
            worst_steps, worst_rate = max(worst_steps, steps), max(worst_rate, steps / area)
            assert steps <= (3.0 if dtype == "f32s" else 1.0) + 1e-6, (l.id, netp[l.id], want, area, steps)
            total_steps, total_area = total_steps + steps, total_area + area
            n_pos += want > 0
        assert n_pos >= 5
    # the run-level gate: all boxes of the three pages together at ten times the whole-frame uint8 mismatch rate (1.4e-5 of the pixels,
    # tests/test_full_frame_gpu.py), and never less than two steps
    assert total_steps <= max(2.0, 10 * 1.4e-5 * total_area) + 1e-6, (total_steps, total_area)
    print(f"\nheading line boxes ({dtype}, color={color}): worst box {worst_steps:.2f} uint8 steps off, worst rate {worst_rate:.2e} of a box's pixels, "
          f"run: {total_steps:.1f} steps over {total_area} box pixels = {total_steps / max(total_area, 1):.2e}")


@pytest.mark.parametrize("mode", ["separator", "heading"])
def test_unreadable_scan_in_a_pipelined_run_raises_instead_of_hanging(tmp_path, mode):
    """a corrupt file in the middle of the list with host workers around the GPU owner (pages in flight on the device, tickets
    pending): the decode worker's error reaches the caller as an IOError naming the file; nothing waits forever"""
    import shutil
    from citlab_article_separation_new_amd import run_net_post_processing as cli
    from citlab_article_separation_new_amd import synth
    pb, _, data = _setup(tmp_path)
    names = []
    for k in range(6):
        name = f"e{k}"
        if k == 3:
            (data / f"{name}.png").write_bytes(b"\x89PNG\r\n\x1a\n" + b"not an image" * 10)
        else:
            Image.fromarray(synth.synth_page(20 + k, W=400, H=600)).save(data / f"{name}.png")
        shutil.copy(data / "page" / "p0.xml", data / "page" / f"{name}.xml")
        names.append(name)
    lst = tmp_path / "six.lst"
    lst.write_text("\n".join(str(data / f"{n}.png") for n in names) + "\n")
    with pytest.raises(IOError, match="e3.png"):
        cli.main(["--path_to_image_list", str(lst), "--path_to_pb", pb, "--mode", mode, "--fixed_height", "300",
                  "--num_processes", "4"])


@pytest.mark.parametrize("color", [False, True])
def test_a_group_of_heading_pages_of_different_sizes_gives_the_measurements_of_the_single_pages(tmp_path, color):
    """round 6: HeadingNetPostProcessor.enqueue_group -- PAGE_GROUP decoded pages of whatever sizes through ONE batched net call
    (asep_aru_forward_batch_dev2), gray conversion + distance transform per page behind it: every line's three measurements equal those of the
    single-page form exactly"""
    from citlab_article_separation_new_amd import image_io
    from citlab_article_separation_new_amd.heading_net_post_processor import HeadingNetPostProcessor, LineGeometry
    pb, lst, data = _setup(tmp_path, color=color)
    base = image_io.load_image_bgr(str(data / "p0.png"))
    pages = [base, np.ascontiguousarray(base[:700, :500]), np.ascontiguousarray(base[::-1]), np.ascontiguousarray(base[100:, 40:])]
    lines = [LineGeometry(f"l{i}", [(x0, y0), (x1, y0), (x1, y1), (x0, y1)])
             for i, (x0, y0, x1, y1) in enumerate([(60, 70 + 60 * k, 300, 110 + 60 * k) for k in range(6)] + [(320, 80, 460, 170), (0, 0, 30, 12)])]
    lines.append(LineGeometry("none", []))
    hp = HeadingNetPostProcessor([], pb, 450, 1.0, weight_dict={"net": 0.8, "stroke_width": 0.0, "text_height": 0.2})
    hp.gpu_devices = "0"
    want = [hp.collect_page(hp.enqueue_page(p), lines) for p in pages]
    got = [hp.collect_page(t, lines) for t in hp.enqueue_group(pages, lane=1)]
    assert got == want and any(v > 0 for v in want[0][2].values())
