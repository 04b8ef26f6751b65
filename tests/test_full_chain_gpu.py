"""The whole reference workflow on the GPU, CLI to CLI: scan + PAGE-XML -> separators -> headings -> graph json
(feature generation with the GPU distance transform) -> GNN + clustering with confidence masking -> PAGE-XML with
article ids.  Each stage is also checked against the oracle on the same inputs (ids identical)."""
import json
import os
import shutil

import numpy as np
import pytest
from PIL import Image

pytestmark = pytest.mark.gpu

MASK = [1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1]


def _write_page(path, W, H, blocks):
    regs = []
    for i, (x0, y0, x1, y1, nl) in enumerate(blocks):
        lines = []
        lh = (y1 - y0) // nl
        for k in range(nl):
            ya, yb = y0 + k * lh, y0 + (k + 1) * lh - 4
            lines.append(f'<TextLine id="r{i}l{k}"><Coords points="{x0},{ya} {x1},{ya} {x1},{yb} {x0},{yb}"/>'
                         f'<Baseline points="{x0},{yb - 3} {x1},{yb - 3}"/><TextEquiv><Unicode>t{k}</Unicode>'
                         f'</TextEquiv></TextLine>')
        regs.append(f'<TextRegion id="r{i}"><Coords points="{x0},{y0} {x1},{y0} {x1},{y1} {x0},{y1}"/>'
                    + "".join(lines) + '</TextRegion>')
    path.write_text('<?xml version="1.0" encoding="UTF-8"?>\n<PcGts xmlns="http://schema.primaresearch.org/PAGE/gts/'
                    'pagecontent/2013-07-15"><Metadata><Creator>t</Creator><Created>2020-01-01T00:00:00</Created>'
                    '<LastChange>2020-01-01T00:00:00</LastChange></Metadata>'
                    f'<Page imageFilename="p0.png" imageWidth="{W}" imageHeight="{H}">' + "".join(regs)
                    + '</Page></PcGts>')


def test_full_chain(tmp_path):
    from citlab_article_separation_new_amd import (feature_generation as fg, pb_import, run_feature_generation,
                                                   run_gnn_clustering, run_net_post_processing, synth)
    from citlab_article_separation_new_amd.clustering import TextblockClustering
    from citlab_article_separation_new_amd.config import AruConfig, GnnConfig
    from citlab_article_separation_new_amd.page_xml import Page
    from citlab_article_separation_new_amd.weights import init_aru_weights, init_gnn_weights
    from oracle import classical_oracle as co, gnn_oracle
    W, H = 600, 900
    data = tmp_path / "data"
    (data / "page").mkdir(parents=True)
    gray = synth.synth_page(9, W=W, H=H)
    Image.fromarray(gray).save(data / "p0.png")
    blocks = [(40 + 190 * c, 60 + 140 * r, 200 + 190 * c, 180 + 140 * r, 3) for r in range(5) for c in range(3)]
    _write_page(data / "page" / "p0.xml", W, H, blocks)
    lst = tmp_path / "images.lst"
    lst.write_text(str(data / "p0.png") + "\n")
    # models: two ARU-Nets (separator / heading) and the GNN as frozen graphs
    acfg = AruConfig()
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    import tf_aru_graph
    for name, seed in (("sep.pb", 21), ("head.pb", 22)):          # frozen graphs in TF1 layout, serialised by protobuf
        (tmp_path / name).write_bytes(tf_aru_graph.build_aru_pb(
            init_aru_weights(acfg, seed, bias_jitter=0.05, logit_scale=0.05), acfg))
    gcfg = GnnConfig()

    # 1. separators (threshold mid-range so random weights give structure), output page/p0.xml.xml
    assert run_net_post_processing.main(["--path_to_image_list", str(lst), "--path_to_pb", str(tmp_path / "sep.pb"),
                                         "--mode", "separator", "--fixed_height", "450", "--threshold", "0.5",
                                         "--num_processes", "1"]) == 0
    shutil.move(str(data / "page" / "p0.xml.xml"), str(data / "page" / "p0.xml"))
    # 2. headings on top of that
    assert run_net_post_processing.main(["--path_to_image_list", str(lst), "--path_to_pb", str(tmp_path / "head.pb"),
                                         "--mode", "heading", "--fixed_height", "300", "--num_processes", "1"]) == 0
    shutil.move(str(data / "page" / "p0.xml.xml"), str(data / "page" / "p0.xml"))
    page = Page(str(data / "page" / "p0.xml"))
    assert len(page.get_textlines()) == 45
    assert {r.region_type for r in page.get_text_regions()} <= {"heading", "paragraph"}
    # 3. graph json
    plist = tmp_path / "pages.lst"
    plist.write_text(str(data / "page" / "p0.xml") + "\n")
    assert run_feature_generation.main(["--pagexml_list", str(plist), "--separators", "bb"]) == 0
    jpath = data / "json15d2bb" / "p0.json"
    feat = json.loads(jpath.read_text())
    assert feat["num_nodes"] == 15 and len(feat["node_features"][0]) == 15
    # the same json from the oracle-side distance transform
    ref = fg.build_input_and_target(str(data / "page" / "p0.xml"), separators="bb",
                                    swt_img=co.swt_distance_transform(gray))
    assert np.array_equal(np.array(feat["node_features"], np.float32), ref[3])
    assert np.array_equal(np.array(feat["interacting_nodes"]), ref[1])
    assert np.array_equal(np.array(feat["edge_features"], np.float32), ref[4])
    # the relation net for THIS page graph: seeded weights whose pair classifier is calibrated (oracle/gnn_cases.py) so that
    # blocks of one column mostly get confidences above 0.5 and the rest below -- the ids compared below then come from
    # a clustering that is neither "one article" nor "all singletons"
    from oracle import gnn_cases
    keep = [i for i, m in enumerate(MASK) if m]
    column = np.array([i % 3 for i in range(15)])
    gw = gnn_cases.calibrate_l1_classifier(init_gnn_weights(gcfg, 23, bias_jitter=0.05), gcfg, 15, ref[1], ref[3][:, keep],
                                           ref[4], column[:, None] == column[None, :], wrong_side=0.2, seed=5)
    (tmp_path / "gnn.pb").write_bytes(pb_import.weights_to_graphdef(gw, "graph/", meta={"num_transition_steps": 3}))
    # 4. GNN + clustering with masking
    jl = tmp_path / "eval.lst"
    jl.write_text(str(jpath) + "\n")
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        outs = run_gnn_clustering.main([
            "--model_dir", str(tmp_path / "gnn.pb"), "--eval_list", str(jl), "--out_dir", "out",
            "--input_params", "node_feature_dim=15", "edge_feature_dim=2",
            "node_input_feature_mask=" + str(MASK).replace(" ", ""), "--clustering_method", "dbscan",
            "--mask_horizontally_separated_confs", "True", "--mask_heading_separated_confs", "True"])
    finally:
        os.chdir(cwd)
    assert len(outs) == 1
    out = outs[0] if os.path.isabs(outs[0]) else os.path.join(tmp_path, outs[0])
    got = [r.text_lines[0].get_article_id() for r in Page(out).get_regions()["TextRegion"]]
    probs = gnn_oracle.forward(15, ref[1], ref[3][:, keep], ref[4], None, gw, gcfg)
    confs = probs[:, 1].reshape(15, 15)
    if "SeparatorRegion" in Page(str(data / "page" / "p0.xml")).get_regions():
        confs = fg.mask_horizontally_separated_confs(confs, str(data / "page" / "p0.xml"))

    class F:
        clustering_params = {}
    tb = TextblockClustering(F())
    tb.set_confs(confs)
    tb.calc("dbscan")
    assert got == [f"a{l}" for l in tb.tb_labels]
    sizes = np.bincount(np.asarray(tb.tb_labels))[1:]
    print("full chain article sizes:", sizes.tolist(), "min|conf-0.5| = %.2e" % np.abs(probs[:, 1] - 0.5).min())
    assert 2 <= len(sizes) < 15 and sizes.max() >= 2, "degenerate clustering"
