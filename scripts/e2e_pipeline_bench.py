"""Wall-clock of the four command-line stages on real files (development aid; numbers quoted in DESIGN.md):
scan PNG + PAGE-XML -> separators -> headings -> graph json -> GNN + clustering -> PAGE-XML with article ids.
Usage: python scripts/e2e_pipeline_bench.py [n_pages] [W H]"""
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image

from citlab_article_separation_new_amd import (pb_import, run_feature_generation, run_gnn_clustering,
                                               run_net_post_processing, synth)
from citlab_article_separation_new_amd.config import AruConfig, GnnConfig
from citlab_article_separation_new_amd.weights import init_aru_weights, init_gnn_weights

n_pages = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3000, 4500)
MASK = [1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1]


def page_xml(path, rng):
    """~200 text regions in 6 columns, 4-12 lines each, like the synthetic scans."""
    regs = []
    colw = (W - 120 - 5 * 40) // 6
    rid = 0
    for c in range(6):
        x0 = 60 + c * (colw + 40)
        y = 60
        while y < H - 300:
            nl = int(rng.integers(4, 13))
            pitch = int(rng.integers(28, 37))
            y1 = y + nl * pitch
            lines = "".join(
                f'<TextLine id="r{rid}l{k}"><Coords points="{x0},{y + k * pitch} {x0 + colw},{y + k * pitch} '
                f'{x0 + colw},{y + (k + 1) * pitch - 4} {x0},{y + (k + 1) * pitch - 4}"/>'
                f'<Baseline points="{x0},{y + (k + 1) * pitch - 8} {x0 + colw},{y + (k + 1) * pitch - 8}"/>'
                f'<TextEquiv><Unicode>t</Unicode></TextEquiv></TextLine>' for k in range(nl))
            regs.append(f'<TextRegion id="r{rid}"><Coords points="{x0},{y} {x0 + colw},{y} {x0 + colw},{y1} {x0},{y1}"/>'
                        + lines + '</TextRegion>')
            rid += 1
            y = y1 + int(rng.integers(20, 60))
    path.write_text('<?xml version="1.0" encoding="UTF-8"?>\n<PcGts xmlns="http://schema.primaresearch.org/PAGE/gts/'
                    'pagecontent/2013-07-15"><Metadata><Creator>t</Creator><Created>2020-01-01T00:00:00</Created>'
                    '<LastChange>2020-01-01T00:00:00</LastChange></Metadata>'
                    f'<Page imageFilename="x.png" imageWidth="{W}" imageHeight="{H}">' + "".join(regs) + '</Page></PcGts>')
    return rid


tmp = tempfile.mkdtemp(prefix="asep_e2e_")
try:
    from pathlib import Path
    root = Path(tmp)
    (root / "data" / "page").mkdir(parents=True)
    rng = np.random.default_rng(0)
    imgs, n_regions = [], 0
    for k in range(n_pages):
        Image.fromarray(synth.synth_page(k, W=W, H=H)).save(root / "data" / f"p{k}.png", compress_level=1)
        n_regions += page_xml(root / "data" / "page" / f"p{k}.xml", rng)
        imgs.append(str(root / "data" / f"p{k}.png"))
    (root / "images.lst").write_text("\n".join(imgs) + "\n")
    acfg = AruConfig()
    extra = [{"name": f"graph/aru_net/attMapG/AvgPool_{i}", "op": "AvgPool"} for i in range(acfg.num_scales_att - 1)]
    extra.append({"name": "output", "op": "Softmax", "input": ["graph/aru_net/logit/logits"]})
    for name, seed in (("sep.pb", 21), ("head.pb", 22)):
        (root / name).write_bytes(pb_import.weights_to_graphdef(init_aru_weights(acfg, seed, logit_scale=0.05), "graph/", extra))
    (root / "gnn.pb").write_bytes(pb_import.weights_to_graphdef(init_gnn_weights(GnnConfig(), 23), "graph/", meta={"num_transition_steps": 3}))
    print(f"{n_pages} pages {W}x{H}, {n_regions} text regions in total")

    def timed(label, fn):
        t0 = time.time()
        fn()
        dt = time.time() - t0
        print(f"{label:58s} {dt:7.2f} s  = {dt / n_pages * 1e3:8.1f} ms/page")

    def promote():
        for k in range(n_pages):
            shutil.move(str(root / "data" / "page" / f"p{k}.xml.xml"), str(root / "data" / "page" / f"p{k}.xml"))

    from citlab_article_separation_new_amd import image_io
    timed("PNG decode only (Pillow)", lambda: [image_io.load_image_bgr(p) for p in imgs])
    # warm-up (library load, first-touch allocations) on page 0
    (root / "one.lst").write_text(imgs[0] + "\n")
    run_net_post_processing.main(["--path_to_image_list", str(root / "one.lst"), "--path_to_pb", str(root / "sep.pb"),
                                  "--mode", "separator", "--threshold", "0.5", "--num_processes", "1"])
    os.remove(root / "data" / "page" / "p0.xml.xml")
    timed("run_net_post_processing --mode separator (defaults)", lambda: run_net_post_processing.main(
        ["--path_to_image_list", str(root / "images.lst"), "--path_to_pb", str(root / "sep.pb"), "--mode", "separator",
         "--threshold", "0.5", "--num_processes", "1"]))
    promote()
    timed("run_net_post_processing --mode heading (defaults)", lambda: run_net_post_processing.main(
        ["--path_to_image_list", str(root / "images.lst"), "--path_to_pb", str(root / "head.pb"), "--mode", "heading",
         "--num_processes", "1"]))
    promote()
    (root / "pages.lst").write_text("\n".join(str(root / "data" / "page" / f"p{k}.xml") for k in range(n_pages)) + "\n")
    timed("run_feature_generation --separators bb", lambda: run_feature_generation.main(
        ["--pagexml_list", str(root / "pages.lst"), "--separators", "bb"]))
    (root / "eval.lst").write_text("\n".join(str(root / "data" / "json15d2bb" / f"p{k}.json") for k in range(n_pages)) + "\n")
    cwd = os.getcwd()
    os.chdir(root)
    try:
        timed("run_gnn_clustering (dbscan)", lambda: run_gnn_clustering.main(
            ["--model_dir", str(root / "gnn.pb"), "--eval_list", str(root / "eval.lst"), "--out_dir", "out",
             "--input_params", "node_feature_dim=15", "edge_feature_dim=2",
             "node_input_feature_mask=" + str(MASK).replace(" ", ""), "--clustering_method", "dbscan"]))
    finally:
        os.chdir(cwd)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
