"""Files in, files out for the relation net: graph jsons (+ scans for the visual net) + PAGE-XML -> run_gnn_clustering command
line -> PAGE-XML with article ids.  200 text blocks / ~20k directed edges / 40k pairs per page (BASELINE configs[3]).

    python scripts/e2e_gnn_bench.py [n_pages=64] [workers=8] [visual=1]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

n_pages = int(sys.argv[1]) if len(sys.argv) > 1 else 64
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
visual = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
W, H, N = 3000, 4500, 200
LAYERS = ["scale_0_unet_up_2_conv", "scale_0_unet_up_1_conv", "scale_0_unet_up_0_conv"]


def write_page_xml(path, n):
    regs = []
    for i in range(n):
        x, y = 60 + (i % 5) * 580, 60 + (i // 5) * 105
        regs.append(f'<TextRegion id="tr{i}"><Coords points="{x},{y} {x+540},{y} {x+540},{y+90} {x},{y+90}"/>'
                    + "".join(f'<TextLine id="tr{i}l{j}"><Coords points="{x},{y+30*j} {x+540},{y+30*j} {x+540},{y+30*j+28} {x},{y+30*j+28}"/>'
                              f'</TextLine>' for j in range(3)) + '</TextRegion>')
    with open(path, "w") as f:
        f.write('<?xml version="1.0" encoding="UTF-8"?>\n<PcGts xmlns="http://schema.primaresearch.org/PAGE/gts/'
                'pagecontent/2013-07-15"><Metadata><Creator>t</Creator><Created>2020-01-01T00:00:00</Created>'
                '<LastChange>2020-01-01T00:00:00</LastChange></Metadata><Page imageFilename="x.png" '
                f'imageWidth="{W}" imageHeight="{H}">' + "".join(regs) + '</Page></PcGts>')


def main():
    from PIL import Image
    from citlab_article_separation_new_amd import pb_import, run_gnn_clustering, synth
    from citlab_article_separation_new_amd.config import GnnConfig
    from citlab_article_separation_new_amd.weights import init_gnn_weights
    cfg = GnnConfig(node_feature_dim=7, visual_dims=[16, 16, 16] if visual else [], visual_layers=LAYERS if visual else [])
    w = init_gnn_weights(cfg, 3, bias_jitter=0.05)
    mask = [1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1]
    keep = [i for i, m in enumerate(mask) if m]
    with tempfile.TemporaryDirectory(prefix="asep_gnn_e2e_") as tmp:
        os.makedirs(os.path.join(tmp, "model", "export"))
        with open(os.path.join(tmp, "model", "export", "gnn_best_1.pb"), "wb") as f:
            f.write(pb_import.weights_to_graphdef(w, "graph/", meta={"num_transition_steps": cfg.num_transition_steps}))
        data = os.path.join(tmp, "data")
        os.makedirs(os.path.join(data, "page"))
        os.makedirs(os.path.join(data, "json15d2bb"))
        jsons = []
        t0 = time.perf_counter()
        for k in range(n_pages):
            name = f"p{k:03d}"
            if k < 4:
                g = synth.synth_graph(k, N=N, n_pairs=10000, node_dim=7)
                feats15 = np.zeros((N, 15), np.float32)
                feats15[:, keep] = g["node_features"]
                d = {"num_nodes": N, "interacting_nodes": g["interacting_nodes"].tolist(),
                     "num_interacting_nodes": int(g["interacting_nodes"].shape[0]), "node_features": feats15.tolist(),
                     "edge_features": g["edge_features"].tolist(), "gt_relations": [], "gt_num_relations": 0}
                if visual:
                    page = synth.cached_synth_page(k, W, H)
                    _, regions, npts = synth.visual_inputs(page, N, k)
                    d["visual_regions_nodes"] = np.asarray(regions).tolist()
                    d["num_points_visual_regions_nodes"] = np.asarray(npts).tolist()
                    Image.fromarray(page).save(os.path.join(data, f"{name}.png"), compress_level=1)
                with open(os.path.join(data, "json15d2bb", f"{name}.json"), "w") as f:
                    json.dump(d, f)
            else:
                os.symlink(os.path.join(data, "json15d2bb", f"p{k % 4:03d}.json"), os.path.join(data, "json15d2bb", f"{name}.json"))
                if visual:
                    os.symlink(os.path.join(data, f"p{k % 4:03d}.png"), os.path.join(data, f"{name}.png"))
            write_page_xml(os.path.join(data, "page", f"{name}.xml"), N)
            jsons.append(os.path.join(data, "json15d2bb", f"{name}.json"))
        lst = os.path.join(tmp, "eval.lst")
        with open(lst, "w") as f:
            f.write("\n".join(jsons) + "\n")
        print(f"inputs written in {time.perf_counter() - t0:.1f} s; json {os.path.getsize(jsons[0]) / 1e6:.2f} MB per page")
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            for nw in sorted({1, workers}):
                argv = ["--model_dir", os.path.join(tmp, "model"), "--eval_list", lst, "--out_dir", f"out{nw}",
                        "--input_params", "node_feature_dim=15", "edge_feature_dim=2",
                        "node_input_feature_mask=" + str(mask).replace(" ", ""), "--clustering_method", "dbscan", "--gpu_devices", "0",
                        "--num_workers", str(nw)]
                if visual:
                    argv += ["--image_input", "True", "--visual_layers"] + LAYERS
                part = jsons if nw > 1 else jsons[: max(8, n_pages // 8)]
                with open(lst, "w") as f:
                    f.write("\n".join(part) + "\n")
                t0 = time.perf_counter()
                outs = run_gnn_clustering.main(argv)
                dt = time.perf_counter() - t0
                print(f"run_gnn_clustering, {'visual' if visual else 'geometric'} net, {nw:2d} worker(s): {len(outs)} pages in {dt:.2f} s = "
                      f"{len(outs) / dt:.1f} pages/s ({dt / len(outs) * 1e3:.1f} ms/page incl. start-up)")
        finally:
            os.chdir(cwd)


if __name__ == "__main__":
    main()
