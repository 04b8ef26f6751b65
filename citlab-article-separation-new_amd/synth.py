"""Seeded synthetic workloads of the shapes BASELINE.json names.

No network, no datasets, no shipped nets: pages, graphs and weights are generated.
"""
import numpy as np


def synth_page(k: int = 0, W: int = 3000, H: int = 4500, seed0: int = 20261002):
    """Newspaper-like uint8 grayscale scan: paper background, 5-7 text columns with glyph boxes, headings,
    vertical separators in gutters and horizontal rules between articles, blur + salt noise."""
    rng = np.random.default_rng(seed0 + k)
    img = np.clip(rng.normal(225, 6, size=(H, W)), 0, 255).astype(np.float32)
    ncol = int(rng.integers(5, 8))
    gutter = 40
    margin = 60
    colw = (W - 2 * margin - (ncol - 1) * gutter) // ncol
    for c in range(ncol):
        x0 = margin + c * (colw + gutter)
        y = margin
        blocks = 0
        next_heading = int(rng.integers(3, 7))
        while y < H - margin - 80:
            nlines = int(rng.integers(4, 41))
            pitch = int(rng.integers(28, 37))
            heading = blocks == next_heading
            if heading:
                next_heading += int(rng.integers(3, 7))
                nlines = int(rng.integers(1, 3))
                pitch = int(rng.integers(70, 100))
            for _ in range(nlines):
                if y + pitch >= H - margin:
                    break
                x = x0
                gh_lo, gh_hi = (40, 71) if heading else (14, 23)
                while x < x0 + colw - 24:
                    gw = int(rng.integers(6, 23)) * (3 if heading else 1)
                    gh = int(rng.integers(gh_lo, gh_hi))
                    if rng.random() < 0.85:
                        img[y + pitch - gh:y + pitch, x:min(x + gw, x0 + colw)] = np.clip(rng.normal(60, 25), 0, 255)
                    x += gw + int(rng.integers(2, 6))
                y += pitch
            blocks += 1
            y += int(rng.integers(20, 60))
            if rng.random() < 0.5 and y < H - margin - 10:
                t = int(rng.integers(2, 4))
                img[y:y + t, x0:x0 + colw] = 30            # horizontal rule between articles
                y += t + int(rng.integers(15, 40))
        if c < ncol - 1 and rng.random() < 0.7:
            t = int(rng.integers(2, 5))
            xs = x0 + colw + gutter // 2
            img[margin:H - margin, xs:xs + t] = 30            # vertical separator in the gutter
    # separable gaussian blur, sigma 0.8
    r = 2
    kx = np.exp(-0.5 * (np.arange(-r, r + 1) / 0.8) ** 2)
    kx /= kx.sum()
    pad = np.pad(img, ((r, r), (0, 0)), mode="edge")
    img = sum(kx[i] * pad[i:i + H] for i in range(2 * r + 1))
    pad = np.pad(img, ((0, 0), (r, r)), mode="edge")
    img = sum(kx[i] * pad[:, i:i + W] for i in range(2 * r + 1))
    salt = rng.random((H, W)) < 0.001
    img[salt] = 255
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def cached_synth_page(k: int = 0, W: int = 3000, H: int = 4500, cache_dir=None):
    """``synth_page`` through a per-seed file cache (the generator costs ~3 s of numpy per 3000 x 4500 page; bench.py's ranks
    and repeated runs on one box share the files).  Writes are atomic (rename), a corrupt file is regenerated."""
    import os
    import tempfile
    d = cache_dir or os.path.join(tempfile.gettempdir(), "asep_synth_cache")
    path = os.path.join(d, f"page_{k}_{W}x{H}.npy")
    try:
        a = np.load(path)
        if a.shape == (H, W) and a.dtype == np.uint8:
            return a
    except Exception:
        pass
    a = synth_page(k, W, H)
    try:
        os.makedirs(d, exist_ok=True)
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            np.save(f, a)
        os.replace(tmp, path)
    except OSError:
        pass
    return a


def visual_inputs(page_u8, N: int, k: int = 0, max_dim: int = 1024, P: int = 4, seed0: int = 977):
    """Image feeds of the visual relation net for one synthetic scan: the page resized like the GNN input pipeline does
    (longer side -> ``max_dim``, TF1 bilinear without half-pixel offset, image_resizer.py:197-223; values 0..255 as fed,
    input_dataset.py:279-280) and one rectangular visual region per node in RELATIVE coordinates [N,2,P]."""
    from .gnn_input import compute_new_size, resize_bilinear_tf1
    nh, nw = compute_new_size(page_u8.shape[0], page_u8.shape[1], 256, max_dim)
    small = resize_bilinear_tf1(page_u8.astype(np.float32), nh, nw)[:, :, 0]
    rng = np.random.default_rng(seed0 + k)
    reg = np.zeros((N, 2, P), np.float32)
    for n in range(N):
        bx, by = rng.random() * 0.8, rng.random() * 0.8
        reg[n, 0] = [bx, bx + 0.15, bx + 0.15, bx]
        reg[n, 1] = [by, by, by + 0.05, by + 0.05]
    return np.ascontiguousarray(small, dtype=np.float32), reg, np.full(N, P, np.int32)


def synth_graph(k: int = 0, N: int = 200, n_pairs: int = 10000, node_dim: int = 7, edge_dim: int = 2,
                seed0: int = 4321):
    """C4 graph: N nodes, `n_pairs` distinct unordered pairs emitted in one direction and shuffled
    (-> 2*n_pairs directed edges after correction), U(0,1) node features, Bernoulli(0.15) edge features."""
    rng = np.random.default_rng(seed0 + k)
    iu, ju = np.triu_indices(N, k=1)
    n_pairs = min(n_pairs, iu.shape[0])
    sel = rng.choice(iu.shape[0], size=n_pairs, replace=False)
    a, b = iu[sel], ju[sel]
    flip = rng.random(n_pairs) < 0.5
    edges = np.stack([np.where(flip, b, a), np.where(flip, a, b)], axis=1).astype(np.int32)
    rng.shuffle(edges, axis=0)
    node_feat = rng.random((N, node_dim), dtype=np.float32)
    edge_feat = (rng.random((n_pairs, edge_dim)) < 0.15).astype(np.float32)
    return {"num_nodes": N, "interacting_nodes": edges, "node_features": node_feat, "edge_features": edge_feat}


def synth_page_xml(path, W: int = 3000, H: int = 4500, k: int = 0, columns: int = 6):
    """A PAGE-XML with the text lines of a newspaper page for the heading pipeline's benchmarks: ``columns`` columns of text
    regions of 4-12 lines each, about 700 lines on 3000 x 4500 (every line with outline and baseline).  -> number of lines"""
    rng = np.random.default_rng(1000 + k)
    colw = (W - 120 - (columns - 1) * 40) // columns
    regs, rid, n_lines = [], 0, 0
    for c in range(columns):
        x0, y = 60 + c * (colw + 40), 60
        while y < H - 300:
            nl, pitch = int(rng.integers(4, 13)), int(rng.integers(28, 37))
            y1 = y + nl * pitch
            lines = "".join(
                f'<TextLine id="r{rid}l{i}"><Coords points="{x0},{y + i * pitch} {x0 + colw},{y + i * pitch} '
                f'{x0 + colw},{y + (i + 1) * pitch - 4} {x0},{y + (i + 1) * pitch - 4}"/>'
                f'<Baseline points="{x0},{y + (i + 1) * pitch - 8} {x0 + colw},{y + (i + 1) * pitch - 8}"/></TextLine>'
                for i in range(nl))
            regs.append(f'<TextRegion id="r{rid}"><Coords points="{x0},{y} {x0 + colw},{y} {x0 + colw},{y1} {x0},{y1}"/>'
                        + lines + '</TextRegion>')
            rid += 1
            n_lines += nl
            y = y1 + int(rng.integers(20, 60))
    with open(path, "w") as f:
        f.write('<?xml version="1.0" encoding="UTF-8"?>\n<PcGts xmlns="http://schema.primaresearch.org/PAGE/gts/'
                'pagecontent/2013-07-15"><Metadata><Creator>t</Creator><Created>2020-01-01T00:00:00</Created>'
                '<LastChange>2020-01-01T00:00:00</LastChange></Metadata>'
                f'<Page imageFilename="x.png" imageWidth="{W}" imageHeight="{H}">' + "".join(regs) + '</Page></PcGts>')
    return n_lines


GNN_VISUAL_LAYERS = ["scale_0_unet_up_2_conv", "scale_0_unet_up_1_conv", "scale_0_unet_up_0_conv"]
GNN_FEATURE_MASK = [1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1]          # the 7 of the 15 json features the nets read


def write_gnn_cli_inputs(root, n_pages, visual=True, W: int = 3000, H: int = 4500, N: int = 200):
    """Inputs of the relation net's command line for benchmarks: a frozen graph (random weights, the visual net BASELINE
    configs[3] names or the geometric one), per page a graph json of ``N`` text blocks / ~20k directed edges (+ the scan for the
    visual net) and a PAGE-XML with ``N`` text regions.  Four distinct pages, the rest links to them.
    -> argv for ``run_gnn_clustering.main`` (without --num_workers / --out_dir)"""
    import json
    import os
    from PIL import Image
    from . import pb_import
    from .config import GnnConfig
    from .weights import init_gnn_weights
    cfg = GnnConfig(node_feature_dim=7, visual_dims=[16, 16, 16] if visual else [], visual_layers=GNN_VISUAL_LAYERS if visual else [])
    w = init_gnn_weights(cfg, 3, bias_jitter=0.05)
    keep = [i for i, m in enumerate(GNN_FEATURE_MASK) if m]
    os.makedirs(os.path.join(root, "model", "export"))
    with open(os.path.join(root, "model", "export", "gnn_best_1.pb"), "wb") as f:
        f.write(pb_import.weights_to_graphdef(w, "graph/", meta={"num_transition_steps": cfg.num_transition_steps}))
    data = os.path.join(root, "data")
    os.makedirs(os.path.join(data, "page"))
    os.makedirs(os.path.join(data, "json15d2bb"))
    jsons = []
    for k in range(n_pages):
        name = f"p{k:03d}"
        if k < 4:
            g = synth_graph(k, N=N, n_pairs=10000, node_dim=7)
            feats15 = np.zeros((N, 15), np.float32)
            feats15[:, keep] = g["node_features"]
            d = {"num_nodes": N, "interacting_nodes": g["interacting_nodes"].tolist(),
                 "num_interacting_nodes": int(g["interacting_nodes"].shape[0]), "node_features": feats15.tolist(),
                 "edge_features": g["edge_features"].tolist(), "gt_relations": [], "gt_num_relations": 0}
            if visual:
                page = cached_synth_page(k, W, H)
                _, regions, npts = visual_inputs(page, N, k)
                d["visual_regions_nodes"] = np.asarray(regions).tolist()
                d["num_points_visual_regions_nodes"] = np.asarray(npts).tolist()
                Image.fromarray(page).save(os.path.join(data, f"{name}.png"), compress_level=1)
            with open(os.path.join(data, "json15d2bb", f"{name}.json"), "w") as f:
                json.dump(d, f)
        else:
            os.symlink(os.path.join(data, "json15d2bb", f"p{k % 4:03d}.json"), os.path.join(data, "json15d2bb", f"{name}.json"))
            if visual:
                os.symlink(os.path.join(data, f"p{k % 4:03d}.png"), os.path.join(data, f"{name}.png"))
        regs = []
        for i in range(N):
            x, y = 60 + (i % 5) * 580, 60 + (i // 5) * 105
            regs.append(f'<TextRegion id="tr{i}"><Coords points="{x},{y} {x+540},{y} {x+540},{y+90} {x},{y+90}"/>'
                        + "".join(f'<TextLine id="tr{i}l{j}"><Coords points="{x},{y+30*j} {x+540},{y+30*j} {x+540},{y+30*j+28} '
                                  f'{x},{y+30*j+28}"/></TextLine>' for j in range(3)) + '</TextRegion>')
        with open(os.path.join(data, "page", f"{name}.xml"), "w") as f:
            f.write('<?xml version="1.0" encoding="UTF-8"?>\n<PcGts xmlns="http://schema.primaresearch.org/PAGE/gts/'
                    'pagecontent/2013-07-15"><Metadata><Creator>t</Creator><Created>2020-01-01T00:00:00</Created>'
                    '<LastChange>2020-01-01T00:00:00</LastChange></Metadata><Page imageFilename="x.png" '
                    f'imageWidth="{W}" imageHeight="{H}">' + "".join(regs) + '</Page></PcGts>')
        jsons.append(os.path.join(data, "json15d2bb", f"{name}.json"))
    lst = os.path.join(root, "eval.lst")
    with open(lst, "w") as f:
        f.write("\n".join(jsons) + "\n")
    argv = ["--model_dir", os.path.join(root, "model"), "--eval_list", lst, "--input_params", "node_feature_dim=15",
            "edge_feature_dim=2", "node_input_feature_mask=" + str(GNN_FEATURE_MASK).replace(" ", ""), "--clustering_method",
            "dbscan"]
    if visual:
        argv += ["--image_input", "True", "--visual_layers"] + GNN_VISUAL_LAYERS
    return argv
