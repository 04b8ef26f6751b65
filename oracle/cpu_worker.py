"""CPU ORACLE worker (test/measurement infrastructure, NOT product code).

One process of bench.py's `cpu_baseline` leg: runs the torch-CPU ARU-Net oracle on a horizontal band of a
synthetic page (and optionally the numpy GNN oracle on one graph) with a fixed thread count and prints the
wall time as JSON.  bench.py starts several of these side by side, mirroring the reference's process fan-out
(run_net_post_processing.py:61-82: ProcessPoolExecutor over page sub-lists).

    python -m oracle.cpu_worker --threads 16 --page 0 --rows 1500 --width 3000 --height 4500 [--gnn]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--page", type=int, default=0)
    ap.add_argument("--rows", type=int, default=1500)
    ap.add_argument("--width", type=int, default=3000)
    ap.add_argument("--height", type=int, default=4500)
    ap.add_argument("--gnn", action="store_true")
    ap.add_argument("--visual", action="store_true", help="the visual relation net (backbone on 683 x 1024 + 55 features)")
    a = ap.parse_args()
    os.environ["OMP_NUM_THREADS"] = str(a.threads)
    import numpy as np
    import torch
    torch.set_num_threads(a.threads)
    from citlab_article_separation_new_amd import synth
    from citlab_article_separation_new_amd.config import AruConfig, GnnConfig
    from citlab_article_separation_new_amd.weights import init_aru_weights, init_gnn_weights
    from oracle import aru_oracle, gnn_oracle
    cfg = AruConfig()
    w = init_aru_weights(cfg, 1234)
    # the same synthetic scan the GPU path is timed on (generated outside the timed region); a band of it if rows < height
    page = synth.synth_page(a.page, a.width, a.height)
    img = (page[:a.rows].astype(np.float32) / np.float32(255.0))
    aru_oracle.forward_torch(img[:256, :512], w, cfg)        # warm-up (thread pool, first-touch allocations)
    t0 = time.perf_counter()
    aru_oracle.forward_torch(img, w, cfg)
    t_aru = time.perf_counter() - t0
    t_gnn = 0.0
    if a.gnn:
        g = synth.synth_graph(a.page)
        if a.visual:
            gcfg = GnnConfig(visual_dims=[16, 16, 16], mvn=True,
                             visual_layers=["scale_0_unet_up_2_conv", "scale_0_unet_up_1_conv", "scale_0_unet_up_0_conv"])
            gw = init_gnn_weights(gcfg, 1234)
            small, regions, npts = synth.visual_inputs(page, g["num_nodes"], a.page)
            t0 = time.perf_counter()
            gnn_oracle.forward_visual(g["num_nodes"], g["interacting_nodes"], g["node_features"], g["edge_features"], small, regions,
                                      npts, None, gw, gcfg)
        else:
            gcfg = GnnConfig()
            gw = init_gnn_weights(gcfg, 1234)
            t0 = time.perf_counter()
            gnn_oracle.forward(g["num_nodes"], g["interacting_nodes"], g["node_features"], g["edge_features"], None, gw, gcfg)
        t_gnn = time.perf_counter() - t0
    print(json.dumps({"t_aru": t_aru, "t_gnn": t_gnn, "rows": a.rows, "threads": a.threads}), flush=True)


if __name__ == "__main__":
    main()
