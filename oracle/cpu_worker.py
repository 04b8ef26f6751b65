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
    # a cheap page stand-in of the right statistics is enough for timing; the real generator costs 4 s/page
    rng = np.random.default_rng(20261002 + a.page)
    img = np.clip(rng.normal(0.88, 0.03, size=(a.rows, a.width)), 0, 1).astype(np.float32)
    img[::31, :] = 0.2
    aru_oracle.forward_torch(img[:128, :256], w, cfg)        # warm-up
    t0 = time.perf_counter()
    aru_oracle.forward_torch(img, w, cfg)
    t_aru = time.perf_counter() - t0
    t_gnn = 0.0
    if a.gnn:
        gcfg = GnnConfig()
        gw = init_gnn_weights(gcfg, 1234)
        g = synth.synth_graph(a.page)
        t0 = time.perf_counter()
        gnn_oracle.forward(g["num_nodes"], g["interacting_nodes"], g["node_features"], g["edge_features"], None, gw, gcfg)
        t_gnn = time.perf_counter() - t0
    print(json.dumps({"t_aru": t_aru, "t_gnn": t_gnn, "rows": a.rows, "threads": a.threads}), flush=True)


if __name__ == "__main__":
    main()
