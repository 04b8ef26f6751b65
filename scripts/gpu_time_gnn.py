"""Timing of the GNN engine on the C4 graph (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from citlab_article_separation_new_amd.config import GnnConfig
from citlab_article_separation_new_amd.weights import init_gnn_weights
from citlab_article_separation_new_amd import gnn_io, synth, _lib
cfg = GnnConfig()
g = gnn_io.GnnGraph(init_gnn_weights(cfg, 1234), cfg)
lib = _lib.init_device(0); h = g.handle(0)
gr = synth.synth_graph(0); N = gr["num_nodes"]; E = gr["interacting_nodes"].shape[0]
e = torch.from_numpy(gr["interacting_nodes"]).cuda(); u = torch.from_numpy(gr["node_features"]).cuda(); ef = torch.from_numpy(gr["edge_features"]).cuda()
out = torch.empty(N * N, 2, device='cuda')
s = torch.cuda.current_stream().cuda_stream
def step(): _lib.check(lib.asep_gnn_forward_dev(h, N, E, e.data_ptr(), u.data_ptr(), ef.data_ptr(), N * N, None, out.data_ptr(), s), "gnn")
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.time()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for _ in range(iters): step()
torch.cuda.synchronize(); dt = (time.time() - t0) / iters
print(f"GNN C4: {dt*1e6:.1f} us/page, {lib.asep_gnn_flops(h, N, 2*E, N*N)/dt/1e12:.2f} TFLOP/s")
