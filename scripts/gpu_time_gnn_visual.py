"""Timing of the visual GNN (vn7e2 shape: 7 + 3 x 16 node features, image 683 x 1024 backbone) on the C4 graph,
device-resident entry (development aid).  Prints the whole call and the graph part alone (wide-feature step kernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from citlab_article_separation_new_amd.config import GnnConfig
from citlab_article_separation_new_amd.weights import init_gnn_weights
from citlab_article_separation_new_amd import gnn_io, synth, _lib
cfg = GnnConfig(visual_dims=[16, 16, 16], visual_layers=["scale_0_unet_up_2_conv", "scale_0_unet_up_1_conv", "scale_0_unet_up_0_conv"], mvn=True)
g = gnn_io.GnnGraph(init_gnn_weights(cfg, 1234), cfg)
lib = _lib.init_device(0); h = g.handle(0)
print("step kernel:", gnn_io.step_mode(g))
gr = synth.synth_graph(0); N = gr["num_nodes"]; E = gr["interacting_nodes"].shape[0]
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
e, u, ef = dev(gr["interacting_nodes"]), dev(gr["node_features"]), dev(gr["edge_features"])
H, W, P = 1024, 683, 4
img = dev(synth.synth_page(0, W, H).astype(np.float32))
rng = np.random.default_rng(0)
reg = np.zeros((N, 2, P), np.float32)
for n in range(N):
    x0, y0 = rng.random() * 0.8, rng.random() * 0.8
    reg[n, 0] = [x0, x0 + 0.15, x0 + 0.15, x0]; reg[n, 1] = [y0, y0, y0 + 0.05, y0 + 0.05]
reg, npts = dev(reg), dev(np.full(N, P, np.int32))
out = torch.empty(N * N, 2, device='cuda')
s = torch.cuda.current_stream().cuda_stream
def step(): gnn_io.gnn_forward_visual_dev(g, N, E, e.data_ptr(), u.data_ptr(), ef.data_ptr(), img.data_ptr(), H, W, reg.data_ptr(), P, npts.data_ptr(), N * N, None, out.data_ptr(), s)
ucat = torch.rand(N, 55, device='cuda')
def graph_only(): _lib.check(lib.asep_gnn_forward_dev(h, N, E, e.data_ptr(), ucat.data_ptr(), ef.data_ptr(), N * N, None, out.data_ptr(), s), "gnn")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for name, fn in (("visual GNN (backbone 683x1024 + ROI + graph)", step), ("graph part alone (U = 55, K = 350)", graph_only)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); dt = (time.time() - t0) / iters
    print(f"{name}: {dt*1e6:.1f} us/page")
print(f"graph FLOPs/page: {lib.asep_gnn_flops(h, N, 2*E, N*N)/1e9:.2f} GFLOP")
