"""Quick timing of the ARU-Net engine on one device-resident page (development aid)."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from citlab_article_separation_new_amd.config import AruConfig
from citlab_article_separation_new_amd.weights import init_aru_weights
from citlab_article_separation_new_amd import net_post_processing_helper as helper, _lib
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4500, 3000)
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dtype = sys.argv[4] if len(sys.argv) > 4 else 'f32'
levels = int(sys.argv[5]) if len(sys.argv) > 5 else 5
att = int(sys.argv[6]) if len(sys.argv) > 6 else 3
cfg = AruConfig(compute_dtype=dtype, scale_space_num=levels, num_scales_att=att)
g = helper.AruGraph(init_aru_weights(cfg, 1234), cfg)
lib = _lib.init_device(0)
h = g.handle(0)
img = torch.rand(H, W, device='cuda')
out = torch.empty(H, W, cfg.n_classes, device='cuda')
u8 = torch.empty(H, W, cfg.n_classes, device='cuda', dtype=torch.uint8)
s = torch.cuda.current_stream().cuda_stream
def step():
    _lib.check(lib.asep_aru_forward_dev(h, img.data_ptr(), H, W, out.data_ptr(), u8.data_ptr(), None, 0.05, s), "fwd")
step(); torch.cuda.synchronize()
t0 = time.time()
for _ in range(iters): step()
torch.cuda.synchronize()
dt = (time.time() - t0) / iters
fl = lib.asep_aru_flops(h, H, W)
print(f"{H}x{W} [{dtype}, {levels} levels, {att} attention scales]: {dt*1e3:.2f} ms/page  {1/dt:.2f} pages/s  {fl/1e9:.1f} GFLOP -> {fl/dt/1e12:.1f} TFLOP/s  mem {torch.cuda.memory_allocated()/1e9:.2f} GB torch")
