import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from citlab_article_separation_new_amd import _lib, image_ops
lib, ws = image_ops._workspace(0)
H, W = 4500, 3000
m = np.zeros((H, W), np.uint8)
m[100:4400, 1500:1506] = 255; m[2000:2005, 100:2900] = 255; m[300:310, 200:1200] = 255      # a few separators
d = torch.from_numpy(m).cuda()
keys = torch.empty((2, 1 << 14), dtype=torch.int32, device="cuda"); tot = torch.empty(2, dtype=torch.int64, device="cuda")
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
f = lambda: lib.asep_post_boundary_segments_enqueue_dev(ws, d.data_ptr(), H, W, 255, keys[0].data_ptr(), keys[1].data_ptr(), 1 << 14, tot.data_ptr(), sp)
for _ in range(3): f()
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): f()
e1.record(); torch.cuda.synchronize()
print("boundary segments, 3000x4500 separator mask:", round(e0.elapsed_time(e1) / 50 * 1e3, 1), "us;", tot.tolist(), "segments")
