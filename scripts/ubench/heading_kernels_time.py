import ctypes as C, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from citlab_article_separation_new_amd import _lib, image_ops
lib, ws = image_ops._workspace(0)
dev = torch.device('cuda', 0)
H, W = 4500, 3000
img = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device=dev)
out = torch.empty((H, W), dtype=torch.uint8, device=dev)
sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for _ in range(3): lib.asep_prep_gray_u8_dev(ws, img.data_ptr(), H, W, out.data_ptr(), sp)
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): lib.asep_prep_gray_u8_dev(ws, img.data_ptr(), H, W, out.data_ptr(), sp)
e1.record(); torch.cuda.synchronize(); print('gray_u8 us', e0.elapsed_time(e1) / 20 * 1e3)
u8 = torch.randint(0, 256, (H, W, 2), dtype=torch.uint8, device=dev)
rng = np.random.default_rng(0)
boxes = []
for c in range(6):
    x0 = 60 + c * 490
    for y in range(60, H - 300, 38): boxes.append([x0, y, x0 + 450, y + 30])
boxes = boxes[:704]
for _ in range(3): image_ops.box_sums_dev(u8.data_ptr(), (H, W, 2), boxes)
t = time.perf_counter()
for _ in range(20): image_ops.box_sums_dev(u8.data_ptr(), (H, W, 2), boxes)
print('box_sums (704 boxes, incl. copies and sync) us', (time.perf_counter() - t) / 20 * 1e6)
