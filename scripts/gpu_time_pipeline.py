"""Stage timing of the separator / heading pipelines on one device-resident 3000x4500 page (development aid and the
source of the classical-stage numbers in DESIGN.md).  Usage: python scripts/gpu_time_pipeline.py [iters]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from citlab_article_separation_new_amd import _lib, image_ops, polygonize, synth
from citlab_article_separation_new_amd.config import AruConfig
from citlab_article_separation_new_amd.net_post_processing_helper import AruGraph
from citlab_article_separation_new_amd.weights import init_aru_weights

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
W, H = 3000, 4500
cfg = AruConfig()
graph = AruGraph(init_aru_weights(cfg, 1234, logit_scale=0.05), cfg)
lib = _lib.init_device(0)
_, ws = image_ops._workspace(0)
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
gray = synth.synth_page(0, W=W, H=H)
bgr = np.repeat(gray[:, :, None], 3, axis=2)
d_bgr = torch.from_numpy(bgr).cuda()
d_gray_u8 = torch.from_numpy(gray).cuda()


def timed(name, fn, n=iters):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    print(f"{name:44s} {dt * 1e3:9.3f} ms")
    return dt


for fixed_h, label in ((1500, "separator default (1/3)"), (4500, "full resolution (C2)")):
    sc = fixed_h / H
    h, w = image_ops.scaled_size(H, W, sc)
    d_g = torch.empty((h, w), dtype=torch.float32, device="cuda")
    d_out = torch.empty((h, w, 2), dtype=torch.float32, device="cuda")
    d_u8 = torch.empty((h, w, 2), dtype=torch.uint8, device="cuda")
    d_mask = torch.empty((h, w, 2), dtype=torch.uint8, device="cuda")
    d_hz = torch.empty((h, w), dtype=torch.uint8, device="cuda")
    d_vt = torch.empty((h, w), dtype=torch.uint8, device="cuda")
    print(f"--- {label}: net input {w}x{h}")
    timed("a1  resize + gray (BGR u8 -> f32)", lambda: _lib.check(lib.asep_prep_scale_gray_dev(
        ws, d_bgr.data_ptr(), H, W, 3, sc, None, d_g.data_ptr(), sp), "prep"))
    timed("a3  ARU-Net forward (+u8 +mask epilogue)", lambda: _lib.check(lib.asep_aru_forward_dev(
        graph.handle(0), d_g.data_ptr(), h, w, d_out.data_ptr(), d_u8.data_ptr(), d_mask.data_ptr(), 0.5, sp), "aru"))
    # a synthetic separator-like mask so the classical stages see realistic sparsity
    rng = np.random.default_rng(0)
    m = np.zeros((h, w, 2), np.uint8)
    for i in range(30):
        y = int(rng.integers(0, h - 4)); m[y:y + 3, int(rng.integers(0, w // 3)):int(rng.integers(w // 2, w)), 0] = 255
        x = int(rng.integers(0, w - 4)); m[int(rng.integers(0, h // 3)):int(rng.integers(h // 2, h)), x:x + 3, 0] = 255
    m[rng.random((h, w)) < 0.001, 0] = 255
    d_m = torch.from_numpy(m).cuda()
    size = h * w
    ks = image_ops.separator_kernel_sizes(h, w)
    timed("a9  CC filter + openings + subtract", lambda: _lib.check(lib.asep_post_separator_dev(
        ws, d_m.data_ptr(), h, w, 2, 0, int(size * (1 / size * 100)), ks[0], ks[1], ks[2], d_hz.data_ptr(),
        d_vt.data_ptr(), sp), "sep"))
    hz, vt = d_hz.cpu().numpy(), d_vt.cpu().numpy()
    t0 = time.time()
    for _ in range(3):
        d_hz.cpu(); d_vt.cpu()
    print(f"{'    D2H of the two masks':44s} {(time.time() - t0) / 3 * 1e3:9.3f} ms")
    t0 = time.time()
    p = polygonize.shapes(hz) + polygonize.shapes(vt)
    print(f"{'a10 polygon rings, host only (from masks)':44s} {(time.time() - t0) * 1e3:9.3f} ms  ({len(p)} polygons)")
    image_ops.boundary_segments_dev(d_hz.data_ptr(), h, w, 255, 0, sp)
    t0 = time.time()
    kh = image_ops.boundary_segments_dev(d_hz.data_ptr(), h, w, 255, 0, sp)
    kv = image_ops.boundary_segments_dev(d_vt.data_ptr(), h, w, 255, 0, sp)
    t1 = time.time()
    q = polygonize.shapes_from_segments(*kh, h, w) + polygonize.shapes_from_segments(*kv, h, w)
    t2 = time.time()
    assert q == p
    print(f"{'a10 GPU boundary segments + D2H':44s} {(t1 - t0) * 1e3:9.3f} ms  ({kh[0].size + kv[0].size} segments)")
    print(f"{'a10 ring chaining on the host':44s} {(t2 - t1) * 1e3:9.3f} ms")

print("--- heading: stroke-width distance transform at full resolution")
d_swt = torch.empty((H, W), dtype=torch.uint8, device="cuda")
timed("a12 255-gray, blur, Otsu, exact EDT", lambda: _lib.check(lib.asep_swt_distance_transform_dev(
    ws, d_gray_u8.data_ptr(), H, W, d_swt.data_ptr(), sp), "swt"))

# per text line statistics: 3000 line crops of ~600 x 40 px on the device-resident distance transform
rng = np.random.default_rng(1)
boxes = []
for _ in range(3000):
    w_, h_ = int(rng.integers(300, 900)), int(rng.integers(28, 50))
    x_, y_ = int(rng.integers(0, W - w_)), int(rng.integers(0, H - h_))
    boxes.append([x_, y_, x_ + w_ + 1, y_ + h_ + 1])
dimg = image_ops.DeviceImage(d_swt, 0)
image_ops.swt_line_features(dimg, boxes)
t0 = time.time()
sw, hh = image_ops.swt_line_features(dimg, boxes)
t1 = time.time()
print(f"{'a11 per-line CC statistics, 3000 lines (GPU)':44s} {(t1 - t0) * 1e3:9.3f} ms")
swt_host = d_swt.cpu().numpy()
t0 = time.time()
for b in boxes[:300]:
    image_ops._line_features_host(swt_host, b)
print(f"{'    same on the host (scipy), scaled x10':44s} {(time.time() - t0) * 1e4:9.3f} ms")
