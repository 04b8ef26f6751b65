"""Cost of the drop-in seam `get_net_output(image, graph)` on host arrays vs. the device-resident forward (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from citlab_article_separation_new_amd import _lib, net_post_processing_helper as helper, synth
from citlab_article_separation_new_amd.config import AruConfig
from citlab_article_separation_new_amd.weights import init_aru_weights
cfg = AruConfig()
g = helper.AruGraph(init_aru_weights(cfg, 1234), cfg)
for (H, W) in ((4500, 3000), (1500, 1000), (768, 512)):
    img64 = synth.synth_page(0, W, H) / 255.0
    img32 = img64.astype(np.float32)
    for name, img in (("float64", img64), ("float32", img32)):
        helper.get_net_output(img, g, "0")
        t0 = time.perf_counter()
        for _ in range(8):
            out = helper.get_net_output(img, g, "0")
        dt = (time.perf_counter() - t0) / 8
        print(f"get_net_output({name} {W}x{H}): {dt * 1e3:7.2f} ms/page")
    lib = _lib.init_device(0); h = g.handle(0)
    d_in = torch.from_numpy(img32).cuda(); d_out = torch.empty(H, W, 2, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    step = lambda: _lib.check(lib.asep_aru_forward_dev(h, d_in.data_ptr(), H, W, d_out.data_ptr(), None, None, 0.05, s), "fwd")
    step(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): step()
    torch.cuda.synchronize()
    print(f"device-resident forward {W}x{H}:        {(time.perf_counter() - t0) / 8 * 1e3:7.2f} ms/page")
    ref = d_out.cpu().numpy()
    assert np.array_equal(ref, helper.get_net_output(img32, g, "0")), "seam result differs from the device-resident call"
