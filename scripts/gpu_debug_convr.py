"""Development aid: where do the end points of the bf16 engine differ between ASEP_BF_CONVR=1 and 0?   python scripts/gpu_debug_convr.py H W"""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
H, W = int(sys.argv[1]), int(sys.argv[2])
names = ["scale_0_unet_down_3_conv", "scale_0_unet_up_3_conv", "scale_1_unet_up_3_conv", "scale_2_unet_up_3_conv"]
if len(sys.argv) > 3:
    from citlab_article_separation_new_amd.config import AruConfig
    from citlab_article_separation_new_amd.weights import init_aru_weights
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    cfg = AruConfig(compute_dtype="bf16")
    g = helper.AruGraph(init_aru_weights(cfg, 12), cfg)
    img = np.random.default_rng(91).random((H, W), dtype=np.float32)
    helper.get_net_output(img, g, "0")
    np.savez(sys.argv[3], **{n: helper.get_endpoint(g, n) for n in names})
    sys.exit(0)
for v in ("1", "0"):
    subprocess.check_call([sys.executable, __file__, str(H), str(W), f"/tmp/convr_dbg_{v}.npz"], env={**os.environ, "ASEP_BF_CONVR": v})
a, b = np.load("/tmp/convr_dbg_1.npz"), np.load("/tmp/convr_dbg_0.npz")
for n in names:
    d = (a[n] != b[n])
    print(n, a[n].shape, "differ", int(d.sum()), "rows", np.nonzero(d.any(axis=(1, 2)))[0].tolist()[:20], "cols", np.nonzero(d.any(axis=(0, 2)))[0].tolist()[:50],
          "channels", np.nonzero(d.any(axis=(0, 1)))[0].tolist()[:70], "max", float(np.abs(a[n] - b[n]).max()))
