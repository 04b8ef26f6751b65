import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from citlab_article_separation_new_amd.config import AruConfig
from citlab_article_separation_new_amd.weights import init_aru_weights
from citlab_article_separation_new_amd import net_post_processing_helper as helper
from oracle import aru_oracle
cfg = AruConfig()
w = init_aru_weights(cfg, 1234, bias_jitter=0.05)
g = helper.AruGraph(w, cfg)
for (H, W) in [(96, 64), (37, 53)]:
    img = np.random.default_rng(1).random((H, W), dtype=np.float32)
    ref, inter = aru_oracle.forward_torch(img, w, cfg, return_intermediates=True)
    out = helper.get_net_output(img, g, "0")
    print(H, W, "prob maxabs", np.abs(out - ref).max(), "ref range", ref.min(), ref.max())
    for name in sorted(inter):
        if name.startswith("scale_") or name.startswith("att_"):
            got = helper.get_endpoint(g, name)
            print("  %-32s %-16s maxabs %.3e  (|ref|max %.3f)" % (name, got.shape, np.abs(got - inter[name]).max(), np.abs(inter[name]).max()))
