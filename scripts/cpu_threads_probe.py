import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from citlab_article_separation_new_amd.config import AruConfig
from citlab_article_separation_new_amd.weights import init_aru_weights
from oracle import aru_oracle
cfg = AruConfig(); w = init_aru_weights(cfg, 1234)
img = np.random.default_rng(0).random((1500, 1000)).astype(np.float32)
print("cpu_count", os.cpu_count())
for th in [8, 16, 32, 64, 128]:
    torch.set_num_threads(th)
    aru_oracle.forward_torch(img[:256,:256], w, cfg)
    t = time.time(); aru_oracle.forward_torch(img, w, cfg); dt = time.time() - t
    print(th, "threads: 1500x1000 in %.2fs -> full page est %.1fs" % (dt, dt * 9), flush=True)
