#!/usr/bin/env python3
"""sha256 of the engine's outputs on a whole 3000 x 4500 synthetic page for every arithmetic (probabilities, uint8 map, one deep end point): run once per
library build (ASEP_HIP_LIB=<other libasep_hip.so>) and compare the lines -- bit-identity of a refactoring against the previous round's library
("outputs bit-identical to r4 on the whole-frame fixtures").   usage: python scripts/output_hash.py [H W]"""
import hashlib
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from citlab_article_separation_new_amd import net_post_processing_helper as helper, synth
from citlab_article_separation_new_amd.config import AruConfig
from citlab_article_separation_new_amd.weights import init_aru_weights

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4500, 3000)
page = synth.synth_page(0, W, H).astype(np.float32) / 255.0
for dt in ("f32", "f32s", "bf16"):
    cfg = AruConfig(compute_dtype=dt)
    w = init_aru_weights(cfg, 1234, bias_jitter=0.05, logit_scale=0.05)
    g = helper.AruGraph(w, cfg)
    out, u8, mask = helper.get_net_output_fused(page, g, "0", want_u8=True, threshold=0.5)
    ep = helper.get_endpoint(g, "scale_1_unet_up_2_conv")
    h = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
    print(dt, h(out), h(u8), h(ep), os.environ.get("ASEP_HIP_LIB", "tree"))
    g.close()
