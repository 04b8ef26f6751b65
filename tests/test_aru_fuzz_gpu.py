"""Randomised sizes: the engine against the oracle on 40 seeded page shapes between 1 x 1 and 330 x 330 (prime and odd
sizes, shapes around the tile edges of every kernel: 58-column fused tiles, 16 / 32-pixel MFMA tiles, 4 x 32 Winograd
blocks), ARU and RU graphs -- probabilities within 1e-4, the level-0 block outputs within 3e-5 of their magnitude."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _shapes():
    rng = np.random.default_rng(2026)
    out = [(1, 1), (1, 64), (64, 1), (2, 3), (15, 57), (16, 58), (17, 59), (31, 33), (57, 115), (58, 116), (59, 117)]
    while len(out) < 40:
        out.append((int(rng.integers(1, 331)), int(rng.integers(1, 331))))
    return out


@pytest.mark.parametrize("graph", ["ARU", "RU"])
def test_random_page_shapes_match_the_oracle(graph):
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from citlab_article_separation_new_amd.config import AruConfig
    from citlab_article_separation_new_amd.weights import init_aru_weights
    from oracle import aru_oracle
    cfg = AruConfig(graph=graph)
    w = init_aru_weights(cfg, 99, bias_jitter=0.05, logit_scale=0.05)
    g = helper.AruGraph(w, cfg)
    worst = 0.0
    for k, (H, W) in enumerate(_shapes()):
        img = np.random.default_rng(k).random((H, W), dtype=np.float32)
        ref, inter = aru_oracle.forward_torch(img, w, cfg, return_intermediates=True)
        out = helper.get_net_output(img, g, "0")
        err = float(np.abs(out - ref).max())
        worst = max(worst, err)
        assert err <= 1e-4, (H, W, err)
        for name in ("scale_0_unet_down_0_conv", "scale_0_unet_up_0_conv"):
            got = helper.get_endpoint(g, name)
            scale = max(1.0, float(np.abs(inter[name]).max()))
            assert float(np.abs(got - inter[name]).max()) <= 3e-5 * scale, (H, W, name)
    print(f"\n{graph}: 40 shapes, worst max|dp| = {worst:.2e}")
    g.close()
