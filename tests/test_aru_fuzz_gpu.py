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


def _bf16_shapes():
    rng = np.random.default_rng(515)
    # around the tile edges of the bf16 kernels: 16 x 64 (attention head), 8 rows x 64 columns of the half-resolution input (level-0 deconvolution:
    # 16 x 128 output pixels), 16 x 32 (fused blocks), odd sizes (the deconvolutions' pad_before = 1 cases), and tiny pages
    out = [(1, 1), (2, 3), (5, 130), (16, 64), (17, 65), (31, 127), (33, 129), (48, 257), (15, 63)]
    while len(out) < 16:
        out.append((int(rng.integers(1, 200)), int(rng.integers(1, 300))))
    return out


def test_bf16_random_page_shapes_block_by_block():
    """Round 5 (att_headb_kernel, deconvb8_kernel, the pipelined fused blocks, the difference filter of combine_kernel): the bf16 engine on 16 page
    shapes around its kernels' tile edges against the oracle WITH the engine's roundings, block by block on the engine's own upstream tensors
    (tests/test_aru_gpu.py has the gates and three fixed sizes) -- every end point, and the probabilities against the free-running emulation."""
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from citlab_article_separation_new_amd.config import AruConfig
    from citlab_article_separation_new_amd.weights import init_aru_weights
    from oracle import aru_oracle
    BF16_BLOCK_MAX_GATE, BF16_BLOCK_RMS_GATE, BF16_EMU_PROB_GATE = 1.2e-2, 5e-4, 2e-3       # the gates of tests/test_aru_gpu.py (b), (a)
    cfg = AruConfig(compute_dtype="bf16")
    cfg32 = AruConfig()
    w = init_aru_weights(cfg, 99, bias_jitter=0.05, logit_scale=0.05)
    g = helper.AruGraph(w, cfg)
    worst = (0.0, 0.0, 0.0)
    for k, (H, W) in enumerate(_bf16_shapes()):
        img = np.random.default_rng(100 + k).random((H, W), dtype=np.float32)
        out = helper.get_net_output(img, g, "0")
        ref, inter = aru_oracle.forward_torch(img, w, cfg32, return_intermediates=True, storage="bf16")
        names = [n for n in sorted(inter) if n.startswith("scale_") or n.startswith("att_")]
        eng = {n: helper.get_endpoint(g, n) for n in names}
        _, forced = aru_oracle.forward_torch(img, w, cfg32, return_intermediates=True, storage="bf16", teacher=eng)
        for n in names:
            assert eng[n].shape == forced[n].shape, (H, W, n)
            scale = max(1.0, float(np.abs(forced[n]).max()))
            d = eng[n] - forced[n]
            dm, dr = float(np.abs(d).max()) / scale, float(np.sqrt(np.mean(d.astype(np.float64) ** 2))) / scale
            assert dm <= BF16_BLOCK_MAX_GATE and dr <= BF16_BLOCK_RMS_GATE, (H, W, n, dm, dr)
            worst = (max(worst[0], dm), max(worst[1], dr), worst[2])
        perr = float(np.abs(out - ref).max())
        assert perr <= BF16_EMU_PROB_GATE, (H, W, perr)
        worst = (worst[0], worst[1], max(worst[2], perr))
    print(f"\nbf16, 16 shapes: block by block max {worst[0]:.2e} rms {worst[1]:.2e} of max|ref|; probabilities against the free-running emulation {worst[2]:.2e}")
    g.close()
