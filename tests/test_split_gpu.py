"""GPU parity of the fp32 path with SPLIT PRODUCTS (compute_dtype "f32s", csrc/split_kernels.h): fp32 tensors and accumulation, every
product of the convolutions with >= 12 input channels computed as six bf16 x bf16 partial products of the three-way bfloat16 split of
both factors on v_mfma_f32_16x16x32_bf16.  The path is held to the FP32 gates (end points 2e-5 * max|ref|, probabilities 1e-4 against
the CPU oracle), not to the bf16 ones, and must stay within a few fp32 roundings of the engine's plain fp32 path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PROB_TOL = 1e-4
ENDPOINT_GATE = 2e-5            # max|d| / max(1, max|ref|) per end point: the gate of tests/test_aru_gpu.py for the fp32 path
AGREE_GATE = 4e-6               # f32s against the engine's own fp32 path, same relative measure (measured <= 1e-6)


def _setup(cfg_kwargs=None, seed=1234, logit_scale=0.05):
    from citlab_article_separation_new_amd.config import AruConfig
    from citlab_article_separation_new_amd.weights import init_aru_weights
    from citlab_article_separation_new_amd.net_post_processing_helper import AruGraph
    cfg = AruConfig(**{**(cfg_kwargs or {}), "compute_dtype": "f32s"})
    w = init_aru_weights(cfg, seed, bias_jitter=0.05, logit_scale=logit_scale)
    return cfg, w, AruGraph(w, cfg)


def _image(H, W, seed):
    rng = np.random.default_rng(seed)
    img = rng.random((H, W), dtype=np.float32)
    img[H // 3:H // 3 + 2, :] = 0.05
    return img


def _check_endpoints(helper, graph, inter, gate=ENDPOINT_GATE):
    worst = ("", 0.0)
    for name in sorted(inter):
        if name.startswith("scale_") or name.startswith("att_"):
            got = helper.get_endpoint(graph, name)
            want = inter[name]
            assert got.shape == want.shape, name
            rel = float(np.abs(got - want).max()) / max(1.0, float(np.abs(want).max()))
            worst = max(worst, (name, rel), key=lambda t: t[1])
            assert rel <= gate, (name, rel)
    return worst


@pytest.mark.parametrize("H,W", [(96, 64), (37, 53), (65, 33), (128, 200), (8, 8), (1, 1), (259, 131), (300, 517)])
def test_split_products_match_the_oracle_at_the_fp32_gates(H, W):
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from oracle import aru_oracle
    cfg, w, graph = _setup()
    img = _image(H, W, H * 1000 + W)
    ref, inter = aru_oracle.forward_torch(img, w, cfg, return_intermediates=True)
    out = helper.get_net_output(img, graph, "0")
    assert out.shape == ref.shape and out.dtype == np.float32
    _check_endpoints(helper, graph, inter)
    assert float(np.abs(out - ref).max()) <= PROB_TOL
    graph.close()


@pytest.mark.parametrize("H,W", [(250, 333), (515, 260)])
def test_split_products_agree_with_the_plain_fp32_path(H, W):
    """unit logit scale, every end point and the logits: the two fp32 paths of the engine differ by the order of their sums and by the
    split's dropped terms (<= 2^-23 of a product) only"""
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from citlab_article_separation_new_amd.config import AruConfig
    cfg, w, gs = _setup({"apply_softmax": False}, logit_scale=1.0)
    gf = helper.AruGraph(w, AruConfig(apply_softmax=False, compute_dtype="f32"))       # the plain fp32 MFMA / Winograd kernels
    img = _image(H, W, 77)
    ls, lf = helper.get_net_output(img, gs, "0"), helper.get_net_output(img, gf, "0")
    from oracle import aru_oracle
    _, inter = aru_oracle.forward_torch(img, w, cfg, return_intermediates=True)
    worst = ("", 0.0)
    for name in sorted(inter):
        if name.startswith("scale_") or name.startswith("att_"):
            a, b = helper.get_endpoint(gs, name), helper.get_endpoint(gf, name)
            rel = float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max()))
            worst = max(worst, (name, rel), key=lambda t: t[1])
    lrel = float(np.abs(ls - lf).max()) / max(1.0, float(np.abs(lf).max()))
    print(f"\nf32s vs f32 engine {H}x{W}: worst end point {worst[0]} {worst[1]:.2e}; logits {lrel:.2e} (max|l| {np.abs(lf).max():.1f})")
    assert worst[1] <= AGREE_GATE and lrel <= AGREE_GATE
    assert not np.array_equal(ls, lf)                      # the split kernels really ran
    gs.close(); gf.close()


def test_split_products_at_tile_boundaries_against_the_plain_path():
    """sizes around the kernels' tile edges (8 / 16 rows, 32 / 26 columns at every pyramid level) and ragged ones: f32s against the engine's plain
    fp32 path on logits of a unit-logit-scale net (no oracle run: 16 sizes in a few seconds)"""
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from citlab_article_separation_new_amd.config import AruConfig
    cfg, w, gs = _setup({"apply_softmax": False}, logit_scale=1.0)
    gf = helper.AruGraph(w, AruConfig(apply_softmax=False, compute_dtype="f32"))       # the plain fp32 MFMA / Winograd kernels
    rng = np.random.default_rng(123)
    sizes = [(128, 128), (129, 127), (256, 512), (257, 513), (255, 511), (16, 32), (17, 33), (15, 31), (64, 26), (63, 27), (8, 300), (300, 8)]
    sizes += [(int(rng.integers(1, 400)), int(rng.integers(1, 400))) for _ in range(4)]
    worst = (None, 0.0)
    for H, W in sizes:
        img = _image(H, W, H * 7 + W)
        ls, lf = helper.get_net_output(img, gs, "0"), helper.get_net_output(img, gf, "0")
        rel = float(np.abs(ls - lf).max()) / max(1.0, float(np.abs(lf).max()))
        worst = max(worst, ((H, W), rel), key=lambda t: t[1])
        assert rel <= AGREE_GATE, ((H, W), rel)
    print(f"\nf32s vs f32 engine over {len(sizes)} sizes: worst {worst[0]} {worst[1]:.2e}")
    gs.close(); gf.close()


@pytest.mark.parametrize("kw", [
    {"feat_root": 16}, {"res_depth": 2}, {"scale_space_num": 3}, {"scale_space_num": 6, "num_scales_att": 5}, {"n_classes": 3},
    {"scale_space_num": 1, "graph": "RU"}, {"activation_name": "elu"}, {"graph": "U", "activation_name": "leaky"},
], ids=lambda kw: ",".join(f"{k}={v}" for k, v in kw.items()))
def test_split_products_with_other_hyper_parameters_and_graph_variants(kw):
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from oracle import aru_oracle
    cfg, w, graph = _setup(kw, seed=3)
    img = _image(150, 131, 9)
    out = helper.get_net_output(img, graph, "0")
    ref = aru_oracle.forward_torch(img, w, cfg)
    assert out.shape == (150, 131, cfg.n_classes)
    assert np.abs(out - ref).max() <= PROB_TOL
    graph.close()


def test_batched_pages_equal_single_pages():
    import ctypes as C
    import torch
    from citlab_article_separation_new_amd import _lib, net_post_processing_helper as helper
    cfg, w, graph = _setup()
    lib = _lib.init_device(0)
    h = graph.handle(0)
    H, W, B = 75, 131, 5
    rng = np.random.default_rng(0)
    pages = [rng.random((H, W), dtype=np.float32) for _ in range(B)]
    d_in = [torch.from_numpy(p).cuda() for p in pages]
    d_out = [torch.empty(H, W, 2, device="cuda") for _ in range(B)]
    Arr = C.c_void_p * B
    _lib.check(lib.asep_aru_forward_batch_dev(h, B, Arr(*[t.data_ptr() for t in d_in]), H, W, Arr(*[t.data_ptr() for t in d_out]),
                                              None, None, 0.05, None), "asep_aru_forward_batch_dev")
    torch.cuda.synchronize()
    for b in range(B):
        assert np.array_equal(d_out[b].cpu().numpy(), helper.get_net_output(pages[b], graph, "0")), b
    graph.close()


def test_f32s_is_the_default_arithmetic_of_an_fp32_model():
    """AruConfig() selects compute_dtype "f32s" since round 5 (the fp32 engine's default); "f32" selects the plain fp32 kernels, and the two
    really are different kernels (outputs within the agreement gate, not identical)"""
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from citlab_article_separation_new_amd.config import AruConfig
    from citlab_article_separation_new_amd.weights import init_aru_weights
    assert AruConfig().compute_dtype == "f32s"
    w = init_aru_weights(AruConfig(), 5, bias_jitter=0.05, logit_scale=1.0)
    gd, gs, gf = (helper.AruGraph(w, AruConfig(apply_softmax=False)), helper.AruGraph(w, AruConfig(apply_softmax=False, compute_dtype="f32s")),
                  helper.AruGraph(w, AruConfig(apply_softmax=False, compute_dtype="f32")))
    img = _image(150, 200, 3)
    ld, ls, lf = (helper.get_net_output(img, g, "0") for g in (gd, gs, gf))
    assert np.array_equal(ld, ls) and not np.array_equal(ls, lf)
    assert float(np.abs(ls - lf).max()) / max(1.0, float(np.abs(lf).max())) <= AGREE_GATE
    for g in (gd, gs, gf):
        g.close()
