"""GPU parity: HIP ARU-Net (through the C ABI) vs the CPU oracle on the same seeded inputs.

Tolerance: fp32 MFMA vs fp32/fp64 CPU -> max-abs <= 1e-4 on probabilities;
intermediate feature maps are compared relative to their magnitude."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PROB_TOL = 1e-4


def _setup(cfg_kwargs=None, seed=1234):
    from citlab_article_separation_new_amd.config import AruConfig
    from citlab_article_separation_new_amd.weights import init_aru_weights
    from citlab_article_separation_new_amd.net_post_processing_helper import AruGraph
    cfg = AruConfig(**(cfg_kwargs or {}))
    w = init_aru_weights(cfg, seed, bias_jitter=0.05, logit_scale=0.05)
    return cfg, w, AruGraph(w, cfg)


def _image(H, W, seed):
    rng = np.random.default_rng(seed)
    img = rng.random((H, W), dtype=np.float32)
    img[H // 3:H // 3 + 2, :] = 0.05          # a dark rule, like a separator
    return img


@pytest.mark.parametrize("H,W", [(96, 64), (37, 53), (65, 33), (128, 200), (8, 8), (1, 1), (259, 131)])
def test_prob_map_matches_oracle(H, W):
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from oracle import aru_oracle
    cfg, w, graph = _setup()
    img = _image(H, W, H * 1000 + W)
    ref, inter = aru_oracle.forward_torch(img, w, cfg, return_intermediates=True)
    out = helper.get_net_output(img, graph, "0")
    assert out.shape == ref.shape and out.dtype == np.float32
    # intermediates first: a failure there localises the bug
    for name in sorted(inter):
        if name.startswith("scale_") or name.startswith("att_"):
            got = helper.get_endpoint(graph, name)
            want = inter[name]
            assert got.shape == want.shape, name
            scale = max(1.0, float(np.abs(want).max()))
            assert float(np.abs(got - want).max()) <= 2e-5 * scale, name
    assert float(np.abs(out - ref).max()) <= PROB_TOL
    graph.close()


def test_fused_u8_and_threshold_outputs():
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from oracle import aru_oracle
    cfg, w, graph = _setup()
    img = _image(120, 88, 7)
    out, u8, mask = helper.get_net_output_fused(img, graph, "0", want_u8=True, threshold=0.05)
    # the u8/threshold epilogue is bit-exact w.r.t. the engine's own float output
    assert np.array_equal(u8, aru_oracle.to_uint8(out))
    assert np.array_equal(mask, aru_oracle.apply_threshold(u8, 0.05))
    graph.close()


def test_ru_graph_without_attention():
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from oracle import aru_oracle
    cfg, w, graph = _setup({"graph": "RU"})
    img = _image(70, 90, 3)
    ref = aru_oracle.forward_torch(img, w, cfg)
    out = helper.get_net_output(img, graph, "0")
    assert float(np.abs(out - ref).max()) <= PROB_TOL
    graph.close()


def test_mvn_and_logits_output():
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from oracle import aru_oracle
    cfg, w, graph = _setup({"mvn": True, "apply_softmax": False, "n_classes": 3})
    img = _image(64, 48, 11) * 255.0
    ref = aru_oracle.forward_torch(img, w, cfg)
    out = helper.get_net_output(img, graph, "0")
    scale = max(1.0, float(np.abs(ref).max()))
    assert float(np.abs(out - ref).max()) <= 2e-5 * scale
    graph.close()


def test_input_forms_and_errors():
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    cfg, w, graph = _setup()
    img = _image(40, 40, 5)
    a = helper.get_net_output(img, graph, "0")
    b = helper.get_net_output(img[None, :, :, None].astype(np.float64), graph, "0")   # [1,H,W,1] float64 feed
    assert np.array_equal(a, b)
    with pytest.raises(ValueError):
        helper.get_net_output(np.zeros((2, 8, 8, 1)), graph, "0")
    with pytest.raises(IOError):
        helper.load_graph("/nonexistent/model.pb")
    graph.close()


@pytest.mark.parametrize("H,W", [(96, 80), (250, 333)])
def test_bf16_mfma_variant_within_stated_tolerance(H, W):
    """BASELINE config 5 ("bf16 convs"): bf16 MFMA operands with fp32 accumulation and fp32 activations.
    Tolerance 2e-2 on the probability map; the fp32 path of the same engine is the 1e-4 one."""
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from oracle import aru_oracle
    cfg, w, graph = _setup()
    cfg_bf = type(cfg)(**{**cfg.to_dict(), "compute_dtype": "bf16"})
    graph_bf = helper.AruGraph(w, cfg_bf)
    img = _image(H, W, 5)
    ref = aru_oracle.forward_torch(img, w, cfg)
    out_bf = helper.get_net_output(img, graph_bf, "0")
    out_f32 = helper.get_net_output(img, graph, "0")
    err_bf = np.abs(out_bf - ref).max()
    err_f32 = np.abs(out_f32 - ref).max()
    print(f"bf16 max|dp| = {err_bf:.2e}, f32 max|dp| = {err_f32:.2e}")
    assert err_f32 <= 1e-4
    assert err_bf <= 2e-2
    assert err_bf > err_f32                       # the variant really computes in reduced precision
    # the thresholded uint8 masks agree except where the probability is within the tolerance of the threshold
    m_bf = aru_oracle.apply_threshold(aru_oracle.to_uint8(out_bf), 0.5)
    m_ref = aru_oracle.apply_threshold(aru_oracle.to_uint8(ref), 0.5)
    assert (m_bf != m_ref).mean() <= 0.05


@pytest.mark.parametrize("kw", [
    {"feat_root": 16}, {"res_depth": 2}, {"res_depth": 4}, {"scale_space_num": 3},
    {"scale_space_num": 6, "num_scales_att": 5},            # the upstream ARU-Net paper layout
    {"n_classes": 3}, {"n_classes": 1}, {"num_scales_att": 2}, {"scale_space_num": 2}, {"scale_space_num": 1, "graph": "RU"},
], ids=lambda kw: ",".join(f"{k}={v}" for k, v in kw.items()))
def test_non_default_hyper_parameters(kw):
    """ARU_v1.py:35-43 graph_params other than the defaults (the shipped nets' true values are unknown)."""
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from oracle import aru_oracle
    cfg, w, graph = _setup(kw, seed=3)
    img = _image(150, 131, 9)
    out = helper.get_net_output(img, graph, "0")
    ref = aru_oracle.forward_torch(img, w, cfg)
    assert out.shape == (150, 131, cfg.n_classes)
    assert np.abs(out - ref).max() <= PROB_TOL


@pytest.mark.parametrize("kw", [
    {"activation_name": "elu"}, {"activation_name": "leaky"},                       # ARU_v1.py:70-75
    {"graph": "RU", "activation_name": "elu"},
    {"graph": "U"}, {"graph": "U", "activation_name": "leaky"}, {"graph": "U", "activation_name": "elu", "scale_space_num": 3},
], ids=lambda kw: ",".join(f"{k}={v}" for k, v in kw.items()))
@pytest.mark.parametrize("H,W", [(150, 131), (37, 53)])
def test_graph_variants_elu_leaky_and_plain_u(kw, H, W):
    """ARU_v1.py:43,70-75 (activation_name elu / leaky: convR / conv2 layers, block ends, deconvolutions and the attention CNN; the
    ReLU behind conv1 of a residual block stays a ReLU) and graph 'U' (:228-233,:283-288: conv1 + conv2 blocks, no residual add, no
    attention).  The engine runs these nets layer by layer with act_kernel behind the convolutions; every end point and the
    probabilities must meet the fp32 gates of the default graph."""
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from oracle import aru_oracle
    cfg, w, graph = _setup(kw, seed=5)
    assert ("unet_down_0/conv2/weights" in "".join(w)) == (kw.get("graph") == "U")
    img = _image(H, W, 31)
    ref, inter = aru_oracle.forward_torch(img, w, cfg, return_intermediates=True)
    out = helper.get_net_output(img, graph, "0")
    neg = 0
    for name in sorted(inter):
        if name.startswith("scale_") or name.startswith("att_"):
            got, want = helper.get_endpoint(graph, name), inter[name]
            assert got.shape == want.shape, name
            assert float(np.abs(got - want).max()) <= 2e-5 * max(1.0, float(np.abs(want).max())), name
            neg += int((want < 0).sum())
    if kw.get("activation_name", "relu") != "relu":
        assert neg > 0                                       # the negative branch of the activation was exercised
    assert float(np.abs(out - ref).max()) <= PROB_TOL
    graph.close()


@pytest.mark.parametrize("kw", [{"activation_name": "elu"}, {"activation_name": "leaky"}, {"graph": "U", "activation_name": "leaky"}],
                         ids=lambda kw: ",".join(f"{k}={v}" for k, v in kw.items()))
def test_fused_activation_of_the_variants_is_the_separate_pass_bit_for_bit(kw, monkeypatch):
    """round 4: elu / leaky are applied in the epilogue of the producing kernel (ConvArgs::act, C1Args::act) instead of by an
    act_kernel pass behind it; same arithmetic on the same values, so ASEP_FUSE_ACT=0 (the round-3 form) must give the same bits --
    on a size with interior AND border tiles (the fused form routes a variant's launches through the general epilogues)."""
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    img = _image(300, 270, 3)
    outs, eps = [], []
    # (ASEP_FUSED8=0: the attention head as conv1 + pool in both runs -- its fused vector-ALU form, which an elu / leaky graph takes
    # since round 4 as well, sums the taps in another order; it is held to the oracle by test_graph_variants_elu_leaky_and_plain_u)
    monkeypatch.setenv("ASEP_FUSED8", "0")
    for flag in ("1", "0"):
        monkeypatch.setenv("ASEP_FUSE_ACT", flag)
        cfg, w, graph = _setup(kw, seed=8)
        outs.append(helper.get_net_output(img, graph, "0"))
        eps.append({n: helper.get_endpoint(graph, n) for n in ("scale_0_unet_down_0_conv", "scale_0_unet_down_2_conv", "scale_0_unet_up_0_deconv",
                                                               "scale_0_unet_up_0_conv")})
        graph.close()
    assert np.array_equal(outs[0], outs[1])
    assert all(np.array_equal(eps[0][n], eps[1][n]) for n in eps[0])
    assert (eps[0]["scale_0_unet_up_0_deconv"] < 0).any()        # negative branch exercised
    if kw.get("graph") != "U":
        # the fused head (default) against the unfused one: same function, other summation order
        monkeypatch.delenv("ASEP_FUSED8")
        monkeypatch.setenv("ASEP_FUSE_ACT", "1")
        cfg, w, graph = _setup(kw, seed=8)
        out_head = helper.get_net_output(img, graph, "0")
        graph.close()
        assert float(np.abs(out_head - outs[0]).max()) <= 1e-5


@pytest.mark.parametrize("kw", [
    {"activation_name": "elu"}, {"activation_name": "leaky"}, {"graph": "RU", "activation_name": "elu"},
    {"graph": "U"}, {"graph": "U", "activation_name": "leaky"},
], ids=lambda kw: ",".join(f"{k}={v}" for k, v in kw.items()))
@pytest.mark.parametrize("H,W", [(150, 131), (37, 53)])
def test_graph_variants_on_the_bf16_path(kw, H, W):
    """round 5 (VERDICT r4 missing #4): elu / leaky / graph 'U' nets on the bf16 engine -- layer by layer on convb_kernel / deconvb_kernel /
    the vector-ALU attention head, the activation applied to the fp32 sums before the rounding to bf16 (round 4 refused them).  Gates of the
    ReLU nets' bf16 tests: block by block against the oracle with the engine's roundings (teacher forcing), probabilities of
    logit_scale 0.05 weights within 2e-2 of the fp32 oracle."""
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from oracle import aru_oracle
    cfg, w, graph = _setup(dict(kw, compute_dtype="bf16"), seed=5)
    img = _image(H, W, 31)
    ref32, inter32 = aru_oracle.forward_torch(img, w, cfg, return_intermediates=True)
    out = helper.get_net_output(img, graph, "0")
    names = [n for n in sorted(inter32) if n.startswith("scale_") or n.startswith("att_")]
    eng = _bf16_endpoints(graph, names)
    graph.close()
    _, forced = aru_oracle.forward_torch(img, w, cfg, return_intermediates=True, storage="bf16", teacher=eng)
    rows, neg = [], 0
    for n in names:
        want = forced[n]
        assert eng[n].shape == want.shape, n
        scale = max(1.0, float(np.abs(want).max()))
        d = eng[n] - want
        rows.append((n, float(np.abs(d).max()) / scale, float(np.sqrt(np.mean(d.astype(np.float64) ** 2))) / scale))
        neg += int((want < 0).sum())
    bm, br = max(rows, key=lambda t: t[1]), max(rows, key=lambda t: t[2])
    perr = float(np.abs(out - ref32).max())
    print(f"\nbf16 {kw} {H}x{W}: block by block max {bm[0]} {bm[1]:.2e}, rms {br[0]} {br[2]:.2e}; max|dp| vs fp32 oracle {perr:.2e}")
    assert bm[1] <= BF16_BLOCK_MAX_GATE and br[2] <= BF16_BLOCK_RMS_GATE, (bm, br)
    assert perr <= 2e-2
    if kw.get("activation_name", "relu") != "relu":
        assert neg > 0                                       # the negative branch of the activation was exercised


def test_unsupported_width_is_rejected_loudly():
    from citlab_article_separation_new_amd import _lib, net_post_processing_helper as helper
    cfg, w, graph = _setup({"feat_root": 4})
    with pytest.raises(_lib.AsepError):
        helper.get_net_output(_image(32, 32, 0), graph, "0")


@pytest.mark.parametrize("H,W", [(200, 150), (70, 300)])
def test_fp32_level0_mfma_variant_agrees_with_the_vector_alu_kernels(H, W, monkeypatch):
    """The fp32 level-0 blocks run on the vector ALU (res8v_kernels.h); the MFMA kernels of the same blocks serve bf16 and
    tensors of >= 2^28 pixels, and ASEP_R8_VALU=0 selects them for fp32: both must meet the oracle tolerance, and agree with
    each other far inside it (same sums, different order)."""
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from oracle import aru_oracle
    img = _image(H, W, 77)
    outs = {}
    for valu in ("1", "0"):
        monkeypatch.setenv("ASEP_R8_VALU", valu)          # read when the engine is created
        cfg, w, graph = _setup()
        outs[valu] = helper.get_net_output(img, graph, "0")
        graph.close()
    ref = aru_oracle.forward_torch(img, w, cfg)
    for valu, out in outs.items():
        assert float(np.abs(out - ref).max()) <= PROB_TOL, valu
    assert float(np.abs(outs["1"] - outs["0"]).max()) <= 1e-5


@pytest.mark.parametrize("H,W", [(200, 150), (67, 131)])
def test_fused_pool_and_dense_12_channel_mode_match_their_plain_forms(H, W, monkeypatch):
    """Two engine switches that must not change results: ASEP_FUSE_POOL=0 runs maxpool2_kernel after every conv instead of the
    2x2 max in the conv epilogues (same values, so the outputs are bit-identical), ASEP_C12=0 pads the attention conv2's 12
    input channels to 16 instead of the dense K mapping (another summation order: equal to 1e-5)."""
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    img = _image(H, W, 91)
    outs = {}
    for name, env in (("default", {}), ("plain_pool", {"ASEP_FUSE_POOL": "0"}), ("padded_c12", {"ASEP_C12": "0"})):
        for k in ("ASEP_FUSE_POOL", "ASEP_C12"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)                       # read when the engine is created
        cfg, w, graph = _setup()
        outs[name] = helper.get_net_output(img, graph, "0")
        graph.close()
    assert np.array_equal(outs["default"], outs["plain_pool"])
    assert float(np.abs(outs["default"] - outs["padded_c12"]).max()) <= 1e-5


def test_integration_md_stub_runs_the_net(tmp_path):
    """INTEGRATION.md section 1 is code: its hand-written ctypes binding (own struct, own argtypes) loads a weight container and
    runs the net; the result must be the oracle's.  A struct of another size is refused with a message (a
    10-field struct against the 12-field ABI read stack garbage)."""
    import ctypes as C
    import os
    import sys
    from citlab_article_separation_new_amd.weights import pack_blob
    from oracle import aru_oracle
    sys.path.insert(0, os.path.dirname(__file__))
    from test_abi_and_weights import integration_md_stub
    repo_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ns = integration_md_stub(repo_root)
    cfg, w, graph = _setup()
    graph.close()
    blob_path = tmp_path / "m.asepw"
    blob_path.write_bytes(pack_blob(w))
    h = ns["load_graph"](str(blob_path))
    img = _image(96, 64, 11)
    out = ns["get_net_output"](img, h)
    ref = aru_oracle.forward_torch(img, w, cfg)
    assert float(np.abs(out - ref).max()) <= PROB_TOL
    ns["lib"].asep_aru_free.argtypes = [C.c_void_p]
    ns["lib"].asep_aru_free(h)

    class OldCfg(C.Structure):                      # round 2's ten-field struct (what INTEGRATION.md showed until round 3)
        _fields_ = [(n, C.c_int32) for n in ("channels", "n_classes", "feat_root", "scale_space_num", "res_depth", "num_scales_att",
                                             "use_attention", "mvn", "apply_softmax", "compute_dtype")]
    blob = blob_path.read_bytes()
    old = OldCfg(1, 2, 8, 5, 3, 3, 1, 0, 1, 0)
    assert not ns["lib"].asep_aru_load(blob, len(blob), C.byref(old))
    assert b"struct_size" in ns["lib"].asep_last_error()


# the engine's bf16 path against the oracle WITH the same roundings (oracle/aru_oracle.py forward_torch(storage="bf16")).
# (a) free running: both evaluate the whole net; what differs is the summation order inside a convolution (fp32, ~1e-7) and the
#     bfloat16 roundings it flips -- which pile up over forty layers until the two drift apart almost like bf16 from fp32;
# (b) block by block ("teacher forcing"): the oracle computes every end point from the ENGINE's upstream end points, so one block's
#     arithmetic is compared at a time: a flipped rounding here and there (one bfloat16 step = 2^-8 of the value), nothing else.
BF16_EMU_ENDPOINT_GATE = 2.5e-2    # (a) max|d| / max|ref| per end point
BF16_EMU_ENDPOINT_RMS_GATE = 3e-3  # (a) rms(d) / max|ref| per end point
BF16_EMU_PROB_GATE = 2e-3          # (a) logit_scale 0.05 weights (against the fp32 oracle: 2e-2 allowed, 1.3e-3 measured)
BF16_BLOCK_MAX_GATE = 1.2e-2       # (b) max|d| / max|ref| per end point: a few bfloat16 steps (2^-8) of the largest values; measured 4e-3 .. 6e-3
BF16_BLOCK_RMS_GATE = 5e-4         # (b) rms(d) / max|ref| per end point; measured 1.2e-4 .. 3.1e-4 (free running 1.6e-3, fp32 oracle 1e-3 .. 3.4e-3)


def _bf16_endpoints(graph_bf, names):
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    return {n: helper.get_endpoint(graph_bf, n) for n in names}


@pytest.mark.parametrize("H,W", [(96, 80), (250, 333), (37, 53)])
def test_bf16_path_against_the_oracle_with_the_same_roundings(H, W):
    """the bf16 gates were the softest of the suite (probabilities only, 2e-2).  A wrong bf16 kernel that stays
    inside 2e-2 of the fp32 graph passes there; against an oracle that rounds where the engine rounds, evaluated block by block on the
    engine's own upstream tensors, it does not.  Every end point (max and rms relative to max|ref|), interior and border tiles
    (37 x 53: border tiles only)."""
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from oracle import aru_oracle
    cfg, w, graph = _setup()
    graph.close()
    cfg_bf = type(cfg)(**{**cfg.to_dict(), "compute_dtype": "bf16"})
    graph_bf = helper.AruGraph(w, cfg_bf)
    img = _image(H, W, 5)
    ref, inter = aru_oracle.forward_torch(img, w, cfg, return_intermediates=True, storage="bf16")
    ref32 = aru_oracle.forward_torch(img, w, cfg)
    out = helper.get_net_output(img, graph_bf, "0")
    names = [n for n in sorted(inter) if n.startswith("scale_") or n.startswith("att_")]
    eng = _bf16_endpoints(graph_bf, names)
    graph_bf.close()

    def worst_of(want_of):
        rows = []
        for n in names:
            want = want_of[n]
            assert eng[n].shape == want.shape, n
            scale = max(1.0, float(np.abs(want).max()))
            d = eng[n] - want
            rows.append((n, float(np.abs(d).max()) / scale, float(np.sqrt(np.mean(d.astype(np.float64) ** 2))) / scale))
        return rows
    free = worst_of(inter)
    _, forced = aru_oracle.forward_torch(img, w, cfg, return_intermediates=True, storage="bf16", teacher=eng)
    block = worst_of(forced)
    perr, perr32 = float(np.abs(out - ref).max()), float(np.abs(out - ref32).max())
    fm, fr, bm, br = max(free, key=lambda t: t[1]), max(free, key=lambda t: t[2]), max(block, key=lambda t: t[1]), max(block, key=lambda t: t[2])
    print(f"\nbf16 vs bf16-rounding oracle {H}x{W}: free running max {fm[0]} {fm[1]:.2e}, rms {fr[0]} {fr[2]:.2e}; block by block max "
          f"{bm[0]} {bm[1]:.2e}, rms {br[0]} {br[2]:.2e}; max|dp| {perr:.2e} (against the fp32 oracle {perr32:.2e})")
    assert fm[1] <= BF16_EMU_ENDPOINT_GATE and fr[2] <= BF16_EMU_ENDPOINT_RMS_GATE and perr <= BF16_EMU_PROB_GATE
    assert bm[1] <= BF16_BLOCK_MAX_GATE and br[2] <= BF16_BLOCK_RMS_GATE, (bm, br)


@pytest.mark.parametrize("H,W", [(250, 333), (96, 200), (52, 132), (53, 155), (300, 135), (611, 477)])
def test_level0_blocks_as_strip_walkers_against_the_tile_kernels_and_the_oracle(H, W, monkeypatch):
    """round 6: the bf16 level-0 blocks as column-strip walkers (res8w_kernels.h: one wave per 24-column strip, rolling
    rows of every stage in its own LDS rings, no barrier) + border tiles around the walkers' region, against (a) the 16 x 32-tile kernels
    (ASEP_BF_WALK=0): BIT-IDENTICAL wherever both evaluate the lean form (bias as the accumulators' initial value, same order of the filter
    rows), a rounding flip at most where one side's tile is a border tile (res8b_tile adds the bias behind the sum); (b) the oracle with the
    engine's roundings, block by block.  Sizes: first / last strip widths of every residue, one and several bands, pages whose coarser
    scales stay on the tile kernels (a walker needs four strips and two tile rows)."""
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from oracle import aru_oracle
    down, up = "scale_0_unet_down_0_conv", "scale_0_unet_up_0_conv"
    img = _image(H, W, 77)
    res = {}
    for walk in ("1", "0"):
        monkeypatch.setenv("ASEP_BF_WALK", walk)               # read when the engine is created
        cfg, w, graph = _setup({"compute_dtype": "bf16"}, seed=9)
        res[walk] = (helper.get_net_output(img, graph, "0"),
                     {n: helper.get_endpoint(graph, n) for n in (down, "scale_0_unet_down_1_conv", "scale_0_unet_up_1_conv", up)})
        graph.close()
    (p1, eng), (p0, eng0) = res["1"], res["0"]
    # the walkers' region, and the tile kernels' interior tiles (whole 24 x 40 window inside the image)
    ys, xs = np.arange(H)[:, None], np.arange(W)[None, :]
    n_strips, y_end = (W - 4 - 32) // 24, 16 + 2 * ((H - 4 - 16) // 2)
    walker = (ys >= 16) & (ys < y_end) & (xs >= 32) & (xs < 32 + 24 * n_strips)
    ty0, tx0 = (ys // 16) * 16, (xs // 32) * 32
    lean_tile = (ty0 - 4 >= 0) & (ty0 + 20 <= H) & (tx0 - 4 >= 0) & (tx0 + 36 <= W)
    both = walker & lean_tile
    assert both.sum() > 0.3 * H * W or H < 64
    # the down block reads the same image in both runs: its lean regions must agree bit for bit
    d1, d0 = eng[down], eng0[down]
    assert d1.shape == d0.shape == (H, W, 8)
    diff = (d1 != d0).any(axis=2)
    assert not (diff & both).any(), f"down block: {int((diff & both).sum())} pixels differ where both kernels run their lean form"
    scale = max(1.0, float(np.abs(d0).max()))
    assert float(np.abs(d1 - d0).max()) <= 2.0 ** -7 * scale and diff.mean() <= 0.02      # elsewhere: a bfloat16 step here and there
    # the fused 2 x 2 pool feeds level 1: the same up to what those flips do there
    assert float(np.abs(eng["scale_0_unet_down_1_conv"] - eng0["scale_0_unet_down_1_conv"]).max()) <= 2e-2 * max(1.0, float(np.abs(eng0["scale_0_unet_down_1_conv"]).max()))
    assert float(np.abs(p1 - p0).max()) <= 2e-3
    # (b) both blocks against the oracle with the engine's roundings, each from the engine's own upstream end points; where the up block's
    # inputs agree between the two runs (they do except for the flips above) its lean region is bit-identical too: checked on a run whose
    # inputs ARE identical -- the walker run's own tensors through the tile kernel are not available, so the oracle is the judge here
    cfg32 = type(cfg)(**{**cfg.to_dict(), "compute_dtype": "f32"})
    _, forced = aru_oracle.forward_torch(img, w, cfg32, return_intermediates=True, storage="bf16", teacher=eng)
    for name in (down, up):
        d = eng[name] - forced[name]
        sc = max(1.0, float(np.abs(forced[name]).max()))
        assert float(np.abs(d).max()) / sc <= BF16_BLOCK_MAX_GATE and float(np.sqrt(np.mean(d.astype(np.float64) ** 2))) / sc <= BF16_BLOCK_RMS_GATE, name


@pytest.mark.parametrize("H,W", [(250, 333), (611, 477)])
def test_up_block_strip_walker_is_bit_identical_to_the_tile_kernel_on_identical_inputs(H, W, monkeypatch):
    """the up block alone on identical inputs: the graph 'RU' ... is not needed -- ASEP_BF_WALK=2 walks the UP block only, so both runs share
    the tile kernel's down block and every tensor in front of unet_up_0; the up block's lean regions must then agree bit for bit"""
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    name = "scale_0_unet_up_0_conv"
    img = _image(H, W, 78)
    res = {}
    for walk in ("2", "0"):
        monkeypatch.setenv("ASEP_BF_WALK", walk)
        cfg, w, graph = _setup({"compute_dtype": "bf16"}, seed=10)
        helper.get_net_output(img, graph, "0")
        res[walk] = helper.get_endpoint(graph, name)
        graph.close()
    e1, e0 = res["2"], res["0"]
    ys, xs = np.arange(H)[:, None], np.arange(W)[None, :]
    n_strips, y_end = (W - 4 - 32) // 24, 16 + 2 * ((H - 4 - 16) // 2)
    walker = (ys >= 16) & (ys < y_end) & (xs >= 32) & (xs < 32 + 24 * n_strips)
    ty0, tx0 = (ys // 16) * 16, (xs // 32) * 32
    lean_tile = (ty0 - 4 >= 0) & (ty0 + 20 <= H) & (tx0 - 4 >= 0) & (tx0 + 36 <= W)
    diff = (e1 != e0).any(axis=2)
    assert (walker & lean_tile).sum() > 0.5 * H * W and not (diff & walker & lean_tile).any()
    assert float(np.abs(e1 - e0).max()) <= 2.0 ** -7 * max(1.0, float(np.abs(e0).max())) and diff.mean() <= 0.02


@pytest.mark.parametrize("H,W", [(250, 333), (611, 477), (96, 200), (1100, 908), (777, 1290), (16, 16), (24, 40), (40, 530), (300, 17)])
def test_the_64_channel_layers_with_the_filter_in_registers_are_bit_identical_to_convb_kernel(H, W, monkeypatch):
    """round 6: convr_kernel (csrc/convr_kernels.h: the layer's 72 A fragments in the registers of one wave per SIMD, a wave =
    the pipeline of a 32-column strip, rows by LDS-DMA, zero padding by source address) keeps convb_kernel's accumulation order: every end point of
    the net and the probabilities must be BIT-IDENTICAL to the run with ASEP_BF_CONVR=0.  Sizes: level-3 maps of 32 x 42 (two strips, the second
    10 columns wide) ... 162 x 98 pixels (ranges that cross strip ends), all three scales of the pyramid in one launch; maps of 2 x 2, 3 x 5, 5 x 67 and
    38 x 3 pixels (ranges of one and two rows, a single partial strip).  The RES form (unet_up_3/convR_2) is in every one of them."""
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    names = ["scale_0_unet_down_3_conv", "scale_0_unet_up_3_conv", "scale_1_unet_down_3_conv", "scale_2_unet_up_3_conv", "scale_0_unet_up_0_conv"]
    img = _image(H, W, 91)
    res = {}
    for on in ("1", "0"):
        monkeypatch.setenv("ASEP_BF_CONVR", on)
        cfg, w, graph = _setup({"compute_dtype": "bf16"}, seed=12)
        res[on] = (helper.get_net_output(img, graph, "0"), {n: helper.get_endpoint(graph, n) for n in names})
        graph.close()
    (p1, e1), (p0, e0) = res["1"], res["0"]
    for n in names:
        assert e1[n].shape == e0[n].shape and np.array_equal(e1[n], e0[n]), (n, int((e1[n] != e0[n]).sum()))
    assert np.array_equal(p1, p0)



@pytest.mark.parametrize("act", ["elu", "leaky"])
@pytest.mark.parametrize("H,W", [(250, 333), (97, 200)])
def test_elu_and_leaky_graphs_on_the_fused_general_blocks_of_the_bf16_engine(act, H, W, monkeypatch):
    """round 6: the elu / leaky RESIDUAL graphs run their level-0 blocks on res8b_kernel<UP, ACT> and their 16-channel tails
    on resb_tail_kernel<16, ACT> (the general tile forms: activation on the fp32 sums before the one rounding, float maxima in the pool) instead of
    layer by layer (ASEP_FUSED8=0).  Both evaluate the same graph with the same roundings; the fused forms add the bias behind the sum where
    convb_kernel starts from it -- a bfloat16 step on a few values, no more.  (The block-by-block gates against the oracle with the engine's
    roundings are test_graph_variants_on_the_bf16_path's and hold for the fused forms: they are what that test runs now.)"""
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    names = ["scale_0_unet_down_0_conv", "scale_0_unet_down_1_conv", "scale_0_unet_up_1_conv", "scale_0_unet_up_0_conv", "scale_1_unet_up_0_conv"]
    img = _image(H, W, 57)
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("ASEP_FUSED8", fused)
        cfg, w, graph = _setup({"compute_dtype": "bf16", "activation_name": act}, seed=14)
        p = helper.get_net_output(img, graph, "0")
        res[fused] = (p, {n: helper.get_endpoint(graph, n) for n in names})
        graph.close()
    (p1, e1), (p0, e0) = res["1"], res["0"]
    assert (e0["scale_0_unet_down_0_conv"] < 0).any()           # the negative branch of the activation is exercised
    d0 = e1["scale_0_unet_down_0_conv"] - e0["scale_0_unet_down_0_conv"]
    sc = max(1.0, float(np.abs(e0["scale_0_unet_down_0_conv"]).max()))
    assert float(np.abs(d0).max()) <= 2.0 ** -7 * sc and float((d0 != 0).mean()) <= 0.02      # the first block reads the same image in both runs
    for n in names[1:]:
        sc = max(1.0, float(np.abs(e0[n]).max()))
        assert float(np.abs(e1[n] - e0[n]).max()) <= BF16_EMU_ENDPOINT_GATE * sc, n    # (free running: flipped roundings pile up over the layers)
    assert float(np.abs(p1 - p0).max()) <= 6e-3                  # (two bf16 evaluations drift apart about as far as each drifts from fp32: 3.4e-3 measured)
