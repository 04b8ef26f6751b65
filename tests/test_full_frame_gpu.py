"""Whole-frame parity at the sizes BASELINE.json names -- no interior crops.

  * configs[1]/[4]: one whole 3000 x 4500 page against ``oracle.aru_oracle.forward_torch`` on the same page: fp32 MFMA path
    max|dp| <= 1e-4, bf16 MFMA path <= 2e-2, on every pixel including the page's borders at the odd pyramid levels
    (1125, 563, 282 rows; 375, 188 columns), plus the uint8 / threshold-mask mismatch RATE against the oracle
    (a truncation ``uint8(p * 255)`` flips wherever p * 255 sits within the float error of an integer,
    so the rate is reported and bounded rather than required to be zero).
  * configs[0]: the 512 x 768 crop through ``run_net_post_processing --mode separator --fixed_height 768`` (net input =
    the crop itself) against the reference's step sequence evaluated with the oracle for EVERY step (net included).
  * configs[2]: the stroke-width distance transform and the connected-component / opening stage at 3000 x 4500 against
    ``oracle.classical_oracle``, bit for bit.
"""
import numpy as np
import pytest
from PIL import Image

pytestmark = pytest.mark.gpu

import os  # noqa: E402
import sys  # noqa: E402
sys.path.insert(0, os.path.dirname(__file__))
import tf_aru_graph  # noqa: E402

H, W = 4500, 3000


@pytest.fixture(scope="module")
def whole_page():
    """page, weights and the oracle's output for the whole frame (one oracle run serves the fp32 and the bf16 test)"""
    from citlab_article_separation_new_amd import synth
    from citlab_article_separation_new_amd.config import AruConfig
    from citlab_article_separation_new_amd.weights import init_aru_weights
    from oracle import aru_oracle
    cfg = AruConfig()
    w = init_aru_weights(cfg, 1234, bias_jitter=0.05, logit_scale=0.05)
    page = synth.synth_page(0, W, H).astype(np.float32) / 255.0
    ref = aru_oracle.forward_torch(page, w, cfg)
    assert ref.shape == (H, W, 2)
    return page, w, ref


def _report(tag, out, ref, thr):
    from oracle import aru_oracle
    err = np.abs(out - ref)
    u8, u8_ref = aru_oracle.to_uint8(out), aru_oracle.to_uint8(ref)
    u8_rate = float((u8 != u8_ref).mean())
    u8_max = int(np.abs(u8.astype(np.int16) - u8_ref.astype(np.int16)).max())
    m_rate = float((aru_oracle.apply_threshold(u8, thr) != aru_oracle.apply_threshold(u8_ref, thr)).mean())
    border = np.ones((H, W), bool)
    border[64:-64, 64:-64] = False
    print(f"\n{tag} whole frame {W}x{H}: max|dp| = {err.max():.3e} (border band 64 px: {err[border].max():.3e}, "
          f"interior: {err[~border].max():.3e}); uint8 mismatch rate = {u8_rate:.3e} (max step {u8_max}); "
          f"threshold-mask mismatch rate @thr={thr} = {m_rate:.3e}")
    return float(err.max()), u8_rate, u8_max, m_rate


@pytest.mark.parametrize("dtype", ["f32", "f32s"])
def test_whole_page_fp32_vs_oracle(whole_page, dtype):
    """(f32s: fp32 tensors and accumulation, the wide convolutions' products as six bf16 x bf16 partial products of the three-way split of
    both factors -- csrc/split_kernels.h; held to the fp32 gates, not to the bf16 ones)"""
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from citlab_article_separation_new_amd.config import AruConfig
    from oracle import aru_oracle
    page, w, ref = whole_page
    g = helper.AruGraph(w, AruConfig(compute_dtype=dtype))
    thr = round(float(np.median(ref[:, :, 0])), 3)
    out, u8, mask = helper.get_net_output_fused(page, g, "0", want_u8=True, threshold=thr)
    err, u8_rate, u8_max, m_rate = _report(dtype, out, ref, thr)
    assert err <= 1e-4
    assert u8_max <= 1 and u8_rate <= 2e-3 and m_rate <= 2e-3
    # the fused epilogue is exactly uint8(p * 255) / apply_threshold of the engine's own float output
    assert np.array_equal(u8, aru_oracle.to_uint8(out)) and np.array_equal(mask, aru_oracle.apply_threshold(u8, thr))
    g.close()


def test_whole_page_bf16_vs_oracle(whole_page):
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from citlab_article_separation_new_amd.config import AruConfig
    page, w, ref = whole_page
    g = helper.AruGraph(w, AruConfig(compute_dtype="bf16"))
    out = helper.get_net_output(page, g, "0")
    thr = round(float(np.median(ref[:, :, 0])), 3)
    err, u8_rate, u8_max, m_rate = _report("bf16", out, ref, thr)
    assert 1e-6 < err <= 2e-2
    g.close()


@pytest.fixture(scope="module")
def unit_scale_page():
    """page 1, reference-rule weights (logit_scale = 1: saturating class softmax) and the oracle's end points for the whole frame
    (one oracle run serves the fp32 and the bf16 end-point tests)"""
    from citlab_article_separation_new_amd import synth
    from citlab_article_separation_new_amd.config import AruConfig
    from citlab_article_separation_new_amd.weights import init_aru_weights
    from oracle import aru_oracle
    cfg = AruConfig()
    w = init_aru_weights(cfg, 4321, bias_jitter=0.05, logit_scale=1.0)
    page = synth.synth_page(1, W, H).astype(np.float32) / 255.0
    ref, inter = aru_oracle.forward_torch(page, w, cfg, return_intermediates=True)
    # Random weights with unit logit scale saturate (p > 0.999 on 99.9 % of the page): every threshold mask would be trivially
    # equal.  The class bias is shifted by the median logit margin, so that p = 0.5 splits the page in two halves -- the most
    # sensitive mask a threshold can produce.  Only the last layer's bias changes: the oracle's logits move by the same constant and
    # every other end point is untouched, so no second oracle run is needed.
    margin = float(np.median(inter["logits"][:, :, 0] - inter["logits"][:, :, 1]))
    w = dict(w)
    b = w["aru_net/logit/class/biases"].copy()
    b[0] -= np.float32(margin)
    w["aru_net/logit/class/biases"] = b
    inter["logits"] = inter["logits"] - np.array([np.float32(margin), 0], dtype=np.float32)
    z = inter["logits"] - inter["logits"].max(axis=2, keepdims=True)
    e = np.exp(z.astype(np.float32))
    ref = (e / e.sum(axis=2, keepdims=True)).astype(np.float32)
    assert 0.45 < float((ref[:, :, 0] > 0.5).mean()) < 0.55
    return page, w, cfg, ref, inter


# what the bf16 path holds on a whole 3000 x 4500 frame with unit logit scale (measured values in DESIGN section 2; gates ~1.5x above)
BF16_ENDPOINT_GATE = 3e-2        # max|d| / max|ref| per end point (measured 8e-3 .. 2.1e-2; fp32: 2e-5)
BF16_ENDPOINT_RMS_GATE = 5e-3    # rms(d) / max|ref| per end point (measured 1e-3 .. 3.4e-3)
BF16_LOGIT_GATE = 2.5e-2         # max|d logits| / max|logits| (measured 1.4e-2 .. 1.6e-2 at max|logit| 24 .. 31, i.e. ~0.4 in the logit margin)
BF16_PROB_GATE = 0.15            # max|dp| (measured 0.099): a margin error of 0.44 at p = 0.5 is dp = 0.11 (the stated 2e-2 is the gate of
                                 # logit_scale 0.05 weights, whose margins -- and margin errors -- are 20x smaller)
BF16_CONFIDENT_MARGIN = 2.0      # logit margin beyond which a pixel's mask value must be the oracle's (measured: 54 % of this page)
BF16_MASK_RATE_GATE = 1.2e-2     # share of threshold-mask pixels that differ at p = 0.5 on a mask WITHOUT margin: the threshold cuts the
                                 # page in halves through the densest part of the margin histogram (measured 7.3e-3); with unit-scale logits
                                 # every uint8 value is sensitive to 1/255, so the uint8 mismatch rate is reported, not gated (62 %, steps <= 25)


def test_whole_page_bf16_end_points_logits_and_masks_with_unit_logit_scale(unit_scale_page):
    """the bf16 path was gated on probabilities of logit_scale = 0.05 weights only (a feature error is
    compressed ~20x there).  Same whole frame and weights as the fp32 test below: EVERY end point against the fp32 oracle relative to
    max|ref| (max and rms), the logits, and -- with the saturating softmax of unit-scale logits -- the probabilities, the uint8
    truncation and the threshold mask the separator post-processor consumes (separator_net_post_processor.py:141-157), each with an
    asserted bound."""
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from citlab_article_separation_new_amd.config import AruConfig
    from oracle import aru_oracle
    page, w, cfg, ref, inter = unit_scale_page
    g = helper.AruGraph(w, AruConfig(compute_dtype="bf16"))
    out, u8, mask = helper.get_net_output_fused(page, g, "0", want_u8=True, threshold=0.5)
    worst, worst_rms = ("", 0.0), ("", 0.0)
    rows = []
    for name in sorted(inter):
        if not (name.startswith("scale_") or name.startswith("att_")):
            continue
        got = helper.get_endpoint(g, name)
        want = inter[name]
        assert got.shape == want.shape, name
        scale = max(1.0, float(np.abs(want).max()))
        d = got - want
        rel, rms = float(np.abs(d).max()) / scale, float(np.sqrt(np.mean(d.astype(np.float64) ** 2))) / scale
        rows.append((name, rel, rms))
        worst, worst_rms = max(worst, (name, rel), key=lambda t: t[1]), max(worst_rms, (name, rms), key=lambda t: t[1])
        del got, d
    g.close()
    gl = helper.AruGraph(w, AruConfig(compute_dtype="bf16", apply_softmax=False))
    logits = helper.get_net_output(page, gl, "0")
    gl.close()
    lref = inter["logits"]
    lrel = float(np.abs(logits - lref).max()) / max(1.0, float(np.abs(lref).max()))
    perr = float(np.abs(out - ref).max())
    u8_ref = aru_oracle.to_uint8(ref)
    u8_rate = float((u8 != u8_ref).mean())
    u8_step = int(np.abs(u8.astype(np.int16) - u8_ref.astype(np.int16)).max())
    m_rate = float((mask != aru_oracle.apply_threshold(u8_ref, 0.5)).mean())
    print("\nbf16 whole frame, logit_scale 1: " + "; ".join(f"{n} {a:.1e}/{b:.1e}" for n, a, b in rows))
    print(f"bf16 whole frame, logit_scale 1: worst end point max {worst[0]} {worst[1]:.2e}, rms {worst_rms[0]} {worst_rms[1]:.2e}; logits rel "
          f"{lrel:.2e}; max|dp| = {perr:.2e}; uint8 mismatch rate {u8_rate:.3e} (max step {u8_step}); mask mismatch rate @0.5 = {m_rate:.3e}")
    assert all(rel <= BF16_ENDPOINT_GATE for _, rel, _ in rows), worst
    assert all(rms <= BF16_ENDPOINT_RMS_GATE for _, _, rms in rows), worst_rms
    assert lrel <= BF16_LOGIT_GATE and 1e-6 < perr <= BF16_PROB_GATE
    assert m_rate <= BF16_MASK_RATE_GATE
    # the mask may differ ONLY where the oracle itself has no margin: every pixel whose logit margin |l0 - l1| exceeds 2.0 (five
    # times the largest margin error measured, 0.4) must carry the oracle's mask value -- a wrong kernel fails this on confident
    # pixels, rounding cannot
    margin = np.abs(lref[:, :, 0] - lref[:, :, 1])
    confident = margin > BF16_CONFIDENT_MARGIN
    wrong = mask[:, :, 0] != aru_oracle.apply_threshold(u8_ref, 0.5)[:, :, 0]
    print(f"bf16 mask: {confident.mean():.1%} of the pixels are confident (margin > {BF16_CONFIDENT_MARGIN}); mismatches among them: "
          f"{int((wrong & confident).sum())}; largest margin of a flipped pixel: {float(margin[wrong].max()) if wrong.any() else 0.0:.3f}")
    assert confident.mean() > 0.4 and not (wrong & confident).any()
    # the fused epilogue is exactly uint8(p * 255) / apply_threshold of the engine's own float output, in bf16 too
    assert np.array_equal(u8, aru_oracle.to_uint8(out)) and np.array_equal(mask, aru_oracle.apply_threshold(u8, 0.5))


BF16_BLOCK_FRAME_MAX_GATE = 1.2e-2      # whole frame, block by block against the oracle with the engine's roundings: max|d| / max|ref|
                                        # (measured 7.5e-3: two bfloat16 steps of the largest value of one tensor)
BF16_BLOCK_FRAME_RMS_GATE = 2e-4        # rms(d) / max|ref| (measured 1.0e-4; free running against the same oracle: 2.0e-3; fp32 oracle: 3.4e-3)


def test_whole_page_bf16_block_by_block_against_the_oracle_with_the_same_roundings(unit_scale_page):
    """the same whole frame and weights against ``forward_torch(storage="bf16", teacher=<the engine's end points>)``: the oracle rounds
    filters and stored tensors to bfloat16 where the engine does and computes every end point from the ENGINE's upstream end points,
    so one block (conv1 + residual tail, or one deconvolution, or the attention CNN) is compared at a time.  Free running, two bf16
    evaluations of the net drift apart over forty layers almost like bf16 from fp32 (measured on this frame: max 1.9e-2, rms 2.0e-3 of
    max|ref|): that comparison cannot tell a wrong kernel from rounding; this one can."""
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from citlab_article_separation_new_amd.config import AruConfig
    from oracle import aru_oracle
    page, w, cfg, _, inter32 = unit_scale_page
    names = [n for n in sorted(inter32) if n.startswith("scale_") or n.startswith("att_")]
    g = helper.AruGraph(w, AruConfig(compute_dtype="bf16"))
    helper.get_net_output(page, g, "0")
    eng = {n: helper.get_endpoint(g, n) for n in names}
    g.close()
    _, forced = aru_oracle.forward_torch(page, w, cfg, return_intermediates=True, storage="bf16", teacher=eng)
    rows = []
    for n in names:
        want = forced[n]
        scale = max(1.0, float(np.abs(want).max()))
        d = eng[n] - want
        rows.append((n, float(np.abs(d).max()) / scale, float(np.sqrt(np.mean(d.astype(np.float64) ** 2))) / scale,
                     float((d != 0).mean())))
        del d
    worst, worst_rms = max(rows, key=lambda t: t[1]), max(rows, key=lambda t: t[2])
    print("\nbf16 whole frame, block by block vs the bf16-rounding oracle (max / rms / share of differing elements): "
          + "; ".join(f"{n} {a:.1e}/{b:.1e}/{c:.1e}" for n, a, b, c in rows))
    print(f"bf16 whole frame, block by block: worst max {worst[0]} {worst[1]:.2e}, worst rms {worst_rms[0]} {worst_rms[2]:.2e}")
    assert worst[1] <= BF16_BLOCK_FRAME_MAX_GATE and worst_rms[2] <= BF16_BLOCK_FRAME_RMS_GATE


@pytest.mark.parametrize("dtype", ["f32", "f32s"])
def test_whole_page_fp32_end_points_and_logits_with_unit_logit_scale(unit_scale_page, dtype):
    """the whole-frame gate above is on probabilities of weights with logit_scale = 0.05 (small logits compress a
    feature-map error ~20x before the 1e-4 gate).  Here: reference-rule weights (logit_scale = 1), EVERY end point of the
    3000 x 4500 frame within 2e-5 * max|ref| (the gate tests/test_aru_gpu.py applies up to 259 x 131), the LOGITS within the same
    relative gate, and the probabilities within 1e-4."""
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from citlab_article_separation_new_amd.config import AruConfig
    page, w, cfg, ref, inter = unit_scale_page
    g = helper.AruGraph(w, AruConfig(compute_dtype=dtype))
    out = helper.get_net_output(page, g, "0")
    worst = ("", 0.0)
    for name in sorted(inter):
        if not (name.startswith("scale_") or name.startswith("att_")):
            continue
        got = helper.get_endpoint(g, name)
        want = inter[name]
        assert got.shape == want.shape, name
        rel = float(np.abs(got - want).max()) / max(1.0, float(np.abs(want).max()))
        worst = max(worst, (name, rel), key=lambda t: t[1])
        assert rel <= 2e-5, (name, rel)
        del got
    perr = float(np.abs(out - ref).max())
    g.close()
    # logits: the same net without the class softmax
    cfg_l = AruConfig(apply_softmax=False, compute_dtype=dtype)
    gl = helper.AruGraph(w, cfg_l)
    logits = helper.get_net_output(page, gl, "0")
    lref = inter["logits"]
    lrel = float(np.abs(logits - lref).max()) / max(1.0, float(np.abs(lref).max()))
    gl.close()
    print(f"\n{dtype} whole frame, logit_scale 1: worst end point {worst[0]} rel {worst[1]:.2e}; logits max|l| {np.abs(lref).max():.2f} "
          f"rel {lrel:.2e}; max|dp| = {perr:.2e}; saturated pixels (p > 0.999): {(ref.max(axis=2) > 0.999).mean():.1%}")
    assert lrel <= 2e-5 and perr <= 1e-4


@pytest.fixture(scope="module")
def upstream_layout_page():
    """The UPSTREAM ARU-Net layout (ARU_v1.py:35-43 defaults of the paper's net: scale_space_num = 6, num_scales_att = 5; 1060 GFLOP per
    3000 x 4500 page) -- SURVEY section 8d asks for it because the shipped .pb's true cfg is unknown (nets/README.md:1-7).  Page 2,
    reference-rule weights (unit logit scale), the oracle's end points for the whole frame: the 256-channel level (141 x 94 at scale 0 down to
    9 x 6 at scale 4) finally sees more than one tile."""
    from citlab_article_separation_new_amd import synth
    from citlab_article_separation_new_amd.config import AruConfig
    from citlab_article_separation_new_amd.weights import init_aru_weights
    from oracle import aru_oracle
    cfg = AruConfig(scale_space_num=6, num_scales_att=5)
    w = init_aru_weights(cfg, 6565, bias_jitter=0.05, logit_scale=1.0)
    page = synth.synth_page(2, W, H).astype(np.float32) / 255.0
    ref, inter = aru_oracle.forward_torch(page, w, cfg, return_intermediates=True)
    return page, w, cfg, ref, inter


@pytest.mark.parametrize("dtype", ["f32s", "f32", "bf16"])
def test_upstream_layout_6_levels_5_attention_scales_whole_frame(upstream_layout_page, dtype):
    """VERDICT r4 missing #1 / next #3: the 6-level / 5-scale layout at full size.  fp32 arithmetics: every end point within 2e-5 max|ref|, logits
    within 2e-5, probabilities within 1e-4 -- the gates of the 5 / 3 layout; bf16: its end-point / logit gates."""
    from citlab_article_separation_new_amd import net_post_processing_helper as helper
    from citlab_article_separation_new_amd.config import AruConfig
    page, w, cfg, ref, inter = upstream_layout_page
    kw = dict(scale_space_num=6, num_scales_att=5, compute_dtype=dtype)
    g = helper.AruGraph(w, AruConfig(**kw))
    out = helper.get_net_output(page, g, "0")
    rows = []
    for name in sorted(inter):
        if not (name.startswith("scale_") or name.startswith("att_")):
            continue
        got = helper.get_endpoint(g, name)
        want = inter[name]
        assert got.shape == want.shape, name
        scale = max(1.0, float(np.abs(want).max()))
        d = got - want
        rows.append((name, float(np.abs(d).max()) / scale, float(np.sqrt(np.mean(d.astype(np.float64) ** 2))) / scale))
        del got, d
    g.close()
    gl = helper.AruGraph(w, AruConfig(apply_softmax=False, **kw))
    logits = helper.get_net_output(page, gl, "0")
    gl.close()
    lref = inter["logits"]
    lrel = float(np.abs(logits - lref).max()) / max(1.0, float(np.abs(lref).max()))
    perr = float(np.abs(out - ref).max())
    worst, worst_rms = max(rows, key=lambda t: t[1]), max(rows, key=lambda t: t[2])
    n256 = [r for r in rows if "_unet_down_5_" in r[0]]
    print(f"\n{dtype} upstream layout (6 levels, 5 attention scales) whole frame: {len(rows)} end points, worst max {worst[0]} {worst[1]:.2e}, worst rms "
          f"{worst_rms[0]} {worst_rms[2]:.2e}; 256-channel level: {max(r[1] for r in n256):.2e}; logits rel {lrel:.2e} (max|l| {np.abs(lref).max():.1f}); "
          f"max|dp| {perr:.2e}")
    assert len(n256) == 5 and inter["scale_0_unet_down_5_conv"].shape == (141, 94, 256)
    if dtype == "bf16":
        assert all(r[1] <= BF16_ENDPOINT_GATE for r in rows), worst
        assert all(r[2] <= BF16_ENDPOINT_RMS_GATE for r in rows), worst_rms
        assert lrel <= BF16_LOGIT_GATE and perr <= BF16_PROB_GATE
    else:
        assert all(r[1] <= 2e-5 for r in rows), worst
        assert lrel <= 2e-5 and perr <= 1e-4


def test_c1_crop_512x768_through_the_separator_cli(tmp_path):
    """BASELINE configs[0]: one 512 x 768 crop, --fixed_height 768 so that the net sees the crop itself."""
    from citlab_article_separation_new_amd import image_io, pb_import, polygonize, synth
    from citlab_article_separation_new_amd import run_net_post_processing as cli
    from citlab_article_separation_new_amd.config import AruConfig
    from citlab_article_separation_new_amd.host_util import rescale_points
    from citlab_article_separation_new_amd.page_xml import Page
    from citlab_article_separation_new_amd.weights import init_aru_weights
    from oracle import aru_oracle, classical_oracle as co
    cfg = AruConfig()
    w = init_aru_weights(cfg, 77, bias_jitter=0.05, logit_scale=0.05)
    pb = tmp_path / "separator_aru.pb"
    pb.write_bytes(tf_aru_graph.build_aru_pb(w, cfg))      # laid out like a TF1 freeze, serialised by protobuf
    data = tmp_path / "data"
    (data / "page").mkdir(parents=True)
    crop = np.ascontiguousarray(synth.synth_page(0, W, H)[0:768, 0:512])          # C1 = page 0 [0:768, 0:512]
    Image.fromarray(crop).save(data / "c1.png")
    lst = tmp_path / "images.lst"
    lst.write_text(str(data / "c1.png") + "\n")
    # the reference's sequence (separator_net_post_processor.py:141-157) with the oracle at every step
    img = image_io.load_image_bgr(str(data / "c1.png"))
    _, grey, sc = co.scale_and_gray(img, 768, 1.0)
    assert sc == 1.0 and grey.shape == (768, 512)
    prob = aru_oracle.forward_torch(grey.astype(np.float32), w, cfg)
    thr = round(float(np.median(prob[:, :, 0])), 3)
    net_u8 = aru_oracle.to_uint8(prob)
    post = co.separator_post_process(aru_oracle.apply_threshold(net_u8, thr))
    polygons = {f"SeparatorRegion_{o}": [[rescale_points(r, 1 / sc) for r in poly] for poly in polygonize.shapes(post[o])]
                for o in ("horizontal", "vertical")}
    assert polygons["SeparatorRegion_horizontal"] or polygons["SeparatorRegion_vertical"], "no separators; adjust the threshold"
    assert co.cc_min_size(768 * 512, 1 / (768 * 512) * 100) == 99     # float64 gives 99 at this size
    # expected regions: the writer on the oracle-side polygons (hole cutting: tests/test_region_writer_split.py)
    from citlab_article_separation_new_amd.region_to_page_writer import SeparatorRegionToPageWriter
    writer = SeparatorRegionToPageWriter(str(tmp_path / "none.xml"), str(data / "c1.png"), 768, 1.0, polygons)
    writer.merge_regions()
    want = [(s.get_orientation(), s.points) for s in writer.page_object.get_regions()["SeparatorRegion"]]
    rc = cli.main(["--path_to_image_list", str(lst), "--path_to_pb", str(pb), "--mode", "separator",
                   "--fixed_height", "768", "--threshold", str(thr), "--num_processes", "1"])
    assert rc == 0
    seps = Page(str(data / "page" / "c1.xml.xml")).get_regions()["SeparatorRegion"]
    got = [(s.get_orientation(), s.points) for s in seps]
    # uint8 truncation may flip single pixels where p * 255 is within 1e-6 of an integer AND that integer is the
    # threshold; with identical masks the polygons are identical
    assert got == want and len(got) > 100


def test_swt_and_separator_stage_at_full_size_bit_exact():
    """configs[2] size: 255 - gray -> Gaussian -> Otsu -> exact EDT -> uint8 on a 3000 x 4500 scan, and CC filter +
    openings on a 3000 x 4500 two-class mask, both bit-identical to the classical oracle."""
    from citlab_article_separation_new_amd import image_ops, synth
    from oracle import classical_oracle as co
    g = synth.synth_page(1, W, H)
    g[900:1500, 300:1100] = 12                               # a dark picture block: distances beyond 255 wrap in uint8
    out, thr, d2 = image_ops.swt_distance_transform(g, return_details=True)
    inv = (255 - g.astype(np.int64)).astype(np.uint8)
    blur = co.gaussian5(inv)
    assert thr == co.otsu_threshold(blur)
    assert np.array_equal(d2.astype(np.int64), co.edt_sq(((blur > thr) * 255).astype(np.uint8)))
    ref = co.swt_distance_transform(g)
    assert np.array_equal(out, ref)
    assert int(np.sqrt(d2.max())) > 255 and out.max() > 200, "the wrap-around case must be exercised"
    # separator stage on a mask made from the page itself: dark pixels + their horizontal / vertical rules
    mask = np.zeros((H, W, 2), np.uint8)
    mask[:, :, 0] = (g < 120) * 255
    mask[:, :, 1] = (g < 60) * 255
    post, post_ref = image_ops.separator_post_process(mask), co.separator_post_process(mask)
    for k in ("horizontal", "vertical"):
        assert np.array_equal(post[k], post_ref[k]), k
        assert post_ref[k].any()
