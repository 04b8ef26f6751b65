"""The arithmetic behind compute_dtype "f32s" (csrc/split_kernels.h), checked on the CPU with numpy:
  * an fp32 number is EXACTLY the sum of its three bfloat16 parts h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)
    (round to nearest even at every cut, as v_cvt_pk_bf16_f32 and the host packer `f2bf` do);
  * every partial product of two parts is exact in fp32 (8 x 8 significand bits);
  * the six partial products the kernels add (hh, hm, mh, mm, hl, lh) miss x * w by at most 2^-23 |x w| -- the three dropped ones
    (ml, lm, ll) are bounded by 2 * 2^-25 |x w| + 2^-34 |x w|;
so a split-product convolution is an fp32 convolution whose PRODUCTS are at least as accurate as an fp32 multiply; its SUMS take six fp32
additions per product where the plain path takes one, so the accumulated rounding is a little larger (emulated below: 2.5x on a K = 576 dot
product added term by term; measured on the GPU, where an MFMA adds 32 products inside: 1.7x on the end points of a whole frame -- 3.3e-6 against
1.9e-6 of max|ref|, gate 2e-5).  Range: |x| below the largest bfloat16 (3.39e38; beyond it the first part rounds to infinity)."""
import numpy as np


def bf16_rne(x):
    """fp32 -> the nearest bfloat16 (ties to even), returned as fp32"""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    h = bf16_rne(x)
    r = (x - h).astype(np.float32)
    m = bf16_rne(r)
    l = bf16_rne((r - m).astype(np.float32))
    return h, m, l


def _samples(n=200000, seed=0):
    rng = np.random.default_rng(seed)
    mant = rng.random(n, dtype=np.float32) + np.float32(1.0)
    expo = rng.integers(-60, 60, n)
    sign = rng.choice(np.array([-1.0, 1.0], dtype=np.float32), n)
    x = (sign * np.ldexp(mant, expo)).astype(np.float32)
    x[:16] = np.array([0.0, 1.0, -1.0, 3.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 255.99998, 65504.0, 3.3e38, -3.3e38, 1e-30, 0.1, 1 / 3, 2 / 3,
                       np.float32(np.pi), 1.0000001], dtype=np.float32)
    return x


def test_three_bfloat16_parts_sum_to_the_fp32_number_exactly():
    x = _samples()
    h, m, l = split3(x)
    # the differences x - h and (x - h) - m are exact in fp32 (checked in float64), and so is the reconstruction
    x64, h64, m64, l64 = (a.astype(np.float64) for a in (x, h, m, l))
    assert np.array_equal((x - h).astype(np.float64), x64 - h64)
    assert np.array_equal(((x - h) - m).astype(np.float64), x64 - h64 - m64)
    assert np.array_equal(h64 + m64 + l64, x64)
    # part magnitudes: |m| <= 2^-8 |x|, |l| <= 2^-16 |x| (half an ulp of an 8-bit significand at every cut)
    ax = np.abs(x64)
    assert np.all(np.abs(m64) <= ax * 2.0 ** -8) and np.all(np.abs(l64) <= ax * 2.0 ** -16)
    # every part is a bfloat16 (low 16 bits clear)
    for p in (h, m, l):
        assert not np.any(p.view(np.uint32) & 0xFFFF)


def test_partial_products_are_exact_and_six_of_nine_are_enough():
    x, w = _samples(seed=1), _samples(seed=2)
    with np.errstate(divide="ignore"):
        keep = (np.abs(np.log2(np.abs(x))) < 40) & (np.abs(np.log2(np.abs(w))) < 40) & (x != 0) & (w != 0)   # (no overflow / underflow of x w)
    x, w = x[keep], w[keep]
    xs, ws = split3(x), split3(w)
    exact = x.astype(np.float64) * w.astype(np.float64)
    six = np.zeros_like(exact)
    for i, j in ((0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)):
        p32 = (xs[i] * ws[j]).astype(np.float32)                      # what one lane of the MFMA multiplies
        p64 = xs[i].astype(np.float64) * ws[j].astype(np.float64)
        assert np.array_equal(p32.astype(np.float64), p64)            # 8 x 8 significand bits: exact in fp32
        six += p64
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() <= 2.0 ** -23, rel.max()
    # for comparison: the rounding of the product to fp32 itself is up to 2^-24
    fp32_rel = np.abs((x * w).astype(np.float64) - exact) / np.abs(exact)
    print(f"six-term split: max rel {rel.max():.2e}, mean {rel.mean():.2e};  one fp32 multiply: max {fp32_rel.max():.2e}, mean {fp32_rel.mean():.2e}")
    assert rel.mean() <= fp32_rel.mean()


def test_a_dot_product_of_split_terms_matches_the_fp32_one():
    """K = 576 (a 3x3 conv over 64 channels): six-term split products accumulated in fp32 against fp32 products accumulated in fp32,
    both against float64 -- the split sum is not worse"""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2000, 576)).astype(np.float32)
    w = (rng.standard_normal((576,)) * 0.05).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64)
    plain = np.zeros(2000, np.float32)
    split = np.zeros(2000, np.float32)
    xs, ws = split3(x), split3(w)
    for k in range(576):
        plain = (plain + x[:, k] * w[k]).astype(np.float32)
        for i, j in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)):           # smallest terms first, like the kernels
            split = (split + (xs[i][:, k] * ws[j][k]).astype(np.float32)).astype(np.float32)
    scale = np.abs(ref).max()
    e_plain, e_split = np.abs(plain - ref).max() / scale, np.abs(split - ref).max() / scale
    print(f"K = 576: fp32 products {e_plain:.2e}, split products {e_split:.2e} of max|ref|")
    assert e_split <= 4 * e_plain + 1e-7 and e_split <= 5e-6
