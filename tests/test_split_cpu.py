"""The arithmetic claim behind the split-product rollout (csrc/cem_rollout_split.h), checked on the CPU in numpy:
an fp32 number is the EXACT sum of three bf16 pieces (round-to-nearest-even split), every bf16 x bf16 product is exact in fp32, and
the six products with i + j <= 2 reproduce w * x to within 2^-22 of |w * x| — the scale of one fp32 rounding.  (The truncation
split of the first version is exact too but only reaches 2^-21: its pieces shrink by 2^-7 each, not 2^-8.)"""
import numpy as np


def rn_bf16(x):
    """cem_rn_bf16_bits: round to nearest even at 8 significand bits, as a float32 whose low 16 bits are zero."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x, rn=True):
    """cem_split3_bits: the three pieces as float32 arrays that are bf16 values."""
    x = np.asarray(x, np.float32)
    top = rn_bf16 if rn else (lambda v: (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32))
    a0 = top(x)
    r1 = x - a0
    a1 = top(r1)
    a2 = r1 - a1
    return a0, a1, a2


def _samples():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200000), rng.standard_normal(50000) * 1e-6, rng.standard_normal(50000) * 1e6,
                        np.array([0.0, -0.0, 1.0, -1.0, 3.3e38, -3.3e38, 1.1754944e-38, 2.0 ** -100, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24])]).astype(np.float32)
    return x


def test_three_bf16_pieces_are_an_exact_split():
    x = _samples()
    a0, a1, a2 = split3(x)
    for a in (a0, a1, a2):
        assert np.all((a.view(np.uint32) & np.uint32(0xFFFF)) == 0), 'a piece is not a bf16 value'
    # exact in fp64 (every piece and the sum are representable): no rounding anywhere
    np.testing.assert_array_equal(a0.astype(np.float64) + a1.astype(np.float64) + a2.astype(np.float64), x.astype(np.float64))
    # and the fp32 subtractions that produce the pieces were exact too
    np.testing.assert_array_equal((x - a0) - a1, a2)


def test_six_leading_products_carry_the_product_to_fp32_grade():
    rng = np.random.default_rng(1)
    w = (rng.standard_normal(300000) * rng.choice([1e-3, 1.0, 30.0], 300000)).astype(np.float32)
    x = (rng.standard_normal(300000) * rng.choice([1e-2, 1.0, 5.0], 300000)).astype(np.float32)
    ws, xs = split3(w), split3(x)
    exact = w.astype(np.float64) * x.astype(np.float64)
    kept = np.zeros_like(exact)
    for i in range(3):
        for j in range(3):
            p = ws[i].astype(np.float64) * xs[j].astype(np.float64)
            # a bf16 x bf16 product has at most 16 significant bits: exact in an fp32 accumulator
            np.testing.assert_array_equal(p, (ws[i] * xs[j]).astype(np.float64))
            if i + j <= 2:
                kept += p
    rel = np.abs(kept - exact) / np.maximum(np.abs(exact), 1e-300)
    assert rel.max() <= 2.0 ** -22, rel.max()
    # the truncation split, for comparison: exact as a split, but its dropped products reach 2^-21
    wt, xt = split3(w, rn=False), split3(x, rn=False)
    kept_t = sum(wt[i].astype(np.float64) * xt[j].astype(np.float64) for i in range(3) for j in range(3) if i + j <= 2)
    rel_t = np.abs(kept_t - exact) / np.maximum(np.abs(exact), 1e-300)
    assert 2.0 ** -22 < rel_t.max() <= 2.0 ** -20
    # what an fp32 multiply itself loses, for scale
    assert (np.abs((w * x).astype(np.float64) - exact) / np.maximum(np.abs(exact), 1e-300)).max() <= 2.0 ** -24 * 1.0001


def test_dot_products_of_split_operands_match_fp32_dot_products():
    """A 128-term dot product (one hidden unit): six-product form accumulated in fp32 vs the fp64 value, next to plain fp32."""
    rng = np.random.default_rng(2)
    W = (rng.standard_normal((512, 128)) / np.sqrt(128)).astype(np.float32)
    X = np.maximum(rng.standard_normal((512, 128)), 0).astype(np.float32)
    exact = (W.astype(np.float64) * X.astype(np.float64)).sum(axis=1)
    ws, xs = split3(W), split3(X)
    acc = np.zeros(512, np.float32)
    for (i, j) in ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)):      # the kernel's order: smallest terms first
        for k0 in range(0, 128, 32):                                      # one MFMA adds 32 exact products to the accumulator
            acc = (acc.astype(np.float64) + (ws[i][:, k0:k0 + 32].astype(np.float64) * xs[j][:, k0:k0 + 32].astype(np.float64)).sum(axis=1)).astype(np.float32)
    plain = np.zeros(512, np.float32)
    for k in range(128):
        plain = plain + W[:, k] * X[:, k]
    scale = np.abs(W.astype(np.float64) * X.astype(np.float64)).sum(axis=1)
    e_split, e_plain = np.abs(acc - exact) / scale, np.abs(plain - exact) / scale
    assert e_split.max() <= 4e-7 and e_split.max() <= 2.0 * max(e_plain.max(), 6e-8), (e_split.max(), e_plain.max())
