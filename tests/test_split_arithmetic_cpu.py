"""The arithmetic claims behind the split-precision kernels (csrc/conv_split.hip, csrc/split_f16.h),
checked in numpy without a GPU:

* f16x3: two round-to-nearest f16 pieces of a power-of-two-scaled value leave <= 2^-22 relative
  error, and the three products x0w0 + x0w1 + x1w0 are again fp32-chain accurate — including
  operands whose magnitudes span many binades below the scale group's maximum.
Plus the host-side tile planner of the split conv (pure Python).
"""
import numpy as np
import pytest


def _split_f16(x, scale_exp):
    xs = (x * np.float32(2.0) ** np.float32(14 - scale_exp)).astype(np.float32)
    h0 = xs.astype(np.float16)
    h1 = (xs - h0.astype(np.float32)).astype(np.float32).astype(np.float16)
    return h0, h1


def _exponent(m):
    return int(np.floor(np.log2(m)))


def _blocked_sum(terms, K, blk):
    """fp32 accumulator; each `blk`-wide block of exact products is added with one rounding (the MFMA
    accumulates in fp32; products of 16-bit pieces are exact in fp32)."""
    acc = np.zeros(terms[0][0].shape[:1] + terms[0][1].shape[1:], np.float32)
    for k0 in range(0, K, blk):
        for a, b in terms:
            acc = (acc.astype(np.float64) + a[:, k0:k0 + blk].astype(np.float64) @ b[k0:k0 + blk].astype(np.float64)).astype(np.float32)
    return acc


@pytest.mark.parametrize("spread", [0, 12, 24])
def test_f16_two_piece_split_three_products(spread):
    rng = np.random.default_rng(1 + spread)
    K, M, N = 576, 128, 64
    a = (rng.standard_normal((M, K)) * np.exp2(-rng.integers(0, spread + 1, (M, K)))).astype(np.float32)
    b = (rng.standard_normal((K, N)) * 0.05).astype(np.float32)
    ea, eb = _exponent(np.abs(a).max()), _exponent(np.abs(b).max())
    a0, a1 = _split_f16(a, ea)
    b0, b1 = _split_f16(b, eb)
    assert np.isfinite(a0.astype(np.float32)).all() and np.abs(a0.astype(np.float32)).max() < 2.0 ** 15 + 1
    # representation error of the two pieces: <= 2^-22 relative for values within 2^17 of the maximum
    rec = (a0.astype(np.float64) + a1.astype(np.float64)) * 2.0 ** (ea - 14)
    big = np.abs(a) > np.abs(a).max() * 2.0 ** -17
    assert (np.abs(rec - a)[big] / np.abs(a)[big]).max() <= 2.0 ** -22
    assert np.abs(rec - a).max() <= 2.0 ** -22 * np.abs(a).max()
    terms = [(a1.astype(np.float32), b0.astype(np.float32)), (a0.astype(np.float32), b1.astype(np.float32)),
             (a0.astype(np.float32), b0.astype(np.float32))]
    got = _blocked_sum(terms, K, 16).astype(np.float64) * 2.0 ** (ea - 14) * 2.0 ** (eb - 14)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    chain = np.zeros((M, N), np.float32)
    for k in range(K):
        chain = (chain + a[:, k:k + 1] * b[k:k + 1, :]).astype(np.float32)
    scale = np.abs(ref).max()
    e_split, e_chain = np.abs(got - ref).max() / scale, np.abs(chain - ref).max() / scale
    assert e_split < 1e-6 and e_split < 2 * e_chain + 1e-7, (e_split, e_chain)


def test_split_tile_planner():
    from implicit_depth_amd import nhwc

    # full-resolution / half-resolution decoder layers at the bench batch: 16-row tiles
    assert nhwc.choose_split_rows(32, 192, 256, 64) == 16
    assert nhwc.choose_split_rows(32, 96, 128, 128) == 16
    # 24-row maps would waste a quarter of a second 16-row tile; small batches need more workgroups
    assert nhwc.choose_split_rows(32, 24, 32, 256) == 8
    assert nhwc.choose_split_rows(4, 96, 128, 128) == 8
    assert nhwc.choose_split_rows(1, 192, 256, 64) == 8

    class C:  # the two attributes the eligibility rule looks at
        def __init__(self, k, s):
            self.kernel_size, self.stride = (k, k), (s, s)

    ok = lambda srcs, cout, N=32, H=96, W=128: nhwc.split_eligible(srcs, cout, N, H, W, nhwc.PAD_ZEROS)
    assert ok([(None, C(3, 1))], 64)
    assert ok([(None, C(3, 1)), (None, C(1, 1))], 128)           # fused 1x1 projection
    assert not ok([(None, C(3, 2))], 64)                          # strided
    assert not ok([(None, C(3, 1)), (None, C(3, 2))], 64)         # strided 3x3 projection
    assert not ok([(None, C(3, 1))], 48)                          # Cout % 64
    assert not ok([(None, C(1, 1))], 64)                          # 1x1 main conv
    assert not nhwc.split_eligible([(None, C(3, 1))], 64, 1, 8, 16, nhwc.PAD_ZEROS)  # too few tiles
    assert not nhwc.split_eligible([(None, C(3, 1))], 64, 32, 96, 128, nhwc.PAD_REPLICATE)
