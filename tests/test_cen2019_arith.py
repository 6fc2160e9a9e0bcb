"""The two arithmetic arguments the round-4 cen2019 kernels rest on, checked exhaustively / on dense samples on the CPU
(csrc/cen2019.hip: cen_stats in the integer domain, the fixed-point sum of h from two 32-bit halves)."""
import numpy as np

F = np.float32


def test_largest_byte_difference_gives_the_largest_gradient():
    """g(a, b) = |fl(a / 255) - fl(b / 255)| in fp32, for every pair of bytes: a pair with a larger |a - b| ALWAYS has the
    larger g, so the maximum of g over an image is reached among the pixels with the largest integer difference -- cen_stats
    only forms the float g of those."""
    t = (np.arange(256, dtype=F) / F(255.0)).astype(F)              # correctly rounded division, as __fdiv_rn
    g = np.abs(t[:, None] - t[None, :]).astype(F)                    # fp32 subtraction of two fp32 values
    d = np.abs(np.arange(256)[:, None] - np.arange(256)[None, :])
    lo = np.array([g[d == k].min() for k in range(256)])
    hi = np.array([g[d == k].max() for k in range(256)])
    assert np.all(lo[1:] > hi[:-1]), "a larger byte difference must give a strictly larger g"
    assert float((lo[1:] - hi[:-1]).min()) > 3.9e-3                  # one step of 1 / 255, minus 1.8e-7 of rounding
    assert hi[0] == 0.0
    # ... and within one difference the float values do differ (so the pixels that reach it all have to be looked at)
    assert any(hi[k] > lo[k] for k in range(1, 255))


def _llrint_scaled(h):
    """llrint(h * 2^40) exactly, h fp32 with |h| <= 1 (the product is exact in fp64 for |h| >= 2^-12; below that it is still exact
    because h * 2^40 has at most 24 significant bits)"""
    return np.rint(h.astype(np.float64) * 2.0 ** 40).astype(np.int64)


def test_fixed_point_sum_from_two_halves():
    """x = h * 2^20; hi = rint(x); lo = rint((x - hi) * 2^20): hi * 2^20 + lo == llrint(h * 2^40) for every fp32 h in [-1, 1]
    (x - hi is exact, hi * 2^20 is an even integer, so the tie rule is preserved)."""
    rng = np.random.default_rng(1)
    parts = [rng.uniform(-1, 1, 2_000_000).astype(F),
             (rng.uniform(-1, 1, 500_000) * 2.0 ** -rng.integers(0, 60, 500_000)).astype(F),   # all magnitudes down to 2^-60
             np.array([0.0, -0.0, 1.0, -1.0, 2.0 ** -21, 3 * 2.0 ** -21, 2.0 ** -41, 3 * 2.0 ** -41, -(2.0 ** -41), 1 - 2.0 ** -24,
                       0.5 + 2.0 ** -24, 2.0 ** -20 + 2.0 ** -41, 2.0 ** -20 * 1.5], dtype=F)]
    # products of the kernel's own shape: sv * (1 - gn) with byte-derived operands
    t = (np.arange(256, dtype=F) / F(255.0)).astype(F)
    sv = (t[rng.integers(0, 256, 500_000)] - F(0.0934)).astype(F)
    gn = (np.abs(t[rng.integers(0, 256, 500_000)] - t[rng.integers(0, 256, 500_000)]).astype(F) / F(0.95686275)).astype(F)
    parts.append((sv * (F(1.0) - np.minimum(gn, F(1.0))).astype(F)).astype(F))
    h = np.concatenate(parts)
    x = (h * F(1048576.0)).astype(F)
    assert np.array_equal(x.astype(np.float64), h.astype(np.float64) * 1048576.0)        # a power-of-two scaling: exact
    hi = np.rint(x).astype(F)
    rem = (x - hi).astype(F)
    assert np.array_equal(rem.astype(np.float64), x.astype(np.float64) - hi.astype(np.float64))   # exact difference
    lo = np.rint((rem * F(1048576.0)).astype(F))
    got = hi.astype(np.int64) * (1 << 20) + lo.astype(np.int64)
    assert np.array_equal(got, _llrint_scaled(h))
    assert np.abs(hi).max() <= 2 ** 20 and np.abs(lo).max() <= 2 ** 19


def test_three_operation_division_is_correctly_rounded_on_the_gradient_domain():
    """row_load_h divides a range gradient by the image's largest one with y = RN(1 / m), q = RN(g y), r = fma(-m, q, g),
    q' = fma(r, y, q).  That is the correctly rounded quotient for EVERY pair the kernels can meet: tools/prove_cen_division.py
    tries all of them (598 gradient values, 179 100 pairs with g <= m)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("prove_cen_division", os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "tools", "prove_cen_division.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import sys
    argv, sys.argv = sys.argv, ["prove_cen_division.py"]
    try:
        assert mod.main() == 0
    finally:
        sys.argv = argv
