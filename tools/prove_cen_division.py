"""Is  q' = fma(r, y, q)  with  y = RN(1 / m), q = RN(g * y), r = fma(-m, q, g)  the correctly rounded g / m for EVERY pair the
cen2019 kernels can meet?  g and m are range gradients |fl(a / 255) - fl(b / 255)| of byte pairs (m = the image's largest, so
g <= m): at most 32 641 values each.  All pairs are tried; fp32 operations are emulated in fp64 where that is exact (products of
two fp32, the residual) and checked with exact rationals wherever an fp64 intermediate lands within 2^-50 of an fp32 rounding
boundary.  Usage: python tools/prove_cen_division.py [--sample N]   (the full run takes a few minutes)"""
import sys
from fractions import Fraction

import numpy as np

F = np.float32


def rn32_exact(fr):
    """correctly rounded fp32 of a positive rational (ties to even)"""
    if fr == 0:
        return F(0.0)
    a = float(fr)
    c = F(a)
    lo, hi = np.nextafter(c, F(0)), np.nextafter(c, F(np.inf))
    best = min((abs(Fraction(float(x)) - fr), int(np.float32(x).view(np.uint32)) & 1, float(x)) for x in (lo, c, hi))
    return F(best[2])


def near_boundary(s64):
    """fp64 values within 2^-50 (relative) of the midpoint of two neighbouring fp32 values"""
    c = s64.astype(F)
    up = np.nextafter(c, F(np.inf)).astype(np.float64)
    dn = np.nextafter(c, F(-np.inf)).astype(np.float64)
    mid_up, mid_dn = 0.5 * (c.astype(np.float64) + up), 0.5 * (c.astype(np.float64) + dn)
    tol = np.abs(s64) * 2.0 ** -50
    return (np.abs(s64 - mid_up) <= tol) | (np.abs(s64 - mid_dn) <= tol)


def main():
    t = (np.arange(256, dtype=F) / F(255.0)).astype(F)
    G = np.unique(np.abs(t[:, None] - t[None, :]).astype(F))
    ms = G[G > 0]
    if "--sample" in sys.argv:
        n = int(sys.argv[sys.argv.index("--sample") + 1])
        rng = np.random.default_rng(0)
        ms = np.unique(np.concatenate([ms[:50], ms[-50:], rng.choice(ms, n)]))
    bad = exact_checked = total = 0
    G64 = G.astype(np.float64)
    for m in ms:
        g = G[G <= m]
        g64 = g.astype(np.float64)
        m64 = np.float64(m)
        y = F(np.float64(1.0) / m64)                       # 1 / m in fp64 then fp32: checked exactly below
        if near_boundary(np.array([1.0 / m64]))[0]:
            y = rn32_exact(Fraction(1) / Fraction(float(m)))
        y64 = np.float64(y)
        q = (g64 * y64).astype(F)                          # exact product, one rounding
        r = (g64 - m64 * q.astype(np.float64)).astype(F)    # exact residual, one rounding
        s = q.astype(np.float64) + r.astype(np.float64) * y64   # the product is exact, the sum is not always
        qq = s.astype(F)
        ref64 = g64 / m64
        ref = ref64.astype(F)
        sus = near_boundary(s) | near_boundary(ref64)
        for i in np.nonzero(sus)[0]:
            exact_checked += 1
            fr_s = Fraction(float(q[i])) + Fraction(float(r[i])) * Fraction(float(y))
            qq[i] = rn32_exact(fr_s) if fr_s > 0 else F(0.0)
            ref[i] = rn32_exact(Fraction(float(g[i])) / Fraction(float(m))) if g[i] > 0 else F(0.0)
        bad += int(np.count_nonzero(qq != ref))
        total += len(g)
    print(f"{len(G)} gradient values, {len(ms)} divisors, {total} pairs, {exact_checked} checked with exact rationals, mismatches: {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
