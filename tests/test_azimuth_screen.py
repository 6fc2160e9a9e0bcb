"""The azimuth-row screen of the strip kernel (csrc/frontend.hip az_row_screened), restated in numpy and checked against the exact
form it replaces (az_row_of = oracle/frontend_ref.c feref_cart_map): whenever the screen ACCEPTS -- the wrapped reciprocal-multiply
value w is further than delta = rows * 2^-48 from 0 and from rows, and w - delta and w + delta round to the same float -- that float
must be the float of the exact fp64 division + wrap.  Random angles and grids, and angles constructed to put the quotient within a
few ulps of a float rounding boundary or of a wrap threshold (where the screen must either refuse or still be right)."""
import numpy as np


def exact_row(th, az0, st, rows):
    """(float)wrap((th - az0) / st) as frontend_ref.c / az_row_of compute it (fp64 division, fmod, two corrections)."""
    a = (th - az0) / st
    a = np.fmod(a, float(rows))
    a = np.where(a < 0, a + rows, a)
    a = np.where(a >= rows, a - rows, a)
    return a.astype(np.float32)


def screened_row(th, az0, st, rows):
    """(accepted, float): the device's fast path."""
    R = float(rows)
    rst = np.float64(1.0) / np.float64(st)
    delta = R * 2.0 ** -48
    q = (th - az0) * rst
    w = np.where(q >= R, q - R, q)
    w = np.where(q < 0.0, q + R, w)
    lo = (w - delta).astype(np.float32)
    hi = (w + delta).astype(np.float32)
    ok = (w > delta) & (w < R - delta) & (lo == hi)
    return ok, lo


def _check(th, az0, st, rows):
    with np.errstate(all="ignore"):
        ok, f = screened_row(th, az0, st, rows)
        want = exact_row(th, az0, st, rows)
    assert np.array_equal(f[ok].view(np.uint32), want[ok].view(np.uint32)), int((f[ok] != want[ok]).sum())
    return float(ok.mean())


def test_random_angles_and_grids():
    rng = np.random.default_rng(0)
    accepted = []
    for rows in (400, 7, 1024, 399):
        for _ in range(6):
            az0 = float(np.float32(rng.uniform(-0.2, 6.4)))
            st = float(np.float32(2 * np.pi / rows * rng.uniform(0.98, 1.02)))
            th = rng.uniform(0.0, 2 * np.pi, 400_000)
            accepted.append(_check(th, az0, st, rows))
    assert min(accepted) > 0.999          # the exact branch is the rare one


def test_quotients_next_to_float_rounding_boundaries_and_wrap_thresholds():
    rng = np.random.default_rng(1)
    rows = 400
    refused = 0
    total = 0
    for rep in range(8):
        az0 = float(np.float32(rng.uniform(0.0, 0.02)))
        st = float(np.float32(2 * np.pi / rows * rng.uniform(0.999, 1.001)))
        # float midpoints m in (0, rows): halfway between consecutive floats
        f = rng.uniform(1e-3, rows, 60_000).astype(np.float32)
        m = (f.astype(np.float64) + np.nextafter(f, np.float32(np.inf)).astype(np.float64)) / 2.0
        for wrap in (0.0, float(rows), -float(rows)):                      # the same rows reached directly, from above and from below
            base = az0 + st * (m + wrap)
            for k in (-6, -3, -2, -1, 0, 1, 2, 3, 6):
                th = base * (1.0 + k * 2.0 ** -52)
                th = th + rng.integers(-2, 3, th.shape) * np.spacing(th)
                with np.errstate(all="ignore"):
                    ok, _ = screened_row(th, az0, st, rows)
                refused += int((~ok).sum())
                total += ok.size
                _check(th, az0, st, rows)
        # quotients within a few ulps of 0, rows, -rows, 2 rows
        for thr in (0.0, float(rows), -float(rows), 2.0 * rows):
            th = az0 + st * thr + rng.integers(-40, 41, 20_000) * np.spacing(az0 + st * max(abs(thr), 1.0))
            _check(th, az0, st, rows)
    assert refused > 0.05 * total        # the construction does reach the screen's refusals


def test_degenerate_grids():
    th = np.linspace(0.0, 6.28, 1000)
    for az0, st in ((0.0, 0.0), (0.0, -0.0157), (0.0, np.nan), (np.inf, 0.0157), (0.0, 1e-300)):
        with np.errstate(all="ignore"):
            ok, f = screened_row(th, az0, st, 400)
            want = exact_row(th, az0, st, 400)
        assert np.array_equal(f[ok].view(np.uint32), want[ok].view(np.uint32))   # whatever is accepted is right (a descending grid is a grid)
        if not (np.isfinite(st) and st != 0.0 and np.isfinite(az0)):
            assert not ok.any()                                                    # NaN / inf quotients are never accepted
