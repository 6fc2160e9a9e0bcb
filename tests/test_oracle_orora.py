"""Pins the ORORA oracle (oracle/orora_ref.c).  PARITY UNPINNED w.r.t. the reference (its ORORA
submodule is absent); what can be pinned is that the restated algorithm does its job: it
recovers known SE(2) motions from outlier-contaminated matches, and its scalar TLS estimator
agrees with a brute-force evaluation of the same cost."""
import numpy as np

from navtech_radar_slam_amd import synth


def test_recovers_ground_truth(oracle):
    src, dst, off, truth = synth.orora_pairs(777, 60)
    r = oracle.orora_register_batch(src, dst, off, nthreads=4)
    assert np.all(r["status"] == 0)
    assert np.abs(r["x"] - truth[:, 0]).max() < 0.05 and np.abs(r["y"] - truth[:, 1]).max() < 0.05
    assert np.abs(r["yaw"] - truth[:, 2]).max() < 2e-3
    k = off[1:] - off[:-1]
    assert np.all(r["rot_inliers"] > 0.1 * k) and np.all(r["trans_inliers"] > 0.35 * k)
    assert np.all((r["iterations"] > 5) & (r["iterations"] < 100))


def test_clean_pair_converges_immediately(oracle):
    rng = np.random.default_rng(1)
    s = rng.uniform(-50, 50, (100, 2)).astype(np.float32)
    c, sn = np.cos(0.1), np.sin(0.1)
    d = (s.astype(np.float64) @ np.array([[c, sn], [-sn, c]]) + [1.0, -2.0]).astype(np.float32)
    r = oracle.orora_register_batch(s, d, np.array([0, 100]))[0]
    assert r["iterations"] == 1                       # mu <= 0 on the first pass: all TIMs inside the bound
    assert abs(r["yaw"] - 0.1) < 1e-5 and abs(r["x"] - 1.0) < 1e-3 and abs(r["y"] + 2.0) < 1e-3
    assert r["rot_inliers"] == 100 and r["trans_inliers"] == 100


def test_degenerate(oracle):
    s = np.zeros((1, 2), dtype=np.float32)
    r = oracle.orora_register_batch(s, s, np.array([0, 1]))[0]
    assert r["status"] == 1 and (r["x"], r["y"], r["yaw"]) == (0.0, 0.0, 0.0)


def test_scalar_tls_matches_bruteforce(oracle):
    rng = np.random.default_rng(3)
    for _ in range(20):
        n = int(rng.integers(5, 60))
        x = np.concatenate([rng.normal(2.0, 0.1, n), rng.uniform(-30, 30, n // 2)])
        beta = rng.uniform(0.2, 1.5, x.size)
        est, n_in = oracle.orora_scalar_tls(x, beta)
        # brute force: candidate consensus sets are those of every endpoint position
        best = (np.inf, None)
        ends = sorted([(xi - bi, i + 1) for i, (xi, bi) in enumerate(zip(x, beta))] +
                      [(xi + bi, -i - 1) for i, (xi, bi) in enumerate(zip(x, beta))])
        active = set()
        for v, sid in ends:
            (active.add if sid > 0 else active.discard)(abs(sid) - 1)
            if not active:
                continue
            idx = np.array(sorted(active))
            w = 1.0 / beta[idx] ** 2
            xh = (w * x[idx]).sum() / w.sum()
            cost = (w * (x[idx] - xh) ** 2).sum() + (x.size - idx.size)
            if cost < best[0] - 1e-12:
                best = (cost, xh)
        assert abs(est - best[1]) < 1e-9
        assert n_in == int((np.abs(x - est) <= beta).sum())


def test_modelling_switches(oracle):
    """flags: 1 = TIMs on the complete graph, 2 = TEASER++'s form of the scalar TLS cost.  Both still recover the
    planted motion; the complete graph sees K (K-1) / 2 TIMs (rot_inliers counts them)."""
    import numpy as np
    from navtech_radar_slam_amd import synth
    src, dst, off, truth = synth.orora_pairs(5, 6, k_range=(60, 200))
    base = oracle.orora_register_batch(src, dst, off)
    for flags in (1, 2, 3):
        p = oracle.orora_default_params()
        p.flags = flags
        r = oracle.orora_register_batch(src, dst, off, p)
        assert np.abs(r["yaw"] - truth[:, 2]).max() < 3e-3 and np.abs(r["x"] - truth[:, 0]).max() < 0.08
        if flags & 1:
            k = np.diff(off)
            assert np.all(r["rot_inliers"] > base["rot_inliers"]) and np.all(r["rot_inliers"] <= k * (k - 1) // 2)
    # equal bounds: both cost forms pick the same consensus set
    x = np.array([0.0, 0.05, -0.04, 0.02, 3.0, -2.5])
    b = np.full(6, 0.2)
    import ctypes as C
    L = oracle.lib()
    L.ororaref_scalar_tls_mode.restype = C.c_double
    L.ororaref_scalar_tls_mode.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    e0 = L.ororaref_scalar_tls_mode(x.ctypes.data, b.ctypes.data, 6, 0, None)
    e1 = L.ororaref_scalar_tls_mode(x.ctypes.data, b.ctypes.data, 6, 1, None)
    assert abs(e0 - x[:4].mean()) < 1e-12 and abs(e1 - x[:4].mean()) < 1e-12
