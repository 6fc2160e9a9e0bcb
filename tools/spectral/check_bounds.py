#!/usr/bin/env python3
"""GPU check of the spectral filter's bounds against the fp64 all-shift minimum (and the direct filter)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from navtech_radar_slam_amd import scancontext as sc, synth, _rsx
from test_gpu_sc_filter import all_shift_bound, make_db

for binary in (True, False):
    n, nq = 1000 + 13, 20
    descs = make_db(100 + binary, n, binary)
    rng = np.random.default_rng(9)
    queries = np.stack([synth.rotate_descriptor(descs[int(rng.integers(0, n))], int(rng.integers(0, 60))) for _ in range(nq)])
    queries[::2, rng.integers(0, 1200, 20)] = 0
    gs = sc.SCManager(filter_kind=_rsx.KIND_SPECTRAL)
    gd = sc.SCManager(filter_kind=_rsx.KIND_DIRECT)
    gs.add_descriptors_f32(descs); gd.add_descriptors_f32(descs)
    eps = gs.filter_eps()
    ls = gs.filter_bounds(queries); ld = gd.filter_bounds(queries)
    dn = (np.sqrt((descs.reshape(-1, 60, 20).astype(np.float64) ** 2).sum(2)) > 0).sum(1)
    worst_viol, worst_gap_full = -1, 0
    for qi in range(nq):
        want = all_shift_bound(queries[qi], descs)
        fin = np.isfinite(want)
        qn = (np.sqrt((queries[qi].reshape(60, 20).astype(np.float64) ** 2).sum(1)) > 0).sum()
        viol = (ls[qi][fin] - eps - want[fin]).max() if fin.any() else -1
        worst_viol = max(worst_viol, viol)
        assert np.all(ls[qi][~fin] == np.inf), (qi, ls[qi][~fin][:5])
        full = fin & ((dn == 60) | (qn == 60)) & (want <= 1.0)   # S < 0 (negative heights) is clamped to 0: valid, looser
        if full.any():
            nlo = np.maximum(qn + dn - 60, 1)
            expect = want - 2.05e-3 * np.sqrt(qn * dn) / nlo + eps
            worst_gap_full = max(worst_gap_full, np.abs(ls[qi][full] - expect[full]).max())
        if qi < 3:
            print(f" q{qi}: nq={qn} spectral-direct mean {np.mean(ls[qi][fin]-ld[qi][fin]):+.5f} min {np.min(ls[qi][fin]-ld[qi][fin]):+.5f} max {np.max(ls[qi][fin]-ld[qi][fin]):+.5f}")
    print(f"binary={binary}: max(lb - eps - true) = {worst_viol:.3e} (must be <= 0); |lb - expected| on exact-n pairs = {worst_gap_full:.3e}")
