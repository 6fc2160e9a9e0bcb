"""GPU parity of the ORORA front end (frontend.hip through the C-ABI) against oracle/frontend_ref.c: Cartesian image
bit-exact (same fp32 operations in the same order, transcendental-free on the device), descriptors and validity
byte-identical, matches and both distances identical."""
import numpy as np
import pytest

from navtech_radar_slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fe():
    from navtech_radar_slam_amd import _rsx, frontend
    assert _rsx.device_count() >= 1
    return frontend


@pytest.mark.parametrize("W,res,flags", [(964, 0.2592, 0), (501, 0.5, 0), (964, 0.2592, 1), (70, 3.0, 0)])
def test_frontend_bit_exact(fe, oracle, W, res, flags):
    """flags = 1 (RSX_FRONTEND_THREE_PASS): remap / blur rows / blur columns as separate kernels; default: the fused tile
    kernel.  Both must give the oracle's images (W = 70: tiles that hang over the image edge, reflected halos everywhere)."""
    rows, cols = 400, 3360
    p = fe.default_params()
    p.cart_pixel_width, p.cart_resolution, p.flags = W, res, flags
    g = fe.Frontend(rows, cols, params=p)
    o = oracle.FrontendRef(rows, cols, W, res)
    descs = []
    for seed, shift in ((3, 0), (3, 21), (4, 0)):
        img, az, centres = synth.polar_image(seed, n_targets=700, noise_seed=50 + shift, shift_rows=shift)
        cart_g = g.cartesian(img, az, synth.RADAR_RESOLUTION)
        cart_o = o.cartesian(img, az, synth.RADAR_RESOLUTION)
        assert np.array_equal(cart_g, cart_o)
        rng = np.random.default_rng(seed)
        a, r = centres[:, 0], centres[:, 1]
        rr = (r + 0.5) * synth.RADAR_RESOLUTION
        xy = np.stack([rr * np.cos(az[a]), rr * np.sin(az[a])], axis=1).astype(np.float32)
        xy = np.concatenate([xy, rng.uniform(-140, 140, (300, 2)).astype(np.float32)])   # incl. points near / beyond the border
        dg, vg = g.describe(xy)
        do, vo = o.describe(xy)
        assert np.array_equal(vg, vo) and np.array_equal(dg, do)
        assert 0 < vg.sum() < len(vg)
        descs.append((dg, vg))
    for (q, qv), (t, tv) in ((descs[0], descs[1]), (descs[1], descs[2]), (descs[2], (descs[0][0][:0], descs[0][1][:0]))):
        for ratio in (0.8, 1.0, 0.5):
            got = g.match(q, qv, t, tv, ratio)
            want = o.match(q, qv, t, tv, ratio)
            for a_, b_ in zip(got, want):
                assert np.array_equal(a_, b_)
    idx, _, _ = g.match(descs[0][0], descs[0][1], descs[1][0], descs[1][1], 0.8)
    assert W != 964 or (idx >= 0).sum() > 20                                                       # the rotated scene is recognised
    g.close()
