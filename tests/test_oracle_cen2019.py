"""Pins the cen2019 oracle (oracle/cen2019_ref.c).  PARITY UNPINNED w.r.t. the reference (source
absent); these tests check the restated method against its own definition on small images."""
import numpy as np

from navtech_radar_slam_amd import synth


def _img(rows, cols, fill=10):
    return np.full((rows, cols), fill, dtype=np.uint8)


def test_constant_image_has_no_keypoints(oracle):
    t, d = oracle.cen2019_extract(_img(8, 64), col_offset=0, min_range=0, debug=True)
    assert len(t) == 0 and d["ncand"] == 0 and d["mean_h"] == 0.0


def test_single_blob_needs_adjacent_azimuth(oracle):
    # a small region budget keeps the (nearly empty) background out: on a flat image every
    # s < 0 background run would otherwise be swallowed by the first weak candidate next to it
    img = _img(8, 64)
    img[3, 30:33] = [120, 200, 120]
    t = oracle.cen2019_extract(img, col_offset=0, min_range=0, max_points=1)
    assert len(t) == 0                               # nothing marked on azimuth 2 or 4
    img[4, 30:33] = [100, 180, 100]
    t = oracle.cen2019_extract(img, col_offset=0, min_range=0, max_points=2)
    assert t.tolist() == [[3, 31], [4, 31]]          # row-major order, argmax of h inside each run
    t = oracle.cen2019_extract(img, col_offset=0, min_range=40, max_points=2)
    assert len(t) == 0                               # min_range cuts them off
    img[0, 9:12] = [110, 250, 110]
    img[7, 9:12] = [105, 240, 105]                   # azimuth wrap-around: rows 0 and 7 are neighbours
    t = oracle.cen2019_extract(img, col_offset=0, min_range=0, max_points=4)
    assert t.tolist() == [[0, 10], [3, 31], [4, 31], [7, 10]]
    # with an unlimited budget the flat background is marked end to end: runs that reach the last
    # bin are never closed and yield nothing (documented behaviour of the method)
    assert len(oracle.cen2019_extract(img, col_offset=0, min_range=0)) == 0


def test_budget_and_order(oracle):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 60, (16, 128)).astype(np.uint8)
    for a in range(0, 16, 2):                        # 8 bright two-azimuth targets, decreasing strength
        img[a, 40 + a] = 250 - 10 * a
        img[a + 1, 40 + a] = 245 - 10 * a
    full, d = oracle.cen2019_extract(img, col_offset=0, min_range=0, max_points=10000, debug=True)
    assert d["jstar"] == d["ncand"]                  # budget never reached: every candidate visited
    few, d2 = oracle.cen2019_extract(img, col_offset=0, min_range=0, max_points=4, debug=True)
    assert 4 <= d2["jstar"] < d["ncand"] and len(few) == 4 and len(full) <= 16 * 64
    assert [0, 40] in few.tolist() and [1, 40] in few.tolist()   # the strongest regions come first
    none = oracle.cen2019_extract(img, col_offset=0, min_range=0, max_points=0)
    assert len(none) == 0


def test_oxford_form_image(oracle):
    img, az, centres = synth.polar_image(1, n_targets=600)
    t, d = oracle.cen2019_extract(img, debug=True)
    assert 500 < len(t) < 20000 and d["jstar"] >= 10000
    assert np.all(t[:, 1] >= 58) and np.all(np.diff(t[:, 0]) >= 0)
    hit = 0
    for a, r in centres[:200]:
        near = (np.abs(((t[:, 0] - a + 200) % 400) - 200) <= 2) & (np.abs(t[:, 1] - r) <= 8)
        hit += bool(near.any())
    assert hit > 100
    xy = oracle.cen2019_to_cartesian(t, az, synth.RADAR_RESOLUTION)
    rng_m = np.hypot(xy[:, 0], xy[:, 1])
    assert np.allclose(rng_m, (t[:, 1] + 0.5) * synth.RADAR_RESOLUTION, rtol=1e-5)
