"""Cross-check of the cen2019 oracle (oracle/cen2019_ref.c) with two further restatements in numpy
(oracle/cen2019_np.py): a sequential one written from SURVEY.md App. B.2 alone, and the sort-free closed form the HIP
kernels implement.  PARITY UNPINNED w.r.t. the reference (source absent) -- this is the only cross-check available."""
import numpy as np

from navtech_radar_slam_amd import synth
from oracle import cen2019_np as cn


def _same(oracle, img, **kw):
    want, dbg = oracle.cen2019_extract(img, debug=True, **kw)
    seq, d1 = cn.extract_sequential(img, **kw)
    par, d2 = cn.extract_parallel(img, **kw)
    assert np.array_equal(seq, want) and np.array_equal(par, want), (kw, len(want), len(seq), len(par))
    assert d1["jstar"] == dbg["jstar"] == d2["jstar"] and d1["ncand"] == dbg["ncand"] == d2["ncand"]
    assert np.float32(d1["mean_h"]) == np.float32(dbg["mean_h"])


def test_small_images_all_three_agree(oracle):
    rng = np.random.default_rng(7)
    for rows, cols in [(8, 64), (16, 128), (5, 70), (64, 300)]:
        for trial in range(6):
            img = rng.integers(0, 80, (rows, cols)).astype(np.uint8)
            for _ in range(rows):
                a, r = int(rng.integers(0, rows)), int(rng.integers(2, cols - 2))
                img[a, r - 1:r + 2] = rng.integers(150, 255, 3)
                img[(a + 1) % rows, r - 1:r + 2] = rng.integers(150, 255, 3)
            if trial == 5:
                img[:, ::3] = 255          # saturated columns: thousands of equal h, order decided by pixel index
            for mp in (0, 3, 17, 10000):
                _same(oracle, img, col_offset=0, min_range=trial, max_points=mp)


def test_two_level_image(oracle):
    rng = np.random.default_rng(5)
    two = np.where(rng.uniform(size=(32, 200)) < 0.2, 200, 20).astype(np.uint8)
    for mp in (1, 10, 300, 10000):
        _same(oracle, two, col_offset=0, min_range=2, max_points=mp)


def test_mulran_shape_closed_form(oracle):
    """400 x 3360: the closed form against the C oracle (the literal numpy walk takes a minute at this size: one budget)."""
    img, _, _ = synth.polar_image(3, n_targets=1400)
    for mp in (1500, 10000):
        want, dbg = oracle.cen2019_extract(img, max_points=mp, debug=True)
        par, d2 = cn.extract_parallel(img, max_points=mp)
        assert np.array_equal(par, want) and d2["jstar"] == dbg["jstar"] and d2["ncand"] == dbg["ncand"]
    seq, d1 = cn.extract_sequential(img, max_points=1500)
    assert np.array_equal(seq, oracle.cen2019_extract(img, max_points=1500))
