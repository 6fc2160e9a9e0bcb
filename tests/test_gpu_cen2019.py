"""GPU parity of cen2019 keypoint extraction (cen2019.hip through the C-ABI) against the oracle:
keypoint indices bit-exact (integer/index work); Cartesian points within 1e-5 relative (cosf/sinf)."""
import numpy as np
import pytest

from navtech_radar_slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cen():
    from navtech_radar_slam_amd import cen2019
    return cen2019


@pytest.mark.parametrize("seed,max_points", [(1, 10000), (2, 10000), (3, 1500), (4, 100000)])
def test_mulran_shape_bit_exact(cen, oracle, seed, max_points):
    img, az, _ = synth.polar_image(seed, n_targets=800 + 200 * seed)
    ex = cen.Cen2019(400, 3360)
    got, xy = ex.extract(img, max_points=max_points, azimuths=az, resolution=synth.RADAR_RESOLUTION)
    want = oracle.cen2019_extract(img, max_points=max_points)
    assert got.shape == want.shape and np.array_equal(got, want)
    wxy = oracle.cen2019_to_cartesian(want, az, synth.RADAR_RESOLUTION)
    assert np.allclose(xy, wxy, rtol=1e-5, atol=1e-4)
    # idempotent: same handle, same image, same bytes
    got2 = ex.extract(img, max_points=max_points)
    assert np.array_equal(got2, got)


def test_small_and_degenerate_images(cen, oracle):
    rng = np.random.default_rng(7)
    for rows, cols in [(8, 64), (16, 128), (5, 70), (64, 300)]:
        ex = cen.Cen2019(rows, cols)
        for trial in range(4):
            img = rng.integers(0, 80, (rows, cols)).astype(np.uint8)
            for _ in range(rows):
                a, r = int(rng.integers(0, rows)), int(rng.integers(2, cols - 2))
                img[a, r - 1:r + 2] = rng.integers(150, 255, 3)
                img[(a + 1) % rows, r - 1:r + 2] = rng.integers(150, 255, 3)
            for mp in (0, 3, 10000):
                got = ex.extract(img, col_offset=0, min_range=trial, max_points=mp)
                want = oracle.cen2019_extract(img, col_offset=0, min_range=trial, max_points=mp)
                assert np.array_equal(got, want), (rows, cols, trial, mp)
        const = np.full((rows, cols), 17, dtype=np.uint8)
        assert len(ex.extract(const, col_offset=0, min_range=0)) == 0
        sat = np.full((rows, cols), 255, dtype=np.uint8)
        assert len(ex.extract(sat, col_offset=0, min_range=0)) == 0


def test_rotated_scan_gives_rotated_keypoints(cen):
    # size-independent property: rolling the image in azimuth rolls the keypoints (budget not binding)
    img, az, _ = synth.polar_image(11, n_targets=300)
    img2, _, _ = synth.polar_image(11, n_targets=300, shift_rows=37)
    ex = cen.Cen2019(400, 3360)
    a = ex.extract(img, max_points=10**6)
    b = ex.extract(img2, max_points=10**6)
    a_rot = np.stack([(a[:, 0] + 37) % 400, a[:, 1]], axis=1)
    a_rot = a_rot[np.lexsort((a_rot[:, 1], a_rot[:, 0]))]
    assert np.array_equal(a_rot, b)


def test_batch_equals_single_and_oracle(cen, oracle):
    """rsx_cen2019_extract_batch: several scans in one chain of launches; every image bit-identical to the oracle and to
    the single-image entry, different images in one batch do not interact (each has its own budget / selection)."""
    imgs = np.stack([synth.polar_image(20 + i, n_targets=500 + 300 * i)[0] for i in range(5)])
    az = synth.polar_image(20)[1]
    ex = cen.Cen2019(400, 3360)
    for mp in (10000, 700):
        tg, xy = ex.extract_batch(imgs, max_points=mp, azimuths=az, resolution=synth.RADAR_RESOLUTION)
        for i in range(len(imgs)):
            want = oracle.cen2019_extract(imgs[i], max_points=mp)
            assert np.array_equal(tg[i], want), (mp, i, len(tg[i]), len(want))
            one, one_xy = ex.extract(imgs[i], max_points=mp, azimuths=az, resolution=synth.RADAR_RESOLUTION)
            assert np.array_equal(one, want) and np.array_equal(one_xy, xy[i])


def test_plateaus_and_ties(cen, oracle):
    """Saturated plateaus give thousands of region openers with IDENTICAL h: the budget cut then falls inside one
    histogram bin and is decided by pixel index alone (the radix select over the tie list)."""
    rng = np.random.default_rng(5)
    img, _, _ = synth.polar_image(9, n_targets=400)
    for _ in range(60):
        a, r = int(rng.integers(0, 398)), int(rng.integers(100, 3300))
        img[a:a + 3, 11 + r:11 + r + int(rng.integers(5, 60))] = 255
    ex = cen.Cen2019(400, 3360)
    for mp in (50, 400, 2000, 10000):
        got = ex.extract(img, max_points=mp)
        want, dbg = oracle.cen2019_extract(img, max_points=mp, debug=True)
        assert np.array_equal(got, want), (mp, len(got), len(want), dbg["jstar"])
    # two-level images: every pixel is one of two values, h has three distinct values in all
    two = np.where(rng.uniform(size=(64, 500)) < 0.2, 200, 20).astype(np.uint8)
    ex2 = cen.Cen2019(64, 500)
    for mp in (1, 10, 300, 10000):
        for mr in (0, 7):
            assert np.array_equal(ex2.extract(two, col_offset=0, max_points=mp, min_range=mr),
                                  oracle.cen2019_extract(two, col_offset=0, max_points=mp, min_range=mr)), (mp, mr)


def test_wide_rows(cen, oracle):
    """cols > 4096 take the 1024-thread instantiation of the row kernels."""
    rng = np.random.default_rng(12)
    img = rng.gamma(2.0, 12.0, size=(12, 9000)).clip(0, 255).astype(np.uint8)
    for _ in range(40):
        a, r = int(rng.integers(0, 11)), int(rng.integers(10, 8900))
        img[a:a + 2, r:r + 4] = rng.integers(150, 255, (2, 4))
    ex = cen.Cen2019(12, 9000)
    for mp in (25, 10000):
        assert np.array_equal(ex.extract(img, col_offset=0, max_points=mp, min_range=3),
                              oracle.cen2019_extract(img, col_offset=0, max_points=mp, min_range=3)), mp


def test_large_batch_of_an_odd_shape(cen, oracle):
    """Batches of rows x images >= 8192 take the several-azimuths-per-block forms of the row kernels (8 per block in cen_stats /
    cen_collect, 4 in cen_hist, 2 in cen_runs).  53 azimuths (not a multiple of 8, 4 or 2: the last block of every kernel runs
    past the image), 1003 range bins (the last thread of a row holds 3 pixels, the row base is misaligned differently in every
    row), 160 images, two of them constant: every image bit-identical to the oracle."""
    rng = np.random.default_rng(31)
    rows, cols, nb = 53, 1003, 160
    imgs = rng.gamma(2.0, 14.0, size=(nb, rows, cols)).clip(0, 255).astype(np.uint8)
    for b in range(nb):
        for _ in range(30):
            a, r = int(rng.integers(0, rows)), int(rng.integers(3, cols - 6))
            imgs[b, a, r:r + 3] = rng.integers(140, 255, 3)
            imgs[b, (a + 1) % rows, r:r + 3] = rng.integers(140, 255, 3)
    imgs[7] = 33
    imgs[100] = 255
    imgs[5, :, 0] = 255          # the largest gradient next to the first pixel of every row
    imgs[6, :, cols - 1] = 255   # ... and at the last one (both have d = 0 there by the reflect rule)
    ex = cen.Cen2019(rows, cols)
    assert rows * nb >= 8192
    for mp in (10000, 40):
        tg = ex.extract_batch(imgs, col_offset=0, max_points=mp, min_range=2)
        for b in range(nb):
            want = oracle.cen2019_extract(imgs[b], col_offset=0, max_points=mp, min_range=2)
            assert np.array_equal(tg[b], want), (mp, b, len(tg[b]), len(want))


def test_wavefront_per_azimuth_forms(cen, oracle):
    """Batches of rows x images >= 8192 with rows of <= 4096 bins take cen_hist_wave / cen_runs_wave (a wavefront per azimuth, the
    row in chunks of 512 bins, scan totals carried between the chunks, the candidate threads of a row compacted into one pass).
    What those kernels can get wrong that the block forms cannot: runs and maxima that cross a chunk boundary, a run cut at
    min_range whose arg-max lies in a thread without a hit, more than 64 candidate threads in a row (several passes of the
    compacted evaluation), saturated plateaus (thousands of tied hits: every thread of a chunk a candidate), long dark
    stretches flagged by one bright pixel at their end, rows whose last chunk holds a few bins only, 4096-bin rows (eight
    full chunks, no walls) -- every image bit-identical to the oracle, at two budgets and three min_range values."""
    rng = np.random.default_rng(77)
    for rows, cols, nb in ((64, 3360, 128), (33, 4096, 256), (41, 520, 200), (128, 1537, 64)):
        assert rows * nb >= 8192
        imgs = rng.gamma(2.0, 9.0, size=(nb, rows, cols)).clip(0, 255).astype(np.uint8)
        for b in range(nb):
            kind = b % 8
            if kind == 0:    # bright blobs astride every chunk boundary
                for c in range(512, cols - 8, 512):
                    a = int(rng.integers(0, rows))
                    imgs[b, a, c - 3:c + 4] = rng.integers(150, 255, 7)
                    imgs[b, (a + 1) % rows, c - 2:c + 3] = rng.integers(150, 255, 5)
            elif kind == 1:  # saturated plateaus: tied hits, whole chunks of candidates
                for _ in range(6):
                    a, r = int(rng.integers(0, rows - 2)), int(rng.integers(0, max(1, cols - 700)))
                    imgs[b, a:a + 2, r:r + int(rng.integers(80, 700))] = 255
            elif kind == 2:  # dark stretches (below the mean) ended by one bright pixel; the first stretch spans min_range
                imgs[b, :, : min(cols, 900)] = rng.integers(0, 3, (rows, min(cols, 900)))
                for a in range(0, rows, 3):
                    imgs[b, a, 1] = 250
                    imgs[b, (a + 1) % rows, 2] = 240
                    imgs[b, a, min(cols, 900) - 1] = 255
            elif kind == 3:  # two-level image: three distinct h in all
                imgs[b] = np.where(rng.uniform(size=(rows, cols)) < 0.15, 210, 25)
            elif kind == 4:  # many isolated bright pixels: far more than 64 candidate threads in a row
                m = rng.uniform(size=(rows, cols)) < 0.04
                imgs[b][m] = rng.integers(120, 255, int(m.sum()))
            elif kind == 5:  # constant
                imgs[b] = 17 + b % 3
            # 6, 7: the speckle as it is
        ex = cen.Cen2019(rows, cols)
        for mp, mr in ((10000, 58), (60, 0), (10000, 3), (100000, 58)):
            mr = min(mr, cols // 4)
            tg = ex.extract_batch(imgs, col_offset=0, max_points=mp, min_range=mr)
            for b in range(nb):
                if b % 8 in (6, 7) and b >= 32:
                    continue  # (the plain speckle images: four of each are enough for the oracle's time)
                want = oracle.cen2019_extract(imgs[b], col_offset=0, max_points=mp, min_range=mr)
                assert np.array_equal(tg[b], want), (rows, cols, mp, mr, b, b % 8, len(tg[b]), len(want))
                if b < 8:  # the same images one by one: the workgroup-per-azimuth forms with their light path (single scans)
                    assert np.array_equal(ex.extract(imgs[b], col_offset=0, max_points=mp, min_range=mr), want), (rows, cols, mp, mr, b, "single")
