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


@pytest.mark.parametrize("W,res,flags", [(964, 0.2592, 0), (501, 0.5, 0), (964, 0.2592, 1), (70, 3.0, 0), (964, 0.2592, 2), (501, 0.5, 4),
                                         (70, 3.0, 2), (45, 6.0, 0)])
def test_frontend_bit_exact(fe, oracle, W, res, flags):
    """flags = 1 (RSX_FRONTEND_THREE_PASS): remap / blur rows / blur columns as separate kernels; 4 (RSX_FRONTEND_TILES): the
    32 x 32 LDS-tile kernel; default: the strip kernel (one wavefront per 58 columns, azimuth rows by reciprocal multiply where
    the margin decides); 2 (RSX_FRONTEND_EXACT_AZIMUTH): the strip kernel with EVERY azimuth row through its fp64-division
    branch.  All must give the oracle's images (W = 70 / 45: strips and tiles that hang over the image edge, reflected halos
    everywhere, one or two strips, segments shorter than the image)."""
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
        cart_r, blur_r = g.read_images(0)
        assert np.array_equal(cart_r, cart_o) and np.array_equal(blur_r, o.blur)   # the smoothed copy itself, not only through descriptors
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


def test_strip_kernel_batch_with_per_image_grids(fe, oracle):
    """A batch resident in HBM, every image with its OWN azimuth grid (offset first azimuth, a slightly different step, one grid
    whose first azimuth is a pixel's angle exactly, one that starts half a turn round): image by image the oracle's Cartesian
    image and smoothed copy, with 5 images (the four wavefronts of a block straddle images and row segments) and with the
    azimuth screen forced through its exact branch."""
    import torch
    rows, cols, W, res = 400, 3360, 964, 0.2592
    imgs, azs = [], []
    for k in range(5):
        img, az, _ = synth.polar_image(20 + k, n_targets=500)
        az = az.astype(np.float64)
        if k == 1:
            az = az + 0.0123
        if k == 2:
            az = az * (1.0 + 3e-4) + 0.5 * (az[1] - az[0])
        if k == 3:
            az = az + float(np.arctan2(1.0, 3.0))                    # a pixel direction of the even-width grid, up to rounding
        if k == 4:
            az = az + np.pi
        imgs.append(img)
        azs.append(az.astype(np.float32))
    batch = np.ascontiguousarray(np.stack(imgs))
    az_all = np.ascontiguousarray(np.stack(azs))
    d_img = torch.from_numpy(batch).cuda()
    d_az = torch.from_numpy(az_all).cuda()
    for flags in (0, 2, 4):
        p = fe.default_params()
        p.cart_pixel_width, p.cart_resolution, p.flags = W, res, flags
        g = fe.Frontend(rows, cols, params=p)
        g.cartesian_batch_device(d_img.data_ptr(), 5, batch.shape[1] * batch.shape[2], batch.shape[2], d_az.data_ptr(), rows,
                                 synth.RADAR_RESOLUTION)
        for k in range(5):
            o = oracle.FrontendRef(rows, cols, W, res)
            cart_o = o.cartesian(imgs[k], azs[k], synth.RADAR_RESOLUTION)
            cart_g, blur_g = g.read_images(k)
            assert np.array_equal(cart_g, cart_o), (flags, k)
            assert np.array_equal(blur_g, o.blur), (flags, k)
        g.close()


def test_strip_kernel_equals_the_three_kernel_form_on_unusual_grids_and_shapes(fe):
    """Grids and shapes no radar delivers but a caller can pass on the device path (nothing is validated there): a descending
    grid, a step of a thousandth of a bin, a first azimuth many turns away, a zero step (NaN rows), 7 azimuths, 3 range bins,
    a padded row stride with no metadata columns.  There is no oracle for these; the strip kernel (reciprocal multiply + exact
    branch, 16-bit tap pairs, LDS transposition) must give the three-kernel form's images (exact division, byte taps) bit for bit."""
    import torch
    rng = np.random.default_rng(8)
    cases = []
    for rows, cols, stride, off, W, res in ((400, 3360, 3371, 11, 301, 0.9), (7, 3, 8, 0, 64, 0.05), (33, 100, 128, 5, 97, 0.7)):
        base = (np.arange(rows) * (2 * np.pi / rows)).astype(np.float32)
        grids = [base, base[::-1].copy(), (base * 1e-3).astype(np.float32), (base + np.float32(100.0)).astype(np.float32),
                 np.zeros(rows, np.float32), (base * 3.7 - 1.0).astype(np.float32)]
        cases.append((rows, cols, stride, off, W, res, grids))
    for rows, cols, stride, off, W, res, grids in cases:
        n = len(grids)
        imgs = rng.integers(0, 256, (n, rows, stride), dtype=np.uint8)
        d_img = torch.from_numpy(imgs).cuda()
        d_az = torch.from_numpy(np.ascontiguousarray(np.stack(grids))).cuda()
        out = {}
        for flags in (1, 0, 2):
            p = fe.default_params()
            p.cart_pixel_width, p.cart_resolution, p.flags = W, res, flags
            g = fe.Frontend(rows, cols, params=p)
            g.cartesian_batch_device(d_img.data_ptr(), n, rows * stride, stride, d_az.data_ptr(), rows, 0.3, col_offset=off)
            out[flags] = [g.read_images(k) for k in range(n)]
            g.close()
        for k in range(n):
            for flags in (0, 2):
                for a, b in zip(out[flags][k], out[1][k]):
                    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (rows, cols, W, k, flags)
        assert np.isfinite(out[0][0][0]).all() and out[0][0][0].max() > 0


def test_batched_describe_and_consecutive_matching_equal_the_oracle(fe, oracle):
    """The device-resident forms the odometry pipeline chains (rsx_frontend_describe_batch_device: one wavefront per keypoint from
    a table of the disc's pixels, lane sums by DPP, 128 blocks per image; rsx_frontend_match_consecutive_device: compacted valid
    lists, four lanes per query, packed (distance, index) minima): 6 images with 0 / 1 / 3 / 900 / 2100 / 2100 keypoints in slots
    of 2304, many of them in or beyond the image border, descriptors and validity byte for byte, every consecutive pair's matches in
    both directions against oracle.match on the same descriptors (incl. pairs with an empty side)."""
    import torch
    rows, cols, W, res = 400, 3360, 964, 0.2592
    n, K = 6, 2304
    counts = np.array([0, 1, 3, 900, 2100, 2100], dtype=np.int32)
    rng = np.random.default_rng(77)
    imgs, azs, xys = [], [], np.zeros((n, K, 2), dtype=np.float32)
    for k in range(n):
        img, az, centres = synth.polar_image(40 + (k % 3), n_targets=900, noise_seed=60 + k)     # images 3.. share scenes with 0..2
        imgs.append(img)
        azs.append(az)
        a, r = centres[:, 0], centres[:, 1]
        rr = (r + 0.5) * synth.RADAR_RESOLUTION
        pts = np.stack([rr * np.cos(az[a]), rr * np.sin(az[a])], axis=1).astype(np.float32)
        pts = np.concatenate([pts, rng.uniform(-135, 135, (K, 2)).astype(np.float32)])[:K]
        xys[k] = pts
    batch = np.ascontiguousarray(np.stack(imgs))
    d_img, d_az = torch.from_numpy(batch).cuda(), torch.from_numpy(np.ascontiguousarray(np.stack(azs))).cuda()
    d_xy, d_cnt = torch.from_numpy(xys).cuda(), torch.from_numpy(counts).cuda()
    d_desc = torch.zeros((n, K, 32), dtype=torch.uint8, device="cuda")
    d_valid = torch.zeros((n, K), dtype=torch.uint8, device="cuda")
    d_fwd = torch.full((n - 1, K), -7, dtype=torch.int32, device="cuda")
    d_bwd = torch.full((n - 1, K), -7, dtype=torch.int32, device="cuda")
    p = fe.default_params()
    p.cart_pixel_width, p.cart_resolution = W, res
    g = fe.Frontend(rows, cols, params=p)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g.cartesian_batch_device(d_img.data_ptr(), n, batch.shape[1] * batch.shape[2], batch.shape[2], d_az.data_ptr(), rows,
                                 synth.RADAR_RESOLUTION, stream=s.cuda_stream)
        g.describe_batch_device(d_xy.data_ptr(), d_cnt.data_ptr(), n, K, d_desc.data_ptr(), d_valid.data_ptr(), stream=s.cuda_stream)
        g.match_consecutive_device(d_desc.data_ptr(), d_valid.data_ptr(), d_cnt.data_ptr(), K, 0, n - 1, 0.8, d_fwd.data_ptr(),
                                   d_bwd.data_ptr(), stream=s.cuda_stream)
    s.synchronize()
    desc, valid, fwd, bwd = d_desc.cpu().numpy(), d_valid.cpu().numpy(), d_fwd.cpu().numpy(), d_bwd.cpu().numpy()
    want = []
    for k in range(n):
        o = oracle.FrontendRef(rows, cols, W, res)
        o.cartesian(imgs[k], azs[k], synth.RADAR_RESOLUTION)
        do, vo = o.describe(xys[k, :counts[k]])
        assert np.array_equal(valid[k, :counts[k]], vo) and np.array_equal(desc[k, :counts[k]], do), k
        want.append((do, vo, o))
    assert 0 < want[4][1].sum() < counts[4]
    for j in range(n - 1):
        (qa, va, o), (qb, vb, _) = want[j], want[j + 1]
        f_idx, _, _ = o.match(qa, va, qb, vb, 0.8)
        b_idx, _, _ = o.match(qb, vb, qa, va, 0.8)
        assert np.array_equal(fwd[j, :counts[j]], f_idx), j
        assert np.array_equal(bwd[j, :counts[j + 1]], b_idx), j
    assert (fwd[4, :counts[4]] >= 0).sum() > 20           # the same scene under another noise realisation is recognised
    g.close()


def test_consecutive_matching_sizes_and_ties(fe, oracle):
    """rsx_frontend_match_consecutive_device alone (the matrix-core matcher: four fp8 MFMAs per 32 x 32 tile of +-1 descriptors, a lane
    keeps one query's two smallest (distance << 20 | index) keys): list sizes around its tile and block sizes (31 / 32 / 33 / 127 /
    128 / 129 valid descriptors, none, one), more valid queries than one pass of the grid takes (4100 > 16 x 128: a block strides
    over query blocks), descriptors drawn from a small pool (exact duplicates: distance 0 several times -- the smaller train index
    wins and the ratio test fails; equal second-best distances), all-zero and all-one descriptors, invalid keypoints sprinkled in --
    every pair in both directions against oracle.match."""
    import torch
    rng = np.random.default_rng(123)
    K = 4352
    counts = np.array([4100, 33, 32, 31, 129, 128, 127, 0, 1, 4100, 700], dtype=np.int32)
    n = len(counts)
    pool = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    pool[0] = 0
    pool[1] = 255
    desc = np.zeros((n, K, 32), dtype=np.uint8)
    valid = np.zeros((n, K), dtype=np.uint8)
    for k in range(n):
        c = counts[k]
        d = rng.integers(0, 256, (c, 32), dtype=np.uint8)
        dup = rng.uniform(size=c) < 0.3
        d[dup] = pool[rng.integers(0, len(pool), int(dup.sum()))]
        near = rng.uniform(size=c) < 0.2          # a pool entry with one or two bits flipped
        if near.any():
            base = pool[rng.integers(0, len(pool), int(near.sum()))].copy()
            base[np.arange(len(base)), rng.integers(0, 32, len(base))] ^= (1 << rng.integers(0, 8, len(base))).astype(np.uint8)
            d[near] = base
        desc[k, :c] = d
        valid[k, :c] = (rng.uniform(size=c) < (0.97 if k in (0, 9) else 0.8)).astype(np.uint8) if c > 1 else 1
    d_desc, d_valid, d_cnt = torch.from_numpy(desc).cuda(), torch.from_numpy(valid).cuda(), torch.from_numpy(counts).cuda()
    d_fwd = torch.full((n - 1, K), -7, dtype=torch.int32, device="cuda")
    d_bwd = torch.full((n - 1, K), -7, dtype=torch.int32, device="cuda")
    g = fe.Frontend(400, 3360)
    g.match_consecutive_device(d_desc.data_ptr(), d_valid.data_ptr(), d_cnt.data_ptr(), K, 0, n - 1, 0.8, d_fwd.data_ptr(), d_bwd.data_ptr())
    torch.cuda.synchronize()
    fwd, bwd = d_fwd.cpu().numpy(), d_bwd.cpu().numpy()
    o = oracle.FrontendRef(400, 3360, 964, 0.2592)
    some = 0
    for j in range(n - 1):
        qa, va, qb, vb = desc[j, :counts[j]], valid[j, :counts[j]], desc[j + 1, :counts[j + 1]], valid[j + 1, :counts[j + 1]]
        f_idx, _, _ = o.match(qa, va, qb, vb, 0.8)
        b_idx, _, _ = o.match(qb, vb, qa, va, 0.8)
        assert np.array_equal(fwd[j, :counts[j]], f_idx), (j, int((fwd[j, :counts[j]] != f_idx).sum()))
        assert np.array_equal(bwd[j, :counts[j + 1]], b_idx), (j, int((bwd[j, :counts[j + 1]] != b_idx).sum()))
        some += int((f_idx >= 0).sum())
    assert some > 50
    g.close()
