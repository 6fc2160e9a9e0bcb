"""GPU parity of the max-clique inlier selection (csrc/pmc.hip through the C-ABI) against the CPU oracle (oracle/pmc_ref.c):
the selection is integer work -- membership flags, clique size, largest core number, seeds and flags are compared EXACTLY --
and the solver behind it within the pose tolerance of tests/test_gpu_orora.py."""
import numpy as np
import pytest

from navtech_radar_slam_amd import synth

pytestmark = pytest.mark.gpu
TAU = 1.5


@pytest.fixture(scope="module")
def reg():
    from navtech_radar_slam_amd import orora, _rsx
    assert _rsx.device_count() >= 1
    return orora.Orora()


def _same_selection(got_m, got_i, want_m, want_i):
    for f in ("size", "max_core", "seeds", "flags"):
        assert np.array_equal(got_i[f], want_i[f]), (f, np.flatnonzero(got_i[f] != want_i[f])[:5])
    assert np.array_equal(got_m, want_m)


def test_selection_equals_oracle_on_bench_pairs(reg, oracle):
    src, dst, off, _ = synth.orora_pairs(777, 200)
    m, info = reg.max_clique_batch(src, dst, off)
    wm, winfo = oracle.pmc_select_batch(src, dst, off, TAU, nthreads=8)
    _same_selection(m, info, wm, winfo)
    assert (info["size"] > 0.25 * np.diff(off)).all() and (info["size"] <= info["max_core"] + 1).all()


def test_selection_edge_cases(reg, oracle):
    """K = 0, 1 (pass-through), 2, 3, 63, 64, 65, 2047, 2048 (the largest pruned pair), 2049 (pass-through); identical points
    (one clique, proven by the core bound); no consistent pair at all; NaN coordinates (no edges); a graph of two cliques."""
    rng = np.random.default_rng(11)
    src, dst, off = [], [], [0]

    def add(s, d):
        src.append(np.asarray(s, dtype=np.float32).reshape(-1, 2))
        dst.append(np.asarray(d, dtype=np.float32).reshape(-1, 2))
        off.append(off[-1] + len(src[-1]))

    for k in (0, 1, 2, 3, 63, 64, 65, 2047, 2048, 2049):
        s = rng.uniform(-100, 100, (k, 2))
        d = s + [1.0, -2.0] + rng.normal(0, 0.2, (k, 2))
        if k > 8:
            d[::3] = rng.uniform(-100, 100, (len(d[::3]), 2))
        add(s, d)
    add(np.zeros((40, 2)), np.zeros((40, 2)))
    add([[0, 0], [10, 0], [20, 0], [30, 0]], [[0, 0], [50, 0], [150, 0], [300, 0]])
    s = rng.uniform(-50, 50, (30, 2)); d = s.copy(); d[5] = np.nan; s[9] = np.nan
    add(s, d)
    # two groups moving differently: 50 points by (+3, 0), 35 points by (0, -40): two cliques, the larger one wins
    s = rng.uniform(-60, 60, (85, 2)); d = s.copy(); d[:50] += [3, 0]; d[50:] += [0, -40]
    add(s, d)
    src = np.concatenate(src); dst = np.concatenate(dst); off = np.array(off, dtype=np.int64)
    m, info = reg.max_clique_batch(src, dst, off)
    wm, winfo = oracle.pmc_select_batch(src, dst, off, TAU, nthreads=8)
    _same_selection(m, info, wm, winfo)
    assert info["flags"][0] == 2 and info["flags"][1] == 2 and info["flags"][9] == 2 and info["size"][9] == 2049
    assert info["size"][10] == 40 and info["flags"][10] == 1
    assert info["size"][11] == 1
    assert info["size"][13] == 50 and m[off[13]:off[13] + 50].all() and not m[off[13] + 50:off[14]].any()


def test_selection_other_bounds(reg, oracle):
    from navtech_radar_slam_amd import orora
    src, dst, off, _ = synth.orora_pairs(5, 40, k_range=(50, 700))
    for tau in (0.3, 0.75, 4.0):
        p = orora.default_params()
        p.tim_noise_bound = tau
        m, info = reg.max_clique_batch(src, dst, off, p)
        wm, winfo = oracle.pmc_select_batch(src, dst, off, tau, nthreads=8)
        _same_selection(m, info, wm, winfo)


def test_solver_behind_the_selection(reg, oracle):
    """RSX_ORORA_PMC on the registration entry = the oracle's selection followed by the oracle's solver on the selected matches
    (in their original order); small pairs on-chip, one pair above 2048 matches passing through to the HBM-workspace kernel."""
    from navtech_radar_slam_amd import orora, _rsx
    src, dst, off, truth = synth.orora_pairs(778, 120)
    big = synth.orora_pairs(92, 1, k_range=(2300, 2300))
    src = np.concatenate([src, big[0]]); dst = np.concatenate([dst, big[1]])
    off = np.concatenate([off, [off[-1] + 2300]]); truth = np.concatenate([truth, big[3]])
    p = orora.default_params()
    p.flags |= _rsx.ORORA_PMC
    got = reg.register_batch(src, dst, off, p)
    info = reg.last_pmc_info(len(off) - 1)
    wm, winfo = oracle.pmc_select_batch(src, dst, off, TAU, nthreads=8)
    for f in ("size", "max_core", "seeds", "flags"):
        assert np.array_equal(info[f], winfo[f]), f
    s2, d2, o2 = oracle.pmc_compact(src, dst, off, wm)
    want = oracle.orora_register_batch(s2, d2, o2, nthreads=8)
    assert np.array_equal(got["status"], want["status"])
    for f in ("x", "y", "yaw"):
        assert np.abs(got[f] - want[f]).max() < 1e-4, f
    assert np.array_equal(got["iterations"], want["iterations"]) and np.array_equal(got["rot_inliers"], want["rot_inliers"])
    assert np.abs(got["x"] - truth[:, 0]).max() < 0.05 and np.abs(got["yaw"] - truth[:, 2]).max() < 2e-3
    # and it is no worse than the unpruned solver on these data
    plain = reg.register_batch(src, dst, off)
    assert np.abs(got["x"] - truth[:, 0]).max() <= np.abs(plain["x"] - truth[:, 0]).max() + 0.02


def test_device_entry_needs_a_reservation_and_survives_too_small_a_one(reg, oracle):
    import torch
    from navtech_radar_slam_amd import orora, _rsx
    src, dst, off, _ = synth.orora_pairs(779, 12, k_range=(300, 400))
    fresh = orora.Orora()
    p = orora.default_params()
    p.flags |= _rsx.ORORA_PMC
    d_src, d_dst, d_off = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda(), torch.from_numpy(off).cuda()
    d_res = torch.zeros((12, 5), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    with pytest.raises(_rsx.RsxError):
        fresh.register_batch_device(d_src.data_ptr(), d_dst.data_ptr(), d_off.data_ptr(), 12, d_res.data_ptr(), p, stream=st)
    fresh.reserve(int(off[1]))   # room for the first pair only (<= 512 matches after rounding up to the allocation granule)
    fresh.register_batch_device(d_src.data_ptr(), d_dst.data_ptr(), d_off.data_ptr(), 12, d_res.data_ptr(), p, stream=st)
    torch.cuda.synchronize()
    info = fresh.last_pmc_info(12)
    assert info["flags"][0] & 2 == 0
    late = info["flags"] & 4 != 0
    assert late[-1] and (info["flags"][late] & 2 != 0).all() and (info["size"][late] == np.diff(off)[late]).all()
    res = d_res.cpu().numpy().view(_rsx.ORORA_RESULT_DTYPE).reshape(12)
    assert (res["status"] == 0).all()
    fresh.close()


def test_edges_at_the_bound(reg, oracle):
    """The build decides a pair of matches in fp32 where fp32 can and falls back to the oracle's fp64 form otherwise
    (csrc/pmc.hip edge_f32): thousands of two-match pairs whose distance difference lies within 0 .. 1e-2 of the bound on
    either side (K = 2: the selection keeps both matches iff they are consistent), exact equality included (strict <: no
    edge), duplicates (A = 0 or B = 0), far-away points, a tiny and a huge bound; and lines of 64 matches whose neighbours
    sit exactly on / just inside the bound."""
    from navtech_radar_slam_amd import orora
    rng = np.random.default_rng(21)
    src, dst, off = [], [], [0]

    def add(s, d):
        src.append(np.asarray(s, dtype=np.float64).reshape(-1, 2))
        dst.append(np.asarray(d, dtype=np.float64).reshape(-1, 2))
        off.append(off[-1] + len(src[-1]))

    for eps in (0.0, 1e-9, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2):
        for sign in (-1.0, 1.0):
            for _ in range(250):
                r = rng.uniform(0.05, 400.0) if rng.random() < 0.8 else rng.uniform(0.0, 2.0)
                grow = rng.choice([-1.0, 1.0])
                r2 = max(0.0, r + grow * (TAU + sign * eps))
                phi, psi = rng.uniform(0, 2 * np.pi, 2)
                o1, o2 = rng.uniform(-200, 200, 2), rng.uniform(-200, 200, 2)
                add([o1, o1 + r * np.array([np.cos(phi), np.sin(phi)])], [o2, o2 + r2 * np.array([np.cos(psi), np.sin(psi)])])
    for a, b in ((3.0, 4.5), (3.0, 1.5), (3.0, 4.499999), (3.0, 4.500001), (0.0, 1.5), (0.0, 1.4999999), (1e4, 1e4 + 1.5), (1e4, 1e4 + 1.49)):
        add([[0, 0], [a, 0]], [[0, 0], [b, 0]])          # axis-aligned, exactly representable: |da - db| == the bound -> no edge
        add([[5, 5], [5, 5 + a]], [[-7, 2], [-7 + b, 2]])
    add([[1, 1], [1, 1]], [[2, 2], [2, 3.4]])             # A = 0
    add([[1, 1], [1, 2.6]], [[2, 2], [2, 2]])             # B = 0
    i = np.arange(64, dtype=np.float64)
    add(np.stack([i, 0 * i], 1), np.stack([2.5 * i, 0 * i], 1))             # neighbours exactly on the bound: no edges at all
    add(np.stack([i, 0 * i], 1), np.stack([2.4999998 * i, 0 * i], 1))       # ... just inside: a path
    add(np.stack([0 * i, 3 * i], 1), np.stack([4.5 * i, 0 * i], 1))
    src = np.concatenate(src).astype(np.float32); dst = np.concatenate(dst).astype(np.float32); off = np.array(off, dtype=np.int64)
    for tau in (TAU, 1e-3, 37.5):
        p = orora.default_params()
        p.tim_noise_bound = tau
        m, info = reg.max_clique_batch(src, dst, off, p)
        wm, winfo = oracle.pmc_select_batch(src, dst, off, tau, nthreads=8)
        _same_selection(m, info, wm, winfo)
    two = np.diff(off) == 2
    m, info = reg.max_clique_batch(src, dst, off)
    assert 0.3 < np.mean(info["size"][two] == 2) < 0.7     # both sides of the bound are populated
