"""CPU-side checks of the drop-in boundary: librsx.so loads without a GPU, exports exactly the
symbols include/rsx.h declares, fails loudly (no CPU fallback) when no device is present, and its
pure-host logic (top-k merge) agrees with the oracle."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def rsx():
    import __graft_entry__ as ge
    from navtech_radar_slam_amd import _rsx
    if not os.path.exists(_rsx.LIB_PATH):
        ge.build()
    return _rsx


def _header_symbols():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        txt = open(os.path.join(ROOT, "include", fn)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(rsx_[a-z0-9_]+)\s*\(", txt))
    return names


def test_exports_match_header(rsx):
    declared = _header_symbols()
    assert declared == set(rsx.SYMBOLS), declared ^ set(rsx.SYMBOLS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", rsx.LIB_PATH], text=True)
    exported = set(re.findall(r" T (rsx_[a-z0-9_]+)", out))
    assert declared <= exported, declared - exported
    L = rsx.lib()
    for s in declared:
        getattr(L, s)
    assert "gfx950" in rsx.version()


def test_no_cpu_fallback(rsx):
    """Without a GPU, creating a handle must fail with RSX_ERR_NO_DEVICE -- never compute on the CPU."""
    if rsx.device_count() > 0:
        pytest.skip("a GPU is visible here")
    h = C.c_void_p()
    st = rsx.lib().rsx_sc_create(None, C.byref(h))
    assert st == -2 and not h.value
    assert b"no HIP device" in rsx.lib().rsx_last_error_string()
    from navtech_radar_slam_amd import scancontext
    with pytest.raises(rsx.RsxError):
        scancontext.SCManager()
    L = rsx.lib()
    for create in (lambda: L.rsx_orora_create(0, C.byref(h)), lambda: L.rsx_cen2019_create(0, 400, 3360, C.byref(h)),
                   lambda: L.rsx_voxelgrid_create(0, C.byref(h)), lambda: L.rsx_icp_create(0, C.byref(h))):
        assert create() == -2 and not h.value


def test_new_rows_param_defaults(rsx):
    # ICP settings of doICPVirtualRelative (laserPosegraphOptimization.cpp:374-377)
    p = rsx.IcpParams()
    assert rsx.lib().rsx_icp_default_params(C.byref(p)) == 0
    assert (p.max_corr_dist, p.max_iterations, p.transformation_epsilon, p.euclidean_fitness_epsilon) == (150.0, 100, 1e-6, 1e-6)


def test_param_defaults_match_reference(rsx):
    p = rsx.ScParams()
    assert rsx.lib().rsx_sc_default_params(C.byref(p)) == 0
    # Scancontext.h:83-104
    assert (p.lidar_height, p.max_radius, p.num_exclude_recent, p.num_candidates) == (2.0, 80.0, 30, 3)
    assert (p.search_ratio, p.dist_thres, p.tree_making_period) == (0.1, 0.2, 30)
    assert (p.shard_rank, p.shard_world) == (0, 1)


def test_host_merge_matches_oracle(rsx, oracle):
    from navtech_radar_slam_amd import scancontext, synth
    m = oracle.Manager()
    descs = synth.random_descriptors(3, 200, binary=True).astype(np.float64)
    m.add_descriptors(descs)
    G, k, nq = 8, 10, 3
    qs = [descs[5], oracle.circshift(descs[77], 13), np.zeros(1200)]
    parts = np.zeros((G, nq, k), dtype=scancontext.HIT_DTYPE)
    for qi, q in enumerate(qs):
        dist, shift = m.pair_distances(q)
        for g in range(G):
            rec = sorted((dist[i], i, shift[i]) for i in range(g, 170, G) if dist[i] < 1e7)[:k]  # n_eligible = 170
            rec += [(1e7, 0, 0)] * (k - len(rec))
            for j, (d, i, s) in enumerate(rec):
                parts[g, qi, j] = (d, i, s)
    merged = scancontext.merge_topk(parts)
    for qi, q in enumerate(qs):
        want = m.exhaustive(q, n_eligible=170, k=k)
        assert np.array_equal(merged[qi], want.astype(scancontext.HIT_DTYPE))
    # bad arguments are status codes, not crashes
    assert rsx.lib().rsx_sc_merge_topk(None, 1, 1, 1, None) == -1


def test_ringkey_tree_layout_matches_oracle(rsx, oracle):
    """The product's HOST build of the candidate-stage search tree (csrc/sc_kdtree.cpp, no device involved) against the
    oracle's restatement, which tests/test_oracle_pin.py pins to the reference's nanoflann: the same permutation of the
    keys, i.e. the same leaves in the same order with the same order inside every leaf (what decides which tied
    neighbour the detector picks)."""
    import numpy as np
    L = rsx.lib()
    rng = np.random.default_rng(3)
    for n in (1, 5, 10, 11, 40, 333, 2000, 20000):
        for kind in range(3):
            if kind == 0:      # tie-heavy: multiples of 1/30 from a small range, duplicates
                keys = rng.integers(0, 7, size=(n, 20)).astype(np.float32) * np.float32(2.0 / 60.0)
                keys[rng.integers(0, n, size=n // 3)] = keys[rng.integers(0, n, size=n // 3)]
            elif kind == 1:    # continuous
                keys = rng.uniform(0, 2, size=(n, 20)).astype(np.float32)
            else:              # every point identical: the balanced-split rule alone shapes the tree
                keys = np.full((n, 20), np.float32(0.25))
            keys = np.ascontiguousarray(keys)
            vind = np.zeros(n, dtype=np.int32)
            nn, depth = C.c_int32(0), C.c_int32(0)
            assert L.rsx_sc_ringkey_tree_layout(keys.ctypes.data, n, vind.ctypes.data, C.byref(nn), C.byref(depth)) == 0
            want = oracle.KdTree(keys).vind()
            assert np.array_equal(vind.astype(np.int64), want), (n, kind)
            assert sorted(vind.tolist()) == list(range(n)) and nn.value >= 1 and 1 <= depth.value <= 64
    assert L.rsx_sc_ringkey_tree_layout(None, 1, None, None, None) != 0


def test_exception_firewall_is_on_every_status_entry():
    """no C++ exception crosses the C-ABI.  Every `int rsx_*(...)` definition under csrc/ is a function-try-block
    closed by RSX_CATCH_ALL (rsx_common.h): static check over the sources."""
    import glob
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "navtech-radar-slam_amd", "csrc")
    n = 0
    for f in sorted(glob.glob(os.path.join(csrc, "*.cpp")) + glob.glob(os.path.join(csrc, "*.hip"))):
        lines = open(f).read().split("\n")
        for i, line in enumerate(lines):
            if not re.match(r"^int rsx_\w+\(", line):
                continue
            j = i
            while not lines[j].rstrip().endswith("{") and not lines[j].rstrip().endswith(";"):
                j += 1
            if lines[j].rstrip().endswith(";"):
                continue                        # a declaration
            assert lines[j].rstrip().endswith("try {"), f"{os.path.basename(f)}:{i + 1}: {line}"
            k = j + 1
            while not lines[k].startswith("}"):
                k += 1
            assert lines[k].startswith("} RSX_CATCH_ALL"), f"{os.path.basename(f)}:{k + 1}"
            n += 1
    assert n >= 100


def test_exceptions_thrown_inside_the_library_come_back_as_statuses(rsx):
    """rsx_selftest_firewall throws inside an extern "C" entry: an impossible std::vector size (std::length_error), a
    std::bad_alloc, a std::runtime_error (what nanoflann throws through the reference's SCManager, NF.hpp:1228,1324) and a
    non-std exception.  Each returns a status and a message; the process lives and the library keeps working."""
    L = rsx.lib()
    for kind, want, word in ((0, -4, b"allocation"), (1, -4, b"bad_alloc"), (2, -7, b"KDTreeSingleIndexAdaptor"), (3, -7, b"unexpected")):
        assert L.rsx_selftest_firewall(kind) == want, kind
        assert word in L.rsx_last_error_string(), (kind, L.rsx_last_error_string())
    assert L.rsx_selftest_firewall(99) == 0
    keys = np.zeros((4, 20), dtype=np.float32)
    vind = np.zeros(4, dtype=np.int32)
    assert L.rsx_sc_ringkey_tree_layout(keys.ctypes.data, C.c_int64(1 << 61), vind.ctypes.data, None, None) == -1   # refused, not thrown
    assert L.rsx_sc_ringkey_tree_layout(keys.ctypes.data, C.c_int64(4), vind.ctypes.data, None, None) == 0
    assert sorted(vind.tolist()) == [0, 1, 2, 3]


def test_diag_header_is_separate_and_versioned(rsx):
    """The diagnostic entries live in include/rsx_diag.h, not in the boundary header, and the re-scoring counters come
    through ONE call with a struct_size-versioned struct (no _rescoring2/_rescoring3 generations)."""
    main = open(os.path.join(ROOT, "include", "rsx.h")).read()
    diag = open(os.path.join(ROOT, "include", "rsx_diag.h")).read()
    for name in ("rsx_sc_profile_read", "rsx_sc_window_previews", "rsx_sc_ringkey_tree_layout", "rsx_selftest_firewall",
                 "rsx_sc_dominant_kernel_name", "rsx_sc_filter_bounds"):
        assert name + "(" not in re.sub(r"/\*.*?\*/", "", main, flags=re.S), name
        assert name + "(" in diag, name
    assert not re.search(r"rsx_sc_profile_read_rescoring[23]", main + diag)
    assert C.sizeof(rsx.RescoringStats) == 56
    L = rsx.lib()
    assert L.rsx_sc_profile_read_rescoring(None, None) != 0
