"""The max-clique inlier selection between the matcher and the ORORA solver (upstream sources
absent: parity unpinned).  CPU side: the C oracle (oracle/pmc_ref.c) against its definitions, against the independent numpy
restatement (oracle/pmc_np.py) and against an exact branch-and-bound solver."""
import itertools
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navtech_radar_slam_amd import synth  # noqa: E402
from oracle import pmc_np  # noqa: E402

TAU = 1.5  # rsx_orora_default_params().tim_noise_bound


def brute_force_clique_number(adj):
    k = len(adj)
    for size in range(k, 0, -1):
        for c in itertools.combinations(range(k), size):
            if all(adj[a, b] for a, b in itertools.combinations(c, 2)):
                return size
    return 0


def test_edge_predicate_is_the_distance_consistency_test(oracle):
    """The sqrt-free form decides | ||da|| - ||db|| | < tau: checked against the direct form in extended precision away from
    the boundary, symmetric, no self loops, NaN -> no edge."""
    rng = np.random.default_rng(1)
    src = rng.uniform(-150, 150, (300, 2)).astype(np.float32)
    dst = (src + rng.normal(0, 1.0, (300, 2))).astype(np.float32)
    adj = oracle.pmc_adjacency(src, dst, TAU)
    assert np.array_equal(adj, adj.T) and not adj.diagonal().any()
    s, d = src.astype(np.longdouble), dst.astype(np.longdouble)
    da = np.sqrt(((s[None] - s[:, None]) ** 2).sum(-1))
    db = np.sqrt(((d[None] - d[:, None]) ** 2).sum(-1))
    gap = np.abs(da - db) - TAU
    clear = np.abs(gap) > 1e-9
    np.fill_diagonal(clear, False)
    assert np.array_equal(adj[clear].astype(bool), (gap < 0)[clear])
    assert 0.2 < adj.mean() < 0.95
    src[7] = np.nan
    adj = oracle.pmc_adjacency(src, dst, TAU)
    assert not adj[7].any() and not adj[:, 7].any()
    assert np.array_equal(adj.astype(bool), pmc_np.adjacency(src, dst, TAU))


def test_core_numbers_match_the_definition(oracle):
    rng = np.random.default_rng(2)
    for k, p in ((1, 0.5), (2, 1.0), (40, 0.1), (60, 0.5), (90, 0.9), (130, 0.3)):
        a = np.triu(rng.random((k, k)) < p, 1)
        adj = (a | a.T).astype(np.uint8)
        assert np.array_equal(oracle.pmc_core_numbers(adj), pmc_np.core_numbers(adj.astype(bool))), (k, p)
    # a clique of 6 with a pendant path: cores 5 for the clique, 1 for the path
    adj = np.zeros((9, 9), dtype=np.uint8)
    adj[:6, :6] = 1 - np.eye(6, dtype=np.uint8)
    for a, b in ((5, 6), (6, 7), (7, 8)):
        adj[a, b] = adj[b, a] = 1
    assert oracle.pmc_core_numbers(adj).tolist() == [5] * 6 + [1] * 3


def test_exact_solver_against_brute_force(oracle):
    rng = np.random.default_rng(3)
    for k, p in ((8, 0.5), (12, 0.6), (14, 0.8), (16, 0.4)):
        for _ in range(4):
            a = np.triu(rng.random((k, k)) < p, 1)
            adj = (a | a.T).astype(np.uint8)
            assert oracle.pmc_exact_size(adj) == brute_force_clique_number(adj)


def test_select_c_equals_numpy_and_is_a_clique(oracle):
    src, dst, off, _ = synth.orora_pairs(31, 10, k_range=(20, 260))
    member, info = oracle.pmc_select_batch(src, dst, off, TAU, nthreads=4)
    for i in range(10):
        s, d = src[off[i]:off[i + 1]], dst[off[i]:off[i + 1]]
        m_np, inf_np = pmc_np.select(s, d, TAU)
        m = member[off[i]:off[i + 1]].astype(bool)
        assert np.array_equal(m, m_np), i
        assert {f: int(info[i][f]) for f in ("size", "max_core", "seeds", "flags")} == inf_np, i
        adj = oracle.pmc_adjacency(s, d, TAU).astype(bool)
        idx = np.flatnonzero(m)
        assert len(idx) == info[i]["size"] and adj[np.ix_(idx, idx)].sum() == len(idx) * (len(idx) - 1)   # a clique
        assert info[i]["size"] <= info[i]["max_core"] + 1                                                  # the core bound
        # maximal: no vertex outside is adjacent to all members
        assert not (adj[:, idx].all(axis=1) & ~m).any()


def test_greedy_clique_is_maximum_on_planted_data(oracle):
    """On the bench's kind of data (a planted consistent set + random outliers) the greedy clique has the exact clique number
    (independent branch and bound, started from size - 1 so that it has to find a clique of that size itself)."""
    src, dst, off, _ = synth.orora_pairs(32, 6, k_range=(60, 200))
    member, info = oracle.pmc_select_batch(src, dst, off, TAU)
    for i in range(6):
        adj = oracle.pmc_adjacency(src[off[i]:off[i + 1]], dst[off[i]:off[i + 1]], TAU)
        exact = oracle.pmc_exact_size(adj, lb=int(info[i]["size"]) - 1, max_nodes=5_000_000)
        assert exact == info[i]["size"], (i, exact, info[i])


def test_selection_keeps_inliers_and_drops_outliers(oracle):
    src, dst, off, truth = synth.orora_pairs(33, 4, k_range=(300, 500), outlier_range=(0.5, 0.6))
    member, info = oracle.pmc_select_batch(src, dst, off, TAU)
    for i in range(4):
        s, d = src[off[i]:off[i + 1]].astype(np.float64), dst[off[i]:off[i + 1]].astype(np.float64)
        x, y, yaw = truth[i]
        c, sn = np.cos(yaw), np.sin(yaw)
        pred = s @ np.array([[c, sn], [-sn, c]]) + (x, y)
        err = np.hypot(*(pred - d).T)
        m = member[off[i]:off[i + 1]].astype(bool)
        assert (err[m] < 3.0).mean() > 0.99          # what is kept is consistent with the true motion
        assert m.sum() > 0.5 * (err < 1.0).sum()      # and most clear inliers are kept


def test_edge_cases(oracle):
    # K = 0, 1: pass-through; all-identical points: one big clique, proven; no consistent pair at all: size 1
    off = np.array([0, 0, 1, 6, 10], dtype=np.int64)
    src = np.zeros((10, 2), dtype=np.float32)
    dst = np.zeros((10, 2), dtype=np.float32)
    src[6:10] = [[0, 0], [10, 0], [20, 0], [30, 0]]
    dst[6:10] = [[0, 0], [50, 0], [150, 0], [300, 0]]
    member, info = oracle.pmc_select_batch(src, dst, off, TAU)
    assert info["size"].tolist() == [0, 1, 5, 1]
    assert info["flags"].tolist() == [oracle.PMC_PASSTHROUGH, oracle.PMC_PASSTHROUGH, oracle.PMC_PROVEN, oracle.PMC_PROVEN]
    assert member.tolist() == [1] + [1] * 5 + [1, 0, 0, 0]
    s2, d2, o2 = oracle.pmc_compact(src, dst, off, member)
    assert o2.tolist() == [0, 0, 1, 6, 7]
