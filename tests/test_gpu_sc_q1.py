"""GPU parity tests of the one-launch single-query path (csrc/sc_q1.hip): the live detector's regime (PGO.cpp:561,577 ->
Scancontext.cpp:331-422), up to 8 queries per call.

Contract: byte-identical top-k records to the oracle (= the reference's pair function over every eligible entry, ranked by
(distance, index)) and to the exact-all / batched-filter paths -- ties of duplicated descriptors, all-zero and non-finite
descriptors, eligibility prefixes, k = 1..32, N = 1..100 000, ragged last tiles, sharded handles, per-query limits -- and the
same records when the call is repeated back to back (the arrival ticket and the write-through hand-off between workgroups
are re-used launch after launch with a warm cache)."""
import os

import numpy as np
import pytest

from navtech_radar_slam_amd import synth
from test_gpu_sc_filter import make_db

pytestmark = pytest.mark.gpu

AUTO, OFF, FORCE, Q1 = 0, 1, 2, 3


@pytest.fixture(scope="module")
def sc():
    from navtech_radar_slam_amd import _rsx, scancontext
    assert _rsx.device_count() >= 1, "no HIP device: GPU tests must run on the MI355X box"
    return scancontext


def queries_of(descs, nq, seed):
    rng = np.random.default_rng(seed)
    n = len(descs)
    q = np.stack([synth.rotate_descriptor(descs[int(rng.integers(0, n))], int(rng.integers(0, 60))) for _ in range(nq)])
    q[::2, rng.integers(0, 1200, 60)] = 0
    return q


@pytest.mark.parametrize("binary", [True, False])
@pytest.mark.parametrize("k", [1, 10, 32])
@pytest.mark.parametrize("nq", [1, 3, 8, 16])
def test_q1_matches_oracle(sc, oracle, binary, k, nq):
    n = 2500 + 5
    descs = make_db(7 + binary, n, binary)
    g = sc.SCManager(filter_mode=Q1)
    g.add_descriptors_f32(descs)
    o = oracle.Manager()
    o.add_descriptors(descs.astype(np.float64))
    for rep in range(3):
        queries = queries_of(descs, nq, 11 + rep)
        if rep == 1:
            queries[0] = 0   # zero query: padding only
        got = g.query(queries, k=k, n_eligible=n - 30)
        assert g.profiled_kernel_name() == "sc_q1_kernel"
        for qi in range(nq):
            want = o.exhaustive(queries[qi].astype(np.float64), n_eligible=n - 30, k=k, nthreads=4)
            assert np.array_equal(got[qi], want), f"rep {rep} query {qi}"
        if rep == 1:
            assert np.all(got[0]["dist"] == 1e7) and np.all(got[0]["index"] == 0)


def test_q1_edge_cases(sc, oracle):
    descs = make_db(3, 70, binary=False)
    bad = descs.copy()
    bad[11][3] = np.nan                                   # non-finite entries are always scored exactly
    bad[12][100] = np.inf
    for n in (1, 5, 31, 32, 33, 64, 65, 70):
        g = sc.SCManager(filter_mode=Q1)
        g.add_descriptors_f32(bad[:n])
        o = oracle.Manager()
        o.add_descriptors(bad[:n].astype(np.float64))
        queries = np.stack([bad[0], synth.rotate_descriptor(bad[min(n - 1, 20)], 7), np.zeros(1200, np.float32), bad[min(n - 1, 11)]])
        for k, ne in ((1, -1), (10, -1), (32, -1), (3, max(0, n - 2)), (4, 0)):
            got = g.query(queries, k=k, n_eligible=ne)
            assert g.profiled_kernel_name() == "sc_q1_kernel"
            for qi in range(len(queries)):
                want = o.exhaustive(queries[qi].astype(np.float64), n_eligible=(n if ne < 0 else ne), k=k)
                assert np.array_equal(got[qi], want), f"n={n} k={k} ne={ne} q={qi}"
        g.close()


def test_q1_empty_database(sc):
    g = sc.SCManager(filter_mode=Q1)
    got = g.query(synth.random_descriptors(1, 2, binary=True), k=3)
    assert np.all(got["dist"] == 1e7) and np.all(got["index"] == 0) and np.all(got["shift"] == 0)


def test_q1_ties_binary_duplicates(sc, oracle):
    # many exact duplicates: the top-k is decided by the index tie-break alone, and every duplicate has to be scored
    base = synth.random_descriptors(5, 6, binary=True)
    descs = np.stack([synth.rotate_descriptor(base[i % 6], (i * 7) % 60) for i in range(600)])
    g = sc.SCManager(filter_mode=Q1)
    g.add_descriptors_f32(descs)
    o = oracle.Manager()
    o.add_descriptors(descs.astype(np.float64))
    queries = np.stack([synth.rotate_descriptor(base[i], 3 * i) for i in range(6)])
    got = g.query(queries, k=32)
    for qi in range(6):
        assert np.array_equal(got[qi], o.exhaustive(queries[qi].astype(np.float64), k=32))
        assert np.all(np.abs(got[qi]["dist"]) < 1e-14) and np.all(np.diff(got[qi]["index"]) > 0)


def test_q1_more_survivors_than_one_chunk(sc):
    """5000 copies of one scan: every entry survives every bound (2048 survivors per chunk in the last workgroup), and the
    records are the lowest indices"""
    base = synth.random_descriptors(9, 1, binary=True)[0]
    descs = np.stack([synth.rotate_descriptor(base, (i * 11) % 60) for i in range(5000)])
    a = sc.SCManager(filter_mode=Q1, capacity_hint=5000)
    b = sc.SCManager(filter_mode=OFF, capacity_hint=5000)
    a.add_descriptors_f32(descs)
    b.add_descriptors_f32(descs)
    q = synth.rotate_descriptor(base, 17)[None]
    for k in (1, 4):
        ga, gb = a.query(q, k=k), b.query(q, k=k)
        assert np.array_equal(ga, gb)
        assert np.array_equal(ga[0]["index"], np.arange(k))


@pytest.mark.parametrize("n", [10000, 100000])
def test_q1_equals_exact_all_and_filtered(sc, n):
    nq = 48 if n == 10000 else 8
    descs = synth.random_descriptors(1234 + n, n, binary=True)
    rng = np.random.default_rng(4321)
    src = rng.integers(0, n - 100, nq)
    rot = rng.integers(0, 60, nq)
    queries = np.stack([synth.rotate_descriptor(descs[s], int(r)) for s, r in zip(src, rot)])
    np.put_along_axis(queries, rng.integers(0, 1200, (nq, 24)), 0.0, axis=1)
    queries[nq // 2:] = synth.random_descriptors(99, nq - nq // 2, binary=True)   # places never seen
    a = sc.SCManager(capacity_hint=n, filter_mode=Q1)
    b = sc.SCManager(capacity_hint=n, filter_mode=OFF)
    c = sc.SCManager(capacity_hint=n, filter_mode=FORCE)
    for h in (a, b, c):
        h.add_descriptors_f32(descs)
    for k in (1, 10):
        gb = b.query(queries, k=k, n_eligible=n - 30)
        gc = c.query(queries, k=k, n_eligible=n - 30)
        assert np.array_equal(gb, gc)
        for q0 in range(0, nq, 8):   # up to 8 queries per call take the single-query path
            ga = a.query(queries[q0:q0 + 8], k=k, n_eligible=n - 30)
            assert a.profiled_kernel_name() == "sc_q1_kernel"
            assert np.array_equal(ga, gb[q0:q0 + 8]), (n, k, q0)
        ga1 = a.query(queries[3:4], k=k, n_eligible=n - 30)   # and one at a time
        assert np.array_equal(ga1, gb[3:4])
    assert np.array_equal(gb["index"][:nq // 2, 0], src[:nq // 2]) and np.array_equal(gb["shift"][:nq // 2, 0], rot[:nq // 2])
    # auto mode: a handful of queries take this path, a batch does not
    d = sc.SCManager(capacity_hint=n)
    d.add_descriptors_f32(descs)
    assert np.array_equal(d.query(queries[:1], k=10, n_eligible=n - 30), gb[:1])
    assert d.profiled_kernel_name() == "sc_q1_kernel"
    if n == 10000:
        big = np.tile(queries, (6, 1))
        assert np.array_equal(d.query(big, k=10, n_eligible=n - 30), np.tile(gb, (6, 1)))
        assert d.profiled_kernel_name() != "sc_q1_kernel"


def test_q1_back_to_back_calls_with_changing_shapes(sc):
    """the arrival counters are never reset and the workgroup count changes with the eligible prefix: 300 calls in a row on
    one stream, device resident, different queries / limits / k, each compared with the exact-all path"""
    import torch
    n = 6000
    descs = synth.random_descriptors(31, n, binary=True)
    a = sc.SCManager(capacity_hint=n, filter_mode=Q1)
    b = sc.SCManager(capacity_hint=n, filter_mode=OFF)
    a.add_descriptors_f32(descs)
    b.add_descriptors_f32(descs)
    rng = np.random.default_rng(5)
    nqs = rng.integers(1, 9, 300)
    limits = rng.choice([n, n - 30, 4097, 4096, 1000, 33, 32, 31, 1, 0], 300)
    ks = rng.choice([1, 2, 10, 32], 300)
    pool = queries_of(descs, 64, 77)
    d_pool = torch.from_numpy(pool).cuda()
    outs_a, outs_b = [], []
    st = torch.cuda.current_stream().cuda_stream
    for i in range(300):
        nq, k = int(nqs[i]), int(ks[i])
        q0 = int(rng.integers(0, 64 - nq))
        oa = torch.zeros((nq, k, 2), dtype=torch.float64, device="cuda")
        ob = torch.zeros((nq, k, 2), dtype=torch.float64, device="cuda")
        a.query_device(d_pool[q0:].data_ptr(), nq, k, oa.data_ptr(), n_eligible=int(limits[i]), stream=st)
        b.query_device(d_pool[q0:].data_ptr(), nq, k, ob.data_ptr(), n_eligible=int(limits[i]), stream=st)
        outs_a.append(oa)
        outs_b.append(ob)
    torch.cuda.synchronize()
    for i in range(300):
        assert torch.equal(outs_a[i], outs_b[i]), (i, int(nqs[i]), int(limits[i]), int(ks[i]))


def test_q1_under_uneven_load(sc):
    """the hand-off between the workgroups of one launch while another stream keeps part of the chip busy with a batched
    filter query (uneven arrival order, warm caches): the records must not change"""
    import torch
    n = 20000
    descs = synth.random_descriptors(41, n, binary=True)
    a = sc.SCManager(capacity_hint=n, filter_mode=Q1)
    b = sc.SCManager(capacity_hint=n, filter_mode=OFF)
    load = sc.SCManager(capacity_hint=n, filter_mode=FORCE)
    for h in (a, b, load):
        h.add_descriptors_f32(descs)
    pool = queries_of(descs, 256, 3)
    d_pool = torch.from_numpy(pool).cuda()
    want = torch.zeros((256, 10, 2), dtype=torch.float64, device="cuda")
    for q0 in range(0, 256, 64):
        b.query_device(d_pool[q0:].data_ptr(), 64, 10, want[q0:].data_ptr(), n_eligible=n - 30)
    torch.cuda.synchronize()
    s_load, s_q = torch.cuda.Stream(), torch.cuda.Stream()
    big = torch.from_numpy(np.tile(pool, (8, 1))).cuda()
    big_out = torch.zeros((2048, 10, 2), dtype=torch.float64, device="cuda")
    got = torch.zeros((4, 256, 10, 2), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for rep in range(4):
        load.query_device(big.data_ptr(), 2048, 10, big_out.data_ptr(), n_eligible=n - 30, stream=s_load.cuda_stream)
        for q0 in range(0, 256, 4):
            a.query_device(d_pool[q0:].data_ptr(), 4, 10, got[rep, q0:].data_ptr(), n_eligible=n - 30, stream=s_q.cuda_stream)
    torch.cuda.synchronize()
    for rep in range(4):
        assert torch.equal(got[rep], want), rep


def test_q1_sharded_handles_and_per_query_limits(sc, oracle):
    n, world = 1500, 3
    descs = make_db(21, n, True)
    queries = queries_of(descs, 5, 8)
    full = sc.SCManager(filter_mode=OFF)
    full.add_descriptors_f32(descs)
    want = full.query(queries, k=7, n_eligible=n - 30)
    parts = []
    for r in range(world):
        s = sc.SCManager(shard_rank=r, shard_world=world, filter_mode=Q1)
        s.add_descriptors_f32(descs)       # a sharded handle keeps the entries it owns
        parts.append(s.query(queries, k=7, n_eligible=n - 30))
        assert s.profiled_kernel_name() == "sc_q1_kernel"
    assert np.array_equal(sc.merge_topk(np.stack(parts), k=7), want)
    # self queries with the reference's exclusion of recent keyframes: per-query eligibility limits
    import torch
    a = sc.SCManager(filter_mode=Q1)
    b = sc.SCManager(filter_mode=OFF)
    a.add_descriptors_f32(descs)
    b.add_descriptors_f32(descs)
    oa = torch.zeros((6, 3, 2), dtype=torch.float64, device="cuda")
    ob = torch.zeros((6, 3, 2), dtype=torch.float64, device="cuda")
    for q_first in (0, 28, 33, 700, n - 6):
        a.query_self_device(q_first, 6, 3, oa.data_ptr(), exclude_recent=30)
        b.query_self_device(q_first, 6, 3, ob.data_ptr(), exclude_recent=30)
        torch.cuda.synchronize()
        assert torch.equal(oa, ob), q_first


def test_detector_in_exhaustive_mode_takes_the_single_query_path(sc, oracle):
    from navtech_radar_slam_amd._rsx import MODE_EXHAUSTIVE
    clouds, _ = synth.keyframe_clouds(77, 120, binary_z=True, loop_frac=0.2, min_gap=35, n_points=400)
    g = sc.SCManager(sc_dist_thres=0.45)
    o = oracle.Manager(dist_thres=0.45)
    for i, c in enumerate(clouds):
        g.makeAndSaveScancontextAndKeys(c)
        o.add_points(c)
        lid, yaw, md, nn = g.detectLoopClosureID(mode=MODE_EXHAUSTIVE, full=True)
        o.detect_loop_closure()  # advances the oracle's tree period counter identically
        if len(o) < 31:
            assert lid == -1
            continue
        w = o.exhaustive(o.descriptor(i), n_eligible=o.tree_size, k=1)[0]
        assert (md, nn) == (w["dist"], w["index"]), i
        assert lid == (w["index"] if w["dist"] < 0.45 else -1)
    assert g.profiled_kernel_name() == "sc_q1_kernel"
