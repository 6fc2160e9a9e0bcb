"""GPU parity tests of the MFMA lower-bound filter (csrc/sc_filter.hip) in front of the exact kernel.

Two contracts:
  1. bound:   filter_bounds(q, e) - eps <= exact distance(q, e) for EVERY pair, and the bound is the
              all-60-shift minimum of the column-cosine distance up to eps (checks the MFMA fragment /
              circulant layout, not just looseness);
  2. results: an exhaustive query through the filter returns byte-identical top-k records to the
              oracle (and to the unfiltered exact path), including ties, padding and eligibility.
"""
import os

import numpy as np
import pytest

from navtech_radar_slam_amd import synth

pytestmark = pytest.mark.gpu

FORCE, OFF = 2, 1
FILTER_KERNELS = ("sc_filter_kernel", "sc_spec_filter_kernel", "sc_spec2_filter_kernel")


@pytest.fixture(scope="module")
def sc():
    from navtech_radar_slam_amd import _rsx, scancontext
    assert _rsx.device_count() >= 1, "no HIP device: GPU tests must run on the MI355X box"
    return scancontext


@pytest.fixture(autouse=True, params=["direct", "spectral", "spectral2"])
def filter_kind(request):
    """every test of this module runs with every form of the filter (sc_filter.hip / sc_spec.hip, one and two waves per SIMD)"""
    from navtech_radar_slam_amd import _rsx
    _rsx.default_filter_kind = {"direct": _rsx.KIND_DIRECT, "spectral": _rsx.KIND_SPECTRAL, "spectral2": _rsx.KIND_SPECTRAL2}[request.param]
    yield request.param
    _rsx.default_filter_kind = _rsx.KIND_AUTO


def all_shift_bound(q, descs):
    """fp64 reference of the filter's quantity: min over all 60 shifts of distDirectSC (SC.cpp:69-90)."""
    D = descs.reshape(-1, 60, 20).astype(np.float64)
    Q = q.reshape(60, 20).astype(np.float64)
    dn = np.sqrt((D ** 2).sum(2))
    qn = np.sqrt((Q ** 2).sum(1))
    Dn = np.where(dn[..., None] > 0, D / np.where(dn == 0, 1, dn)[..., None], 0)
    Qn = np.where(qn[:, None] > 0, Q / np.where(qn == 0, 1, qn)[:, None], 0)
    lb = np.full(len(D), np.inf)
    for k in range(60):
        S = np.einsum("jr,njr->n", np.roll(Qn, -k, axis=0), Dn)
        ne = (np.roll(qn > 0, -k)[None, :] & (dn > 0)).sum(1)
        with np.errstate(divide="ignore", invalid="ignore"):
            d = np.where(ne > 0, 1.0 - S / ne, np.inf)
        lb = np.minimum(lb, d)
    return lb


def make_db(seed, n, binary):
    descs = synth.random_descriptors(seed, n, binary=binary)
    rng = np.random.default_rng(seed + 1)
    for i in range(0, n, 7):  # rotated (and partly corrupted) copies: near-zero distances and exact ties
        j = int(rng.integers(0, n))
        descs[i] = synth.rotate_descriptor(descs[j], int(rng.integers(0, 60)))
        if i % 3 == 0:
            descs[i][rng.integers(0, 1200, 50)] = 0
    descs[5] = 0                       # all-zero descriptor: never a hit (SC.cpp:87)
    descs[6][20 * 7:20 * 9] = 0        # blank sectors
    if not binary:
        descs[8][0:20] = 1e-4          # a column of tiny values next to O(1) columns
        descs[9] *= -1.0               # negative heights (z < -2)
    return descs


def two_sided(lb, want):
    """|lb - want| with the storage format taken out: the bound matrix is fp16 rounded TOWARD ZERO (sc_kernels.h lb_t), so a
    stored bound may sit up to one fp16 ulp (<= 2^-10 |v|) below the value the filter computed, never above it"""
    return np.maximum(lb - want, (want - lb) - 2.0 ** -10 * np.maximum(np.abs(want), 2.0 ** -14))


@pytest.mark.parametrize("binary", [True, False])
def test_filter_bounds(sc, oracle, binary, filter_kind):
    if filter_kind != "direct":
        pytest.skip("two-sided bound of the direct form; the spectral form: test_gpu_sc_spec.py")
    n, nq = 1000 + 13, 20                                # not a multiple of 32: ragged last tile
    descs = make_db(100 + binary, n, binary)
    rng = np.random.default_rng(9)
    queries = np.stack([synth.rotate_descriptor(descs[int(rng.integers(0, n))], int(rng.integers(0, 60))) for _ in range(nq)])
    queries[:, rng.integers(0, 1200, 20)] = 0
    queries[1] = descs[5]                                # all-zero query
    queries[2] = descs[8] if not binary else descs[6]
    g = sc.SCManager()
    g.add_descriptors_f32(descs)
    o = oracle.Manager()
    o.add_descriptors(descs.astype(np.float64))
    eps = g.filter_eps()
    lb = g.filter_bounds(queries)
    assert lb.shape == (nq, n)
    for qi in range(nq):
        dist, _ = o.pair_distances(queries[qi].astype(np.float64), nthreads=4)
        want = all_shift_bound(queries[qi], descs)
        fin = np.isfinite(want)
        assert np.all(lb[qi][~fin] == np.inf), "no effective column at any shift -> +inf"
        assert not fin.any() or two_sided(lb[qi][fin], want[fin]).max() <= eps, f"q={qi}: bound is not the all-shift minimum"
        hit = dist < 1e7
        assert np.all(lb[qi][hit].astype(np.float64) - eps <= dist[hit]), f"q={qi}: not a lower bound"
    # the observed error is far inside the budget (documents the margin)
    qi = 3
    want = all_shift_bound(queries[qi], descs)
    fin = np.isfinite(want)
    assert two_sided(lb[qi][fin], want[fin]).max() < 0.6 * eps


@pytest.mark.parametrize("binary", [True, False])
@pytest.mark.parametrize("k", [1, 10, 32])
def test_filtered_query_matches_oracle(sc, oracle, binary, k, filter_kind):
    n, nq = 2500 + 5, 40
    descs = make_db(7 + binary, n, binary)
    rng = np.random.default_rng(11)
    queries = np.stack([synth.rotate_descriptor(descs[int(rng.integers(0, n))], int(rng.integers(0, 60))) for _ in range(nq)])
    queries[::2, rng.integers(0, 1200, 60)] = 0
    queries[1] = 0
    g = sc.SCManager(filter_mode=FORCE)
    g.add_descriptors_f32(descs)
    o = oracle.Manager()
    o.add_descriptors(descs.astype(np.float64))
    got = g.query(queries, k=k, n_eligible=n - 30)
    assert g.profiled_kernel_name() == {"direct": "sc_filter_kernel", "spectral": "sc_spec_filter_kernel", "spectral2": "sc_spec2_filter_kernel"}[filter_kind]
    for qi in range(nq):
        want = o.exhaustive(queries[qi].astype(np.float64), n_eligible=n - 30, k=k, nthreads=4)
        assert np.array_equal(got[qi], want), f"query {qi}"
    assert np.all(got[1]["dist"] == 1e7) and np.all(got[1]["index"] == 0)   # zero query: padding only


def test_filtered_equals_unfiltered_10k(sc, oracle):
    n, nq, k = 10000, 256, 10
    descs = synth.random_descriptors(1234, n, binary=True)
    rng = np.random.default_rng(4321)
    src = rng.integers(0, n - 100, nq)
    rot = rng.integers(0, 60, nq)
    queries = np.stack([synth.rotate_descriptor(descs[s], int(r)) for s, r in zip(src, rot)])
    np.put_along_axis(queries, rng.integers(0, 1200, (nq, 24)), 0.0, axis=1)
    a = sc.SCManager(capacity_hint=n, filter_mode=FORCE)
    b = sc.SCManager(capacity_hint=n, filter_mode=OFF)
    a.add_descriptors_f32(descs)
    b.add_descriptors_f32(descs)
    ga = a.query(queries, k=k, n_eligible=n - 30)
    gb = b.query(queries, k=k, n_eligible=n - 30)
    assert a.profiled_kernel_name() in FILTER_KERNELS and b.profiled_kernel_name() == "sc_pair_kernel"
    assert np.array_equal(ga, gb)
    assert np.array_equal(ga["index"][:, 0], src) and np.array_equal(ga["shift"][:, 0], rot)
    o = oracle.Manager()
    o.add_descriptors(descs.astype(np.float64))
    # EVERY query of the batch against the oracle (OpenMP over queries on all host cores)
    want = o.exhaustive_batch(queries.astype(np.float64), n_eligible=n - 30, k=k, nthreads=os.cpu_count() or 8)
    assert np.array_equal(ga, want)
    assert np.array_equal(a.query(queries, k=k, n_eligible=n - 30), ga)   # deterministic
    # auto mode takes the filter for this batch size
    c = sc.SCManager(capacity_hint=n)
    c.add_descriptors_f32(descs)
    assert np.array_equal(c.query(queries, k=k, n_eligible=n - 30), ga)
    assert c.profiled_kernel_name() in FILTER_KERNELS


def test_filtered_edge_cases(sc, oracle):
    descs = make_db(3, 70, binary=False)
    bad = descs.copy()
    bad[11][3] = np.nan                                   # non-finite entries are always re-scored exactly
    bad[12][100] = np.inf
    for n in (1, 5, 31, 32, 33, 70):
        g = sc.SCManager(filter_mode=FORCE)
        g.add_descriptors_f32(bad[:n])
        o = oracle.Manager()
        o.add_descriptors(bad[:n].astype(np.float64))
        queries = np.stack([bad[0], synth.rotate_descriptor(bad[min(n - 1, 20)], 7), np.zeros(1200, np.float32), bad[min(n - 1, 11)]])
        for k, ne in ((1, -1), (10, -1), (32, -1), (3, max(0, n - 2)), (4, 0)):
            got = g.query(queries, k=k, n_eligible=ne)
            for qi in range(len(queries)):
                want = o.exhaustive(queries[qi].astype(np.float64), n_eligible=(n if ne < 0 else ne), k=k)
                assert np.array_equal(got[qi], want), f"n={n} k={k} ne={ne} q={qi}"
        g.close()


def test_filtered_ties_binary_duplicates(sc, oracle):
    # many exact duplicates: the top-k is decided by the index tie-break alone
    base = synth.random_descriptors(5, 6, binary=True)
    descs = np.stack([synth.rotate_descriptor(base[i % 6], (i * 7) % 60) for i in range(600)])
    g = sc.SCManager(filter_mode=FORCE)
    g.add_descriptors_f32(descs)
    o = oracle.Manager()
    o.add_descriptors(descs.astype(np.float64))
    queries = np.stack([synth.rotate_descriptor(base[i], 3 * i) for i in range(6)])
    got = g.query(queries, k=32)
    for qi in range(6):
        assert np.array_equal(got[qi], o.exhaustive(queries[qi].astype(np.float64), k=32))
        assert np.all(np.abs(got[qi]["dist"]) < 1e-14) and np.all(np.diff(got[qi]["index"]) > 0)


def test_filtered_query_self_with_exclusion(sc, oracle):
    import torch
    n = 400
    clouds, truth = synth.keyframe_clouds(1234, n, binary_z=True, loop_frac=0.1, n_points=500)
    g = sc.SCManager(filter_mode=FORCE)
    o = oracle.Manager()
    for c in clouds:
        g.makeAndSaveScancontextAndKeys(c)            # filter images built by the add_points path
        o.add_points(c)
    out = torch.zeros((n, 2, 2), dtype=torch.float64, device="cuda")
    g.query_self_device(0, n, 2, out.data_ptr(), exclude_recent=30, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(sc.HIT_DTYPE).reshape(n, 2)
    for i in list(range(0, 40)) + [i for i, t in enumerate(truth) if t is not None]:
        want = o.exhaustive(o.descriptor(i), n_eligible=max(0, i - 30), k=2)
        assert np.array_equal(got[i], want), f"query {i}"


def test_filtered_sharded_equals_unsharded(sc, oracle):
    import torch
    n, nq, k, world = 2003, 16, 10, 4
    descs = synth.random_descriptors(21, n, binary=True)
    queries = np.stack([synth.rotate_descriptor(descs[i * 17], i) for i in range(nq)])
    full = sc.SCManager(filter_mode=OFF)
    full.add_descriptors_f32(descs)
    want = full.query(queries, k=k, n_eligible=n - 30)
    shards = [sc.SCManager(shard_rank=r, shard_world=world, filter_mode=FORCE) for r in range(world)]
    dq = torch.from_numpy(queries).cuda()
    parts = torch.zeros((world, nq, k, 2), dtype=torch.float64, device="cuda")
    for r, s in enumerate(shards):
        s.add_descriptors_f32(descs)
        s.query_device(dq.data_ptr(), nq, k, parts[r].data_ptr(), n_eligible=n - 30,
                       stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    host_parts = parts.cpu().numpy().view(sc.HIT_DTYPE).reshape(world, nq, k)
    assert np.array_equal(sc.merge_topk(host_parts), want)


@pytest.mark.parametrize("world,mode", [(2, FORCE), (8, FORCE), (3, OFF)])
def test_two_stage_sharded_query(sc, oracle, world, mode):
    """The multi-GPU protocol on one device: stage 1 per shard -> merge -> global bound -> stage 2 per
    shard -> merge must give exactly the unsharded top-k (what sharded.ShardedScanContext does with
    RCCL all-gathers between the stages)."""
    import torch
    n, nq, k = 6000 + 7, 48, 10
    descs = make_db(31, n, binary=True)
    rng = np.random.default_rng(5)
    queries = np.stack([synth.rotate_descriptor(descs[int(rng.integers(0, n))], int(rng.integers(0, 60))) for _ in range(nq)])
    queries[::3, rng.integers(0, 1200, 40)] = 0
    queries[2] = 0
    full = sc.SCManager(filter_mode=OFF)
    full.add_descriptors_f32(descs)
    want = full.query(queries, k=k, n_eligible=n - 30)
    shards = [sc.SCManager(shard_rank=r, shard_world=world, filter_mode=mode) for r in range(world)]
    for s in shards:
        s.add_descriptors_f32(descs)
    # ONE explicit stream for every handle (stream 0 would select each handle's private stream and the
    # shards' stages would be unordered with the merges)
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    st = tstream.cuda_stream
    dq = torch.from_numpy(queries).cuda()
    parts = torch.zeros((world, nq, k, 2), dtype=torch.float64, device="cuda")
    glob = torch.zeros((nq, k, 2), dtype=torch.float64, device="cuda")
    for r, s in enumerate(shards):
        s.query_stage1_device(dq.data_ptr(), nq, k, parts[r].data_ptr(), n_eligible=n - 30, stream=st)
    shards[0].merge_device(parts.data_ptr(), world, nq, k, glob.data_ptr(), stream=st)
    torch.cuda.synchronize()
    stage1 = glob.cpu().numpy().view(sc.HIT_DTYPE).reshape(nq, k)
    # the stage-1 bound is a valid upper bound of the final k-th best distance
    assert np.all(stage1["dist"][:, -1] >= want["dist"][:, -1])
    finals = torch.zeros((world, nq, k, 2), dtype=torch.float64, device="cuda")
    for r, s in enumerate(shards):
        s.query_stage2_device(nq, k, glob.data_ptr(), finals[r].data_ptr(), stream=st)
    out = torch.zeros((nq, k, 2), dtype=torch.float64, device="cuda")
    shards[0].merge_device(finals.data_ptr(), world, nq, k, out.data_ptr(), stream=st)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(sc.HIT_DTYPE).reshape(nq, k)
    assert np.array_equal(got, want)
    o = oracle.Manager()
    o.add_descriptors(descs.astype(np.float64))
    for qi in (0, 2, 47):
        assert np.array_equal(got[qi], o.exhaustive(queries[qi].astype(np.float64), n_eligible=n - 30, k=k, nthreads=4))
    # stage 2 without stage 1 is an error, not a silent wrong answer
    from navtech_radar_slam_amd._rsx import RsxError
    with pytest.raises(RsxError):
        shards[0].query_stage2_device(nq, k, glob.data_ptr(), finals[0].data_ptr(), stream=st)
    torch.cuda.synchronize()
    torch.cuda.set_stream(torch.cuda.default_stream())


def test_full_size_100k_properties(sc, oracle):
    """BASELINE config 5 scale on one GPU: 100 000-keyframe DB (0.8 GB resident), batched exhaustive
    top-10.  Size-independent properties (planted loops come back as top-1 with distance ~0 and the
    right shift, lists sorted, deterministic, filtered == unfiltered on a sample) + oracle spot checks."""
    n, nq, k = 100_000, 96, 10
    descs = synth.random_descriptors(777, n, binary=True)
    rng = np.random.default_rng(778)
    src = rng.integers(0, n - 100, nq)
    rot = rng.integers(0, 60, nq)
    queries = np.stack([synth.rotate_descriptor(descs[s], int(r)) for s, r in zip(src, rot)])
    g = sc.SCManager(capacity_hint=n)                     # auto mode: 96 x 100k pairs -> filter path
    g.add_descriptors_f32(descs)
    assert len(g) == n
    got = g.query(queries, k=k, n_eligible=n - 30)
    assert g.profiled_kernel_name() in FILTER_KERNELS
    assert np.array_equal(got["index"][:, 0], src) and np.array_equal(got["shift"][:, 0], rot)
    assert np.all(np.abs(got["dist"][:, 0]) < 1e-15)
    assert np.all(np.diff(got["dist"], axis=1) >= 0) and np.all(got["dist"] < 1e7)
    assert np.array_equal(g.query(queries, k=k, n_eligible=n - 30), got)
    e = sc.SCManager(capacity_hint=n, filter_mode=OFF)
    e.add_descriptors_f32(descs)
    assert np.array_equal(e.query(queries[:8], k=k, n_eligible=n - 30), got[:8])
    o = oracle.Manager()
    o.add_descriptors(descs.astype(np.float64))
    # every query of the batch against the oracle (96 x 100k exact pair evaluations, all host cores)
    want = o.exhaustive_batch(queries.astype(np.float64), n_eligible=n - 30, k=k, nthreads=os.cpu_count() or 16)
    assert np.array_equal(got, want)


def test_rescore_beyond_the_short_list(sc, oracle):
    """Duplicate-heavy databases: more equal bounds than the 2048-entry short list holds, so the
    re-scoring has to fall back to scanning the bound row (sc_rescore_kernel's rest path)."""
    rng = np.random.default_rng(17)
    base = synth.random_descriptors(40, 8, binary=False)
    rand = make_db(41, 900, binary=False)
    # (i) 5000 rotated copies of ONE descriptor + 900 others: the first histogram bin alone overflows
    #     the short list (empty short list, everything through the rest path)
    dup = np.stack([synth.rotate_descriptor(base[0], int(r)) for r in rng.integers(0, 60, 5000)])
    db1 = np.concatenate([rand[:450], dup, rand[450:]])
    # (ii) 5 strictly better entries + 3000 identical runners-up: the short list holds only the 5, the
    #      k-th best lies in the overflowing bin
    q2 = base[1].copy()
    near = []
    for i in range(5):
        d = q2.copy()
        d[rng.integers(0, 1200, 3 + i)] = 0
        near.append(d)
    far = q2.copy()
    far[rng.integers(0, 1200, 300)] = 0
    db2 = np.concatenate([rand[:100], np.tile(far, (3000, 1)), np.stack(near), rand[100:]])
    for db, q in ((db1, base[0]), (db2, q2)):
        g = sc.SCManager(filter_mode=FORCE, capacity_hint=len(db))
        g.add_descriptors_f32(db)
        o = oracle.Manager()
        o.add_descriptors(db.astype(np.float64))
        queries = np.stack([q, synth.rotate_descriptor(q, 11), rand[3], np.zeros(1200, np.float32)])
        for k, ne in ((1, -1), (10, -1), (32, -1), (10, 2000)):
            got = g.query(queries, k=k, n_eligible=ne)
            for qi in range(len(queries)):
                want = o.exhaustive(queries[qi].astype(np.float64), n_eligible=(len(db) if ne < 0 else ne), k=k, nthreads=4)
                assert np.array_equal(got[qi], want), f"db{1 if db is db1 else 2} k={k} ne={ne} q={qi}"
        g.close()


def test_streaming_slam_sharded(sc, oracle):
    """BASELINE config 4 in miniature: the DB grows keyframe by keyframe (descriptor build on the GPU,
    4 shards); at every 4th keyframe the newest descriptor is searched both ways -- reference
    semantics (3 kd-tree candidates, unsharded handle) and exhaustively over the shards with the
    two-stage protocol -- and both must equal the oracle."""
    import torch
    world, n, k = 4, 600, 3
    clouds, truth = synth.keyframe_clouds(99, n, binary_z=True, loop_frac=0.15, n_points=600, min_gap=40)
    ref = sc.SCManager(sc_dist_thres=0.45)
    shards = [sc.SCManager(shard_rank=r, shard_world=world, filter_mode=FORCE) for r in range(world)]
    o = oracle.Manager(dist_thres=0.45)
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    st = tstream.cuda_stream
    parts = torch.zeros((world, 1, k, 2), dtype=torch.float64, device="cuda")
    glob = torch.zeros((1, k, 2), dtype=torch.float64, device="cuda")
    out = torch.zeros((1, k, 2), dtype=torch.float64, device="cuda")
    found = 0
    for i, c in enumerate(clouds):
        assert ref.makeAndSaveScancontextAndKeys(c) == i
        for s in shards:
            assert s.makeAndSaveScancontextAndKeys(c) == i      # every rank sees every keyframe
        o.add_points(c)
        got = ref.detectLoopClosureID(full=True)
        want = o.detect_loop_closure()
        assert got == want, f"keyframe {i}: {got} vs {want}"
        if i % 4 or i < 31:
            continue
        n_elig = i + 1 - 30                                       # NUM_EXCLUDE_RECENT
        q = torch.from_numpy(o.descriptor(i).astype(np.float32)).cuda()
        for r, s in enumerate(shards):
            s.query_stage1_device(q.data_ptr(), 1, k, parts[r].data_ptr(), n_eligible=n_elig, stream=st)
        shards[0].merge_device(parts.data_ptr(), world, 1, k, glob.data_ptr(), stream=st)
        for r, s in enumerate(shards):
            s.query_stage2_device(1, k, glob.data_ptr(), parts[r].data_ptr(), stream=st)
        shards[0].merge_device(parts.data_ptr(), world, 1, k, out.data_ptr(), stream=st)
        torch.cuda.synchronize()
        hits = out.cpu().numpy().view(sc.HIT_DTYPE).reshape(k)
        wantx = o.exhaustive(o.descriptor(i), n_eligible=n_elig, k=k)
        assert np.array_equal(hits, wantx), f"keyframe {i}"
        if truth[i] is not None and truth[i][0] < n_elig:
            assert hits[0]["index"] == truth[i][0] and hits[0]["shift"] == truth[i][1]
            found += 1
    torch.cuda.synchronize()
    torch.cuda.set_stream(torch.cuda.default_stream())
    assert found >= 10 and sum(s.local_size for s in shards) == n


def test_filter_bounds_adversarial(sc, oracle, filter_kind):
    """The bound must hold for every pair also on descriptors built to stress the fp16 filter: columns
    mixing magnitudes over six decades, negative values, single-element columns, many empty columns,
    huge and tiny overall scales."""
    rng = np.random.default_rng(23)
    n = 640
    base = synth.random_descriptors(50, n, binary=False).reshape(n, 60, 20)
    scale = 10.0 ** rng.uniform(-3, 3, (n, 60, 20))
    d = (base * scale).astype(np.float32)
    d[100:200] *= np.where(rng.uniform(size=(100, 60, 20)) < 0.5, -1.0, 1.0).astype(np.float32)   # mixed signs
    d[200:260, :, 1:] = 0                                                                           # one ring only
    d[260:320][rng.uniform(size=(60, 60)) < 0.8] = 0                                                # mostly empty
    d[320:340] *= np.float32(1e30)                                                                  # huge (finite norms in fp64)
    d[340:360] *= np.float32(1e-30)                                                                 # tiny
    descs = np.ascontiguousarray(d.reshape(n, 1200))
    queries = np.stack([descs[5], synth.rotate_descriptor(descs[150], 17), descs[230], descs[300], descs[330], descs[350],
                        synth.rotate_descriptor(descs[40], 59)])
    g = sc.SCManager()
    g.add_descriptors_f32(descs)
    o = oracle.Manager()
    o.add_descriptors(descs.astype(np.float64))
    eps = g.filter_eps()
    lb = g.filter_bounds(queries)
    worst = 0.0
    for qi in range(len(queries)):
        dist, _ = o.pair_distances(queries[qi].astype(np.float64), nthreads=4)
        want = all_shift_bound(queries[qi], descs)
        fin = np.isfinite(want)
        assert np.all(lb[qi][~fin] == np.inf)
        if filter_kind == "direct":
            worst = max(worst, two_sided(lb[qi][fin], want[fin]).max())     # two-sided: it IS the all-shift minimum
        else:
            worst = max(worst, (lb[qi][fin] - eps - want[fin]).max() + eps)   # one-sided: never above it
        hit = dist < 1e7
        assert np.all(lb[qi][hit].astype(np.float64) - eps <= dist[hit]), f"q={qi}: not a lower bound"
    assert worst <= eps, worst
    # and the filtered top-k equals the oracle on the same data
    f = sc.SCManager(filter_mode=FORCE)
    f.add_descriptors_f32(descs)
    got = f.query(queries, k=10)
    for qi in range(len(queries)):
        assert np.array_equal(got[qi], o.exhaustive(queries[qi].astype(np.float64), k=10, nthreads=4)), qi


def test_self_queries_skip_invisible_tile_blocks(sc):
    """query_self with a growing eligibility prefix runs the filter on a triangular plan (tile-blocks a
    query cannot see are skipped): identical to the exact path for every query."""
    import torch
    n, k = 3000 + 17, 5
    descs = make_db(61, n, binary=True)
    a = sc.SCManager(filter_mode=FORCE, capacity_hint=n)
    b = sc.SCManager(filter_mode=OFF, capacity_hint=n)
    a.add_descriptors_f32(descs)
    b.add_descriptors_f32(descs)
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    st = tstream.cuda_stream
    oa = torch.zeros((n, k, 2), dtype=torch.float64, device="cuda")
    ob = torch.zeros((n, k, 2), dtype=torch.float64, device="cuda")
    for first, cnt, excl in ((0, n, 30), (1000, 1500, 0), (5, 64, 200)):
        oa.zero_()
        ob.zero_()
        a.query_self_device(first, cnt, k, oa.data_ptr(), exclude_recent=excl, stream=st)
        b.query_self_device(first, cnt, k, ob.data_ptr(), exclude_recent=excl, stream=st)
        torch.cuda.synchronize()
        ga = oa.cpu().numpy().view(sc.HIT_DTYPE).reshape(n, k)[:cnt]
        gb = ob.cpu().numpy().view(sc.HIT_DTYPE).reshape(n, k)[:cnt]
        assert np.array_equal(ga, gb), (first, cnt, excl)
    torch.cuda.set_stream(torch.cuda.default_stream())


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_all_pairs_over_shards(sc, world):
    """BASELINE config 5's protocol on one device: every keyframe against the keyframes at least 30
    older than itself, DB striped over `world` shards, queries replicated, per-query eligibility in
    the staged query == the unsharded self-query."""
    import torch
    n, k, excl = 2501, 10, 30
    descs = make_db(91, n, binary=True)
    descs[1700] = synth.rotate_descriptor(descs[40], 13)
    descs[2100] = synth.rotate_descriptor(descs[1700], 5)
    full = sc.SCManager(filter_mode=OFF)
    full.add_descriptors_f32(descs)
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    st = tstream.cuda_stream
    want_d = torch.zeros((n, k, 2), dtype=torch.float64, device="cuda")
    full.query_self_device(0, n, k, want_d.data_ptr(), exclude_recent=excl, stream=st)
    torch.cuda.synchronize()
    want = want_d.cpu().numpy().view(sc.HIT_DTYPE).reshape(n, k)
    assert want[1700, 0]["index"] == 40 and want[2100, 0]["index"] in (40, 1700)
    for mode in (FORCE, OFF):
        shards = [sc.SCManager(shard_rank=r, shard_world=world, filter_mode=mode) for r in range(world)]
        for s in shards:
            s.add_descriptors_f32(descs)
        dq = torch.from_numpy(descs).cuda()
        lim = torch.clamp(torch.arange(n, dtype=torch.int64, device="cuda") - excl, min=0)
        parts = torch.zeros((world, n, k, 2), dtype=torch.float64, device="cuda")
        glob = torch.zeros((n, k, 2), dtype=torch.float64, device="cuda")
        finals = torch.zeros((world, n, k, 2), dtype=torch.float64, device="cuda")
        out = torch.zeros((n, k, 2), dtype=torch.float64, device="cuda")
        for r, s in enumerate(shards):
            s.query_stage1_device(dq.data_ptr(), n, k, parts[r].data_ptr(), stream=st, q_elig_ptr=lim.data_ptr(),
                                  elig_monotone=True)
        shards[0].merge_device(parts.data_ptr(), world, n, k, glob.data_ptr(), stream=st)
        for r, s in enumerate(shards):
            s.query_stage2_device(n, k, glob.data_ptr(), finals[r].data_ptr(), stream=st)
        shards[0].merge_device(finals.data_ptr(), world, n, k, out.data_ptr(), stream=st)
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(sc.HIT_DTYPE).reshape(n, k)
        assert np.array_equal(got, want), mode
    torch.cuda.set_stream(torch.cuda.default_stream())
