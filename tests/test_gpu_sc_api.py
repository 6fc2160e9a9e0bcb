"""C-ABI entry points added in round 2: the reference's public helper methods as GPU calls, the detection record,
rounded import, bulk export, the on-disk database and the single-process multi-device handle (rsx_scs_*)."""
import ctypes as C

import numpy as np
import pytest

from navtech_radar_slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sc():
    from navtech_radar_slam_amd import _rsx, scancontext
    assert _rsx.device_count() >= 1
    return scancontext


def test_public_helpers_match_oracle_on_arbitrary_doubles(sc, oracle):
    """Scancontext.h:60-66 through rsx_sc_make_* / rsx_sc_dist_direct / rsx_sc_fast_align / rsx_sc_distance:
    bit-identical to the oracle (== the reference build) on built descriptors, on arbitrary fp32 values and on full
    doubles that no fp32 database could hold."""
    g = sc.SCManager()
    rng = np.random.default_rng(4)
    clouds, _ = synth.keyframe_clouds(8, 10, binary_z=False, loop_frac=0.4, min_gap=2, n_points=600)
    built = [oracle.make_scancontext(c) for c in clouds]
    for c, d in zip(clouds, built):
        assert np.array_equal(g.makeScancontext(c), d)
    sets = [np.stack(built), synth.random_descriptors(3, 8, binary=True).astype(np.float64), rng.normal(0, 1, (8, 1200))]
    sets[1][2].reshape(60, 20)[5:30] = 0
    sets[1][3][:] = 0
    for D in sets:
        for i in range(len(D)):
            assert np.array_equal(g.makeRingkeyFromScancontext(D[i]), oracle.ringkey(D[i]))
            assert np.array_equal(g.makeSectorkeyFromScancontext(D[i]), oracle.sectorkey(D[i]))
            for j in range(len(D)):
                a, b = g.distDirectSC(D[i], D[j]), oracle.dist_direct(D[i], D[j])
                assert a == b or (np.isnan(a) and np.isnan(b))
                assert g.fastAlignUsingVkey(oracle.sectorkey(D[i]), oracle.sectorkey(D[j])) == \
                    oracle.fast_align(oracle.sectorkey(D[i]), oracle.sectorkey(D[j]))
                assert g.distanceBtnScanContext(D[i], D[j]) == oracle.distance(D[i], D[j], literal=True)
    g.close()


def test_detect_ex_and_rounded_import(sc, oracle):
    g = sc.SCManager(sc_dist_thres=0.45)
    o = oracle.Manager(dist_thres=0.45)
    clouds, _ = synth.keyframe_clouds(12, 70, binary_z=True, loop_frac=0.3, min_gap=35, n_points=400)
    for i, c in enumerate(clouds):
        g.makeAndSaveScancontextAndKeys(c)
        o.add_points(c)
        r = g.detect_ex()
        want = o.detect_loop_closure()
        assert r.query_idx == i and r.searched == (1 if i >= 30 else 0) and r.dist_thres == 0.45
        assert (r.loop_id, r.yaw_diff_rad, r.min_dist, r.nn_idx) == want
    # arbitrary doubles: the strict entry refuses, the rounded one reports the rounding
    d = np.random.default_rng(0).uniform(0, 5, 1200)
    from navtech_radar_slam_amd._rsx import RsxError
    with pytest.raises(RsxError):
        g.saveScancontextAndKeys(d)
    idx, err = g.saveScancontextAndKeysRounded(d)
    assert idx == 70 and 0 < err <= np.abs(d - d.astype(np.float32)).max() + 1e-30
    assert np.array_equal(g.descriptor(70), d.astype(np.float32).astype(np.float64))
    d[5] = np.nan
    with pytest.raises(RsxError):
        g.saveScancontextAndKeysRounded(d)
    g.close()


def test_export_save_load_roundtrip(sc, oracle, tmp_path):
    descs = synth.random_descriptors(77, 301, binary=False)
    g = sc.SCManager()
    g.add_descriptors_f32(descs)
    assert np.array_equal(g.export_descriptors_f32(), descs)
    assert np.array_equal(g.export_descriptors_f32(17, 40), descs[17:57])
    path = str(tmp_path / "db.rsxscdb")
    g.save(path)
    raw = open(path, "rb").read()
    assert raw[:8] == b"RSXSCDB1" and len(raw) == 64 + 301 * 4800
    q = descs[[5, 250]]
    want = g.query(q, k=4)
    # unsharded file -> fresh handle, and -> two shard handles (each keeps its residue class)
    g2 = sc.SCManager()
    assert g2.load(path) == 301 and len(g2) == 301
    assert np.array_equal(g2.query(q, k=4), want)
    assert np.array_equal(g2.sectorkey(123), g.sectorkey(123)) and np.array_equal(g2.ringkey(300), g.ringkey(300))
    parts = []
    for r in range(2):
        s = sc.SCManager(shard_rank=r, shard_world=2)
        s.load(path)
        assert s.local_size == len(range(r, 301, 2))
        parts.append(s.query(q, k=4))
        # a shard file restores exactly that shard
        sp = str(tmp_path / f"shard{r}.rsxscdb")
        s.save(sp)
        s2 = sc.SCManager(shard_rank=r, shard_world=2)
        s2.load(sp)
        assert len(s2) == 301 and np.array_equal(s2.query(q, k=4), parts[-1])
        from navtech_radar_slam_amd._rsx import RsxError
        with pytest.raises(RsxError):
            sc.SCManager(shard_rank=1 - r, shard_world=2).load(sp)
    assert np.array_equal(sc.merge_topk(np.stack(parts)), want)
    from navtech_radar_slam_amd._rsx import RsxError
    bad = tmp_path / "bad.rsxscdb"
    bad.write_bytes(raw[:1000])
    with pytest.raises(RsxError):
        sc.SCManager().load(str(bad))
    # the header is not trusted: a shard file whose n_local is not the size of its residue class, a file that
    # is shorter or longer than its header says, an absurd n_local (must not allocate), a rank outside the world
    import struct
    shard0 = bytearray(open(str(tmp_path / "shard0.rsxscdb"), "rb").read())
    hdr = struct.Struct("<8sIIIIqqii16s")
    f = list(hdr.unpack_from(shard0))
    assert f[5] == 301 and f[6] == 151 and (f[7], f[8]) == (0, 2)

    def rejected(fields, payload):
        pth = tmp_path / "forged.rsxscdb"
        pth.write_bytes(hdr.pack(*fields) + bytes(payload))
        with pytest.raises(RsxError):
            sc.SCManager(shard_rank=0, shard_world=2).load(str(pth))

    body = shard0[64:]
    rejected(f[:6] + [150] + f[7:], body[:150 * 4800])             # one descriptor short of the residue class
    rejected(f[:5] + [303, 151] + f[7:], body)                      # n_global says 152 belong to rank 0
    rejected(f, body[:-4800])                                       # truncated
    rejected(f, body + b"\0" * 16)                                  # trailing bytes
    rejected(f[:5] + [2**61, 2**60] + f[7:], body)                  # would be 4.6e18 bytes: rejected before any allocation
    rejected(f[:7] + [2, 2] + f[9:], body)                          # rank outside the world
    ok = tmp_path / "forged.rsxscdb"
    ok.write_bytes(bytes(shard0))
    s3 = sc.SCManager(shard_rank=0, shard_world=2)
    assert s3.load(str(ok)) == 151


@pytest.mark.parametrize("shards", [2, 5])
def test_single_process_multi_device_handle(sc, oracle, shards):
    """rsx_scs_*: G shard handles driven by one process (here all on device 0), peer-copy exchanges; queries and the
    exhaustive detector equal the oracle / the unsharded handle."""
    from navtech_radar_slam_amd._rsx import HIT_DTYPE
    descs = synth.random_descriptors(5, 2600, binary=True)
    rng = np.random.default_rng(6)
    q = np.stack([synth.rotate_descriptor(descs[i], int(rng.integers(0, 60))) for i in rng.integers(0, 2500, 40)])
    q[3][:] = 0
    hs = sc.ShardedSet([0] * shards)
    hs.add_descriptors_f32(descs[:1000])
    hs.add_descriptors_f32(descs[1000:])
    assert len(hs) == 2600 and hs.num_shards == shards
    one = sc.SCManager()
    one.add_descriptors_f32(descs)
    for k, ne in ((10, 2570), (1, -1), (32, 40), (3, 0)):
        got = hs.query(q, k=k, n_eligible=ne)
        assert got.dtype == HIT_DTYPE and np.array_equal(got, one.query(q, k=k, n_eligible=ne)), (k, ne)
    o = oracle.Manager()
    o.add_descriptors(descs.astype(np.float64))
    got = hs.query(q[:6], k=5, n_eligible=2570)
    for i in range(6):
        assert np.array_equal(got[i], o.exhaustive(q[i].astype(np.float64), n_eligible=2570, k=5))
    assert np.array_equal(hs.descriptor(1234), descs[1234].astype(np.float64))
    hs.close()
    # the same devices as query groups x DB shards: 2 x (shards / 2 ...) -- every layout returns the same records
    for qg in [g for g in (shards, 1) if shards % g == 0] + ([2] if shards % 2 == 0 and shards > 2 else []):
        h2 = sc.ShardedSet([0] * shards, query_groups=qg)
        h2.add_descriptors_f32(descs)
        assert (h2.num_query_groups, h2.num_shards) == (qg, shards // qg)
        for k, ne in ((10, 2570), (1, -1)):
            assert np.array_equal(h2.query(q, k=k, n_eligible=ne), one.query(q, k=k, n_eligible=ne)), (qg, k, ne)
        assert np.array_equal(h2.query(q[:1], k=4, n_eligible=2570), one.query(q[:1], k=4, n_eligible=2570))   # groups with empty slices
        h2.close()
    one.close()


def test_rccl_exchange_world_one(sc):
    """exchange = RCCL (rsx_scs_create_layout, RSX_SCS_EXCHANGE_RCCL): librccl is loaded with dlopen, one communicator per
    device (ncclCommInitAll), the two stages exchange their records with ncclAllGather inside ncclGroupStart / End and
    every shard merges for itself.  This box has one GPU: a group of ONE shard still runs both collectives (a 1-rank
    all-gather), which exercises the binding, the communicator and the stream ordering; a device listed twice in a
    group is refused (RCCL needs distinct devices).  The multi-GPU form is the same code with S > 1."""
    from navtech_radar_slam_amd._rsx import RsxError
    descs = synth.random_descriptors(15, 1500, binary=True)
    rng = np.random.default_rng(16)
    q = np.stack([synth.rotate_descriptor(descs[i], int(rng.integers(0, 60))) for i in rng.integers(0, 1400, 24)])
    one = sc.SCManager()
    one.add_descriptors_f32(descs)
    hr = sc.ShardedSet([0], exchange="rccl")
    hr.add_descriptors_f32(descs)
    for k, ne in ((10, 1470), (1, -1), (32, 40)):
        assert np.array_equal(hr.query(q, k=k, n_eligible=ne), one.query(q, k=k, n_eligible=ne)), (k, ne)
    hr.close()
    with pytest.raises(RsxError, match="distinct devices"):
        sc.ShardedSet([0, 0], exchange="rccl")
    one.close()


@pytest.mark.parametrize("nq,n_db", [(2047, 1500), (2048, 1500), (4096, 1500), (5555, 1500), (8192, 1500), (4096, 24)])
def test_host_entry_in_pieces_equals_the_device_entry(sc, nq, n_db):
    """rsx_sc_query uploads a large batch in pieces (sc_api.cpp host_pieces) and filters each while the next one goes up, the
    stages behind the filter once over the whole batch; where the filter does not apply (24 entries: every pair scored exactly)
    each piece runs its own chain.  The records must be those of one device-resident call over the whole batch, from pageable
    and from pinned memory, for sizes below the cut (2047: one piece), at it, ragged, and the bench's."""
    import torch
    from navtech_radar_slam_amd import _rsx
    k = 3
    db = synth.random_descriptors(11, n_db, binary=True)
    db[5].reshape(60, 20)[10:25] = 0
    n_el = n_db - n_db // 15
    q = synth.random_descriptors(12, nq, binary=True)
    q[::7] = db[np.arange(len(q[::7])) % len(db)]          # revisits: exact ties between pieces and entries
    q[3].reshape(60, 20)[:30] = 0                          # a query with empty columns (the filter's slow path)
    g = sc.SCManager()
    g.add_descriptors_f32(db)
    dq = torch.from_numpy(q).cuda()
    ref = torch.zeros((nq, k, 2), dtype=torch.float64, device="cuda")
    g.query_device(dq.data_ptr(), nq, k, ref.data_ptr(), n_eligible=n_el, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = ref.cpu().numpy().view(sc.HIT_DTYPE).reshape(nq, k)
    got = g.query(q, k=k, n_eligible=n_el)
    assert np.array_equal(got, ref)
    with _rsx.PinnedArray((nq, 1200), np.float32) as pq, _rsx.PinnedArray((nq, k), sc.HIT_DTYPE) as po:
        pq.a[:] = q
        g.query(pq.a, k=k, n_eligible=n_el, out=po.a)
        assert np.array_equal(po.a, ref)
        g.query(pq.a[:100], k=k, n_eligible=n_el, out=po.a[:100])   # a small batch right after: workspaces reused
        assert np.array_equal(po.a[:100], ref[:100])
    g.close()


def test_one_launch_insert_equals_the_bulk_import_path(sc):
    """rsx_sc_add_points builds descriptor, keys and every per-entry image in ONE kernel (sc_insert_kernel); the bulk import
    (rsx_sc_add_descriptors_f32) computes keys and images with the batched kernels.  Same descriptors in, so the databases
    must be indistinguishable: descriptors, keys, filter bounds (= the spectral images) and query records bit for bit."""
    clouds, _ = synth.keyframe_clouds(9, 70, binary_z=False, loop_frac=0.4, min_gap=2, n_points=800)
    a = sc.SCManager()
    for c in clouds:
        a.makeAndSaveScancontextAndKeys(c)
    descs = a.export_descriptors_f32(0, len(clouds))
    b = sc.SCManager()
    b.add_descriptors_f32(descs)
    q = np.stack([synth.rotate_descriptor(descs[(7 * i) % len(descs)], 3 * i) for i in range(24)])
    q[3].reshape(60, 20)[:20] = 0
    assert np.array_equal(a.filter_bounds(q).view(np.uint32), b.filter_bounds(q).view(np.uint32))
    for i in (0, 13, len(clouds) - 1):
        assert np.array_equal(a.descriptor(i), b.descriptor(i))
        assert np.array_equal(a.ringkey(i), b.ringkey(i)) and np.array_equal(a.sectorkey(i), b.sectorkey(i))
    fa, fb = sc.SCManager(filter_mode=2), sc.SCManager(filter_mode=2)
    for c in clouds:
        fa.makeAndSaveScancontextAndKeys(c)
    fb.add_descriptors_f32(descs)
    assert np.array_equal(fa.query(q, k=5), fb.query(q, k=5))          # filter + window previews + exact re-scoring
    assert np.array_equal(a.query(q, k=5), fa.query(q, k=5))


def test_inserts_do_not_wait_and_later_calls_see_them(sc):
    """rsx_sc_add_points returns without waiting for the GPU: the caller's buffer is free at once (it is overwritten here right
    after every call), entries appear in order, and a query on the CALLER's stream -- not the handle's -- sees all of them."""
    import torch
    clouds, _ = synth.keyframe_clouds(10, 40, binary_z=False, loop_frac=0.3, min_gap=2, n_points=700)
    ref = sc.SCManager()
    for c in clouds:
        ref.makeAndSaveScancontextAndKeys(c)
    descs = ref.export_descriptors_f32(0, len(clouds))
    g = sc.SCManager()
    side = torch.cuda.Stream()
    d_q = torch.from_numpy(descs[:8].copy()).cuda()
    out = torch.zeros((8, 1, 2), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    buf = np.zeros((1000, clouds[0].shape[1]), dtype=np.float32)
    for i, c in enumerate(clouds):
        buf[:len(c)] = c
        g.makeAndSaveScancontextAndKeys(buf[:len(c)])                   # a contiguous view: the library reads THIS memory
        buf[:] = np.nan                                                # the call copied the cloud: this must not reach the GPU
        if i % 8 == 7:                                                 # straight after an insert, on a stream of our own
            g.query_device(d_q.data_ptr(), 8, 1, out.data_ptr(), n_eligible=-1, stream=side.cuda_stream)
            side.synchronize()
            got = out.cpu().numpy().view(sc.HIT_DTYPE).reshape(8)
            assert np.array_equal(got["index"], np.arange(8)) and np.all(got["dist"] < 1e-12), i   # every query finds itself
    assert np.array_equal(g.export_descriptors_f32(0, len(clouds)), descs)
