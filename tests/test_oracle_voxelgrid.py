"""Properties of the VoxelGrid oracle (oracle/voxelgrid_ref.c) derived from the published PCL algorithm
(pcl/filters/impl/voxel_grid.hpp, the filter the reference instantiates at
laserPosegraphOptimization.cpp:98 and runs at :482-484 with the 0.4 m leaf of :687-688).
PCL is not available here: parity with it is unpinned; these tests pin the restatement to its spec."""
import numpy as np


def _cloud(seed, n, extent=60.0):
    rng = np.random.default_rng(seed)
    p = np.zeros((n, 4), dtype=np.float32)
    p[:, 0:2] = rng.uniform(-extent, extent, (n, 2))
    p[:, 2] = rng.uniform(-2, 3, n)
    p[:, 3] = rng.uniform(0, 255, n)
    p[: n // 3, 0:2] = np.round(p[: n // 3, 0:2] / 7.0) * 7.0 + rng.normal(0, 0.15, (n // 3, 2))  # clusters: crowded voxels
    return p


def _spec(p, leaf):
    """numpy restatement of steps 3-6 for finite points (float32 arithmetic, stable order)."""
    fin = np.isfinite(p[:, :3]).all(1)
    q = p[fin]
    inv = np.float32(1.0) / np.float32(leaf)
    mn, mx = q[:, :3].min(0), q[:, :3].max(0)
    min_b = np.floor(mn * inv).astype(np.int32)
    div_b = np.floor(mx * inv).astype(np.int32) - min_b + 1
    mul = np.array([1, div_b[0], div_b[0] * div_b[1]], dtype=np.int64)
    ijk = (np.floor(q[:, :3] * inv) - min_b.astype(np.float32)).astype(np.int32)
    idx = (ijk * mul).sum(1)
    order = np.argsort(idx, kind="stable")
    out = []
    for v in np.unique(idx):
        s = np.zeros(4, dtype=np.float32)
        for r in q[order][idx[order] == v]:
            s = (s + r[:4]).astype(np.float32)
        out.append(s / np.float32((idx == v).sum()))
    return np.array(out, dtype=np.float32).reshape(-1, 4)


def test_matches_spec_and_is_sorted_by_voxel(oracle):
    for seed, n in [(1, 800), (2, 3000), (3, 57)]:
        p = _cloud(seed, n)
        got, ov = oracle.voxelgrid_filter(p, 0.4)
        assert not ov
        want = _spec(p, 0.4)
        assert got.shape == want.shape and np.array_equal(got, want)
        assert len(got) < n or n < 100          # clusters really share voxels
    # every centroid lies inside the bounding box of the cloud, and inside its own voxel cell
    p = _cloud(4, 2000)
    got, _ = oracle.voxelgrid_filter(p, 0.4)
    assert np.all(got[:, :3] >= p[:, :3].min(0) - 1e-4) and np.all(got[:, :3] <= p[:, :3].max(0) + 1e-4)


def test_edge_cases(oracle):
    assert len(oracle.voxelgrid_filter(np.zeros((0, 4), np.float32))[0]) == 0
    one = np.array([[1.0, 2.0, 3.0, 9.0]], dtype=np.float32)
    assert np.array_equal(oracle.voxelgrid_filter(one)[0], one)
    same = np.tile(one, (5, 1))
    assert np.array_equal(oracle.voxelgrid_filter(same)[0], one)          # five copies -> their centroid
    p = _cloud(5, 300)
    p[7, 0] = np.nan
    p[8, 2] = np.inf
    got, _ = oracle.voxelgrid_filter(p, 0.4)
    assert np.array_equal(got, _spec(p, 0.4)) and np.isfinite(got).all()
    allnan = np.full((4, 4), np.nan, dtype=np.float32)
    assert len(oracle.voxelgrid_filter(allnan)[0]) == 0
    # xyz-only input: intensity comes back 0
    got3, _ = oracle.voxelgrid_filter(p[:, :3], 0.4)
    assert np.array_equal(got3[:, :3], got[:, :3]) and not got3[:, 3].any()
    # "leaf size too small": the grid would need more than 2^31 cells -> input returned unchanged
    far = np.array([[0, 0, 0, 1], [3e4, 3e4, 3e3, 2], [1, 1, 1, 3]], dtype=np.float32)
    got, ov = oracle.voxelgrid_filter(far, 0.01)
    assert ov and np.array_equal(got, far)


def test_leaf_size_controls_resolution(oracle):
    p = _cloud(6, 4000)
    p[:, :3] += np.float32(100.0)              # all coordinates in (0, 500): one cell of the largest leaf
    sizes = [len(oracle.voxelgrid_filter(p, leaf)[0]) for leaf in (0.1, 0.4, 1.6, 6.4, 500.0)]
    assert sizes == sorted(sizes, reverse=True) and sizes[-1] == 1
    # the single centroid of a huge leaf is the float32 running mean of the cloud
    s = np.zeros(4, dtype=np.float32)
    for r in p:
        s = (s + r).astype(np.float32)
    assert np.array_equal(oracle.voxelgrid_filter(p, 500.0)[0][0], s / np.float32(len(p)))
